// Rulebook construction: coordinate hash, SubMConv3d gather table,
// SparseConv3d output set (ascending linear index) + coupled down/up gather tables, and the
// mask ordering used by the implicit-GEMM tiles.  Integer work, HBM/latency bound.
//
// Semantics restated from [UPSTREAM] spconv v1.0 getIndicePair:
//   pair (in p, out o) under kernel offset kappa iff p_j = o_j*s_j - pad_j + kappa_j, 0<=kappa_j<k_j,
//   0<=o_j<out_j; flat offset = (kappa0*k1+kappa1)*k2+kappa2.
#include <cstdlib>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_select.hpp>

#include "common.h"

using namespace wsis;

namespace {

struct Geo {
  int shape_in[3];
  int shape_out[3];
  int k[3];
  int s[3];
  int p[3];
};

__device__ __forceinline__ int64_t lin_key(int b, int c0, int c1, int c2, const int* S) {
  return (((int64_t)b * S[0] + c0) * S[1] + c1) * S[2] + c2;
}

__global__ void hash_insert_kernel(const int32_t* __restrict__ indices, int64_t M, Geo g,
                                   int64_t* __restrict__ keys, int32_t* __restrict__ vals,
                                   uint64_t mask) {
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < M;
       r += (int64_t)gridDim.x * blockDim.x) {
    const int4 c = reinterpret_cast<const int4*>(indices)[r];
    const int64_t key = lin_key(c.x, c.y, c.z, c.w, g.shape_in);
    uint64_t h = mix64((uint64_t)key) & mask;
    for (;;) {
      unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(keys + h),
                                         (unsigned long long)kEmptyKey, (unsigned long long)key);
      if (old == (unsigned long long)kEmptyKey || old == (unsigned long long)key) {
        atomicMin(vals + h, (int32_t)r);  // duplicates: smallest row wins (deterministic)
        break;
      }
      h = (h + 1) & mask;
    }
  }
}

// one thread per output row, loop over the K offsets: writes nbr[k*M + r] (coalesced per k)
__global__ void subm_kernel(const int32_t* __restrict__ indices, int64_t M, Geo g,
                            const int64_t* __restrict__ keys, const int32_t* __restrict__ vals,
                            uint64_t mask, int32_t* __restrict__ nbr, uint32_t* __restrict__ omask) {
  const int K = g.k[0] * g.k[1] * g.k[2];
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < M;
       r += (int64_t)gridDim.x * blockDim.x) {
    const int4 c = reinterpret_cast<const int4*>(indices)[r];
    uint32_t bits = 0;
    int kf = 0;
    for (int a = 0; a < g.k[0]; ++a) {
      const int x = c.y - g.p[0] + a;
      for (int b = 0; b < g.k[1]; ++b) {
        const int y = c.z - g.p[1] + b;
        for (int d = 0; d < g.k[2]; ++d, ++kf) {
          const int z = c.w - g.p[2] + d;
          int32_t v = -1;
          if (x >= 0 && x < g.shape_in[0] && y >= 0 && y < g.shape_in[1] && z >= 0 &&
              z < g.shape_in[2]) {
            if (x == c.y && y == c.z && z == c.w)
              v = (int32_t)r;
            else
              v = hash_lookup(keys, vals, mask, lin_key(c.x, x, y, z, g.shape_in));
          }
          nbr[(int64_t)kf * M + r] = v;
          if (v >= 0 && kf < 32) bits |= (1u << kf);
        }
      }
    }
    if (omask) omask[r] = bits;
  }
}

// 3x3x3 / pad 1 (every SubMConv3d of the UNet): the same lookups, nine at a time.  The generic kernel walks its 27
// probes one after the other (runtime loop bounds, a dependent key -> value load pair each: 27 x 2 memory latencies
// per thread); here the nine first-probe key loads of a plane are issued together, then the value loads of the hits,
// and only a collision falls back to the probing loop.
__global__ void subm3_kernel(const int32_t* __restrict__ indices, int64_t M, Geo g,
                             const int64_t* __restrict__ keys, const int32_t* __restrict__ vals,
                             uint64_t mask, int32_t* __restrict__ nbr, uint32_t* __restrict__ omask) {
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < M;
       r += (int64_t)gridDim.x * blockDim.x) {
    const int4 c = reinterpret_cast<const int4*>(indices)[r];
    uint32_t bits = 0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const int x = c.y - 1 + a;
      const bool xin = x >= 0 && x < g.shape_in[0];
      int64_t key[9], k0[9];
      uint32_t h[9];
      bool ok[9];
#pragma unroll
      for (int j = 0; j < 9; ++j) {
        const int y = c.z - 1 + j / 3, z = c.w - 1 + j % 3;
        ok[j] = xin && y >= 0 && y < g.shape_in[1] && z >= 0 && z < g.shape_in[2] && !(a == 1 && j == 4);
        key[j] = lin_key(c.x, x, y, z, g.shape_in);
        h[j] = (uint32_t)(mix64((uint64_t)key[j]) & mask);
      }
#pragma unroll
      for (int j = 0; j < 9; ++j) k0[j] = ok[j] ? keys[h[j]] : kEmptyKey;
      int32_t v[9];
#pragma unroll
      for (int j = 0; j < 9; ++j) v[j] = (ok[j] && k0[j] == key[j]) ? vals[h[j]] : -1;
#pragma unroll
      for (int j = 0; j < 9; ++j) {
        if (ok[j] && k0[j] != key[j] && k0[j] != kEmptyKey) {      // first slot taken by another key: keep probing
          uint64_t hh = (h[j] + 1) & mask;
          for (;;) {
            const int64_t k = keys[hh];
            if (k == key[j]) {
              v[j] = vals[hh];
              break;
            }
            if (k == kEmptyKey) break;
            hh = (hh + 1) & mask;
          }
        }
        if (a == 1 && j == 4) v[j] = (int32_t)r;                  // centre: the row itself
        const int kf = a * 9 + j;
        nbr[(int64_t)kf * M + r] = v[j];
        if (v[j] >= 0) bits |= (1u << kf);
      }
    }
    if (omask) omask[r] = bits;
  }
}

// candidate output keys of every active input; invalid candidates get key == invalid
// (batch_limit > 0: rows whose batch index is not in [0, batch_limit) have no candidate -- what the bitmap form does)
__global__ void down_cand_kernel(const int32_t* __restrict__ indices, int64_t M, Geo g, int fast,
                                 int64_t invalid, int64_t* __restrict__ cand, int32_t batch_limit) {
  const int K = g.k[0] * g.k[1] * g.k[2];
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < M;
       r += (int64_t)gridDim.x * blockDim.x) {
    const int4 c = reinterpret_cast<const int4*>(indices)[r];
    const int p[3] = {c.y, c.z, c.w};
    const bool in_batch = batch_limit <= 0 || (c.x >= 0 && c.x < batch_limit);
    if (fast) {
      int o[3];
      bool ok = in_batch;
      for (int j = 0; j < 3; ++j) {
        o[j] = p[j] / g.s[j];
        ok = ok && o[j] < g.shape_out[j];
      }
      cand[r] = ok ? lin_key(c.x, o[0], o[1], o[2], g.shape_out) : invalid;
    } else {
      int kf = 0;
      for (int a = 0; a < g.k[0]; ++a)
        for (int b = 0; b < g.k[1]; ++b)
          for (int d = 0; d < g.k[2]; ++d, ++kf) {
            const int kk[3] = {a, b, d};
            int o[3];
            bool ok = in_batch;
            for (int j = 0; j < 3; ++j) {
              const int t = p[j] + g.p[j] - kk[j];
              ok = ok && t >= 0 && (t % g.s[j]) == 0;
              o[j] = t / g.s[j];
              ok = ok && o[j] < g.shape_out[j];
            }
            cand[r * K + kf] = ok ? lin_key(c.x, o[0], o[1], o[2], g.shape_out) : invalid;
          }
    }
  }
}

// ---- output coordinates of a strided convolution WITHOUT a sort (round 5): the distinct output cells of a level are the
// set bits of a bitmap over the output grid (batch x shape_out, linear index = bit index), so "sorted unique keys" is a
// prefix popcount: mark (atomicOr), count per 1,024-word chunk, scan the chunk sums, emit.  5 launches instead of the 13
// of candidate keys + merge sort + unique (each a few microseconds of pure latency on ~10^5 keys); identical output --
// ascending linear index, SURVEY App. A.1.  Used when k == s, p == 0 (every input voxel has ONE output cell: the UNet's
// SparseConv3d k2 s2) and the bitmap fits the level's sort workspace; otherwise the sort path below.
constexpr int BM_WORDS_PER_WG = 1024;      // 256 threads x 4 words

__global__ void down_mark_kernel(const int32_t* __restrict__ indices, int64_t M, Geo g, int64_t n_cells,
                                 uint32_t* __restrict__ bitmap) {
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < M; r += (int64_t)gridDim.x * blockDim.x) {
    const int4 c = reinterpret_cast<const int4*>(indices)[r];
    const int o0 = c.y / g.s[0], o1 = c.z / g.s[1], o2 = c.w / g.s[2];
    if (o0 >= g.shape_out[0] || o1 >= g.shape_out[1] || o2 >= g.shape_out[2]) continue;
    const int64_t key = lin_key(c.x, o0, o1, o2, g.shape_out);
    if (c.x < 0 || key < 0 || key >= n_cells) continue;      // (a batch index outside [0, batch_size): not an output of this build; down_keys_sorted with a batch limit drops the same rows)
    atomicOr(bitmap + (key >> 5), 1u << (key & 31));
  }
}

__device__ __forceinline__ int bm_block_scan(int v, int* lds, int& total) {      // exclusive scan over 256 threads
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int x = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int y = __shfl_up(x, off, 64);
    if (lane >= off) x += y;
  }
  if (lane == 63) lds[wave] = x;
  __syncthreads();
  int base = 0;
  for (int w = 0; w < wave; ++w) base += lds[w];
  total = lds[0] + lds[1] + lds[2] + lds[3];
  __syncthreads();
  return base + x - v;
}

__global__ __launch_bounds__(256) void bitmap_count_kernel(const uint32_t* __restrict__ bitmap, int64_t n_words,
                                                           int32_t* __restrict__ chunk_sum) {
  __shared__ int lds[4];
  const int64_t w0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  int pc = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (w0 + j < n_words) pc += __popc(bitmap[w0 + j]);
  int total;
  (void)bm_block_scan(pc, lds, total);
  if (threadIdx.x == 0) chunk_sum[blockIdx.x] = total;
}

// one workgroup: chunk sums -> exclusive prefix in place, total -> *count
__global__ __launch_bounds__(256) void bitmap_scan_kernel(int32_t* __restrict__ chunk_sum, int n_chunks,
                                                          int32_t* __restrict__ count) {
  __shared__ int lds[4];
  int carry = 0;
  for (int c0 = 0; c0 < n_chunks; c0 += 256) {
    const int i = c0 + (int)threadIdx.x;
    const int v = i < n_chunks ? chunk_sum[i] : 0;
    int total;
    const int ex = bm_block_scan(v, lds, total);
    if (i < n_chunks) chunk_sum[i] = carry + ex;
    carry += total;
  }
  if (threadIdx.x == 0) *count = carry;
}

__global__ __launch_bounds__(256) void bitmap_emit_kernel(const uint32_t* __restrict__ bitmap, int64_t n_words,
                                                          const int32_t* __restrict__ chunk_base,
                                                          int64_t* __restrict__ out_keys) {
  __shared__ int lds[4];
  const int64_t w0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  uint32_t w[4];
  int pc = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    w[j] = w0 + j < n_words ? bitmap[w0 + j] : 0u;
    pc += __popc(w[j]);
  }
  int total;
  int64_t at = chunk_base[blockIdx.x] + bm_block_scan(pc, lds, total);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    uint32_t m = w[j];
    while (m) {
      const int b = __builtin_ctz(m);
      m &= m - 1u;
      out_keys[at++] = ((w0 + j) << 5) | b;
    }
  }
}

__global__ void down_fix_count_kernel(const int64_t* __restrict__ out_keys, int64_t invalid,
                                      int32_t* __restrict__ count) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    int32_t n = *count;
    if (n > 0 && out_keys[n - 1] == invalid) --n;
    *count = n;
  }
}

// decode sorted unique keys -> out indices, insert (key -> row) into the coarse hash
__global__ void down_decode_kernel(const int64_t* __restrict__ out_keys, int64_t M_out, Geo g,
                                   int32_t* __restrict__ indices_out, int64_t* __restrict__ keys,
                                   int32_t* __restrict__ vals, uint64_t mask) {
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < M_out;
       r += (int64_t)gridDim.x * blockDim.x) {
    int64_t key = out_keys[r];
    int64_t t = key;
    const int c2 = (int)(t % g.shape_out[2]);
    t /= g.shape_out[2];
    const int c1 = (int)(t % g.shape_out[1]);
    t /= g.shape_out[1];
    const int c0 = (int)(t % g.shape_out[0]);
    t /= g.shape_out[0];
    reinterpret_cast<int4*>(indices_out)[r] = make_int4((int)t, c0, c1, c2);
    uint64_t h = mix64((uint64_t)key) & mask;
    for (;;) {
      unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(keys + h),
                                         (unsigned long long)kEmptyKey, (unsigned long long)key);
      if (old == (unsigned long long)kEmptyKey) {
        vals[h] = (int32_t)r;
        break;
      }
      h = (h + 1) & mask;
    }
  }
}

__global__ void down_fill_kernel(const int32_t* __restrict__ indices, int64_t M_in, int64_t M_out,
                                 Geo g, const int64_t* __restrict__ keys,
                                 const int32_t* __restrict__ vals, uint64_t mask,
                                 int32_t* __restrict__ nbr_down, int32_t* __restrict__ nbr_up,
                                 uint32_t* __restrict__ mask_down, uint32_t* __restrict__ mask_up) {
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < M_in;
       r += (int64_t)gridDim.x * blockDim.x) {
    const int4 c = reinterpret_cast<const int4*>(indices)[r];
    const int p[3] = {c.y, c.z, c.w};
    uint32_t bits = 0;
    int kf = 0;
    for (int a = 0; a < g.k[0]; ++a)
      for (int b = 0; b < g.k[1]; ++b)
        for (int d = 0; d < g.k[2]; ++d, ++kf) {
          const int kk[3] = {a, b, d};
          int o[3];
          bool ok = true;
          for (int j = 0; j < 3; ++j) {
            const int t = p[j] + g.p[j] - kk[j];
            ok = ok && t >= 0 && (t % g.s[j]) == 0;
            o[j] = t / g.s[j];
            ok = ok && o[j] < g.shape_out[j];
          }
          if (!ok) continue;
          const int32_t orow =
              hash_lookup(keys, vals, mask, lin_key(c.x, o[0], o[1], o[2], g.shape_out));
          if (orow < 0) continue;  // cannot happen: every valid candidate is an output
          nbr_down[(int64_t)kf * M_out + orow] = (int32_t)r;
          nbr_up[(int64_t)kf * M_in + r] = orow;
          if (kf < 32) {
            bits |= (1u << kf);
            if (mask_down) atomicOr(mask_down + orow, 1u << kf);
          }
        }
    if (mask_up) mask_up[r] = bits;
  }
}

__device__ __forceinline__ uint32_t spread3(uint32_t x) {  // 8 bits -> every third bit
  x &= 0xffu;
  x = (x | (x << 16)) & 0x0300F00Fu;
  x = (x | (x << 8)) & 0x0300F00Fu;
  x = (x | (x << 4)) & 0x030C30C3u;
  x = (x | (x << 2)) & 0x09249249u;
  return x;
}

// sort key of a row: (batch, Morton code of its block of 2^bs voxels per side, set of active offsets).
// Rows of one spatial block become neighbours (their gathers hit the same L1/L2 lines), and inside a block
// rows with the same offset set are adjacent (whole 32-row MFMA slices skip inactive offsets).
__global__ void tile_key_kernel(const int32_t* __restrict__ indices, const uint32_t* __restrict__ mask,
                                int64_t M, int bs, uint64_t* __restrict__ keys, int32_t* __restrict__ iota) {
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < M;
       r += (int64_t)gridDim.x * blockDim.x) {
    const int4 c = reinterpret_cast<const int4*>(indices)[r];
    const uint32_t mort = spread3((uint32_t)c.y >> bs) | (spread3((uint32_t)c.z >> bs) << 1) |
                          (spread3((uint32_t)c.w >> bs) << 2);
    keys[r] = ((uint64_t)((uint32_t)c.x & 0xffu) << 56) | ((uint64_t)(mort & 0xffffffu) << 32) |
              (uint64_t)(mask ? mask[r] : 0u);
    iota[r] = (int32_t)r;
  }
}

// ---- slice scheduling.  A launch lasts as long as its most loaded CU, and the dispatcher hands workgroup i of a launch
// to CU i % n_cu while workgroups fit (measured, tools/conv2_stamps.py: exact for every workgroup that is resident from
// the start; later ones go wherever a slot frees).  The 32-row slices of the locality order -- the work items of the
// convolution kernels -- are therefore re-arranged by weight (number of kernel offsets any of their rows uses = steps of
// the slice), heaviest first, stable (equal weights keep the locality order).  The DEAL is the launch's: the forward /
// dIn kernel maps its workgroup index through a snake over bands of n_cu work items (spconv2.hip), the weight-gradient
// kernel deals its waves the same way (spconv_dw2.hip).  Round 5: the unit went from 128-row tiles (the round-1 kernel's
// work item) to slices, the snake from the order itself into the launches (an item is a slice x an output block: the
// bands of a launch depend on its channel count), and every level is scheduled (levels 2-4 were not: their most
// loaded CU carried 51 steps against a mean of 26).  WSIS_TILE_BAND = n > 0 puts the round-1 snake back into the order.
constexpr int SCHED_TM = 32;

__global__ __launch_bounds__(64) void tile_weight_kernel(const int32_t* __restrict__ order,
                                                         const uint32_t* __restrict__ mask, int64_t n_tiles,
                                                         uint32_t* __restrict__ keys, int32_t* __restrict__ ids) {
  const int64_t t = blockIdx.x;
  if (t >= n_tiles) return;
  const int lane = threadIdx.x;
  uint32_t m = lane < SCHED_TM ? mask[order[t * SCHED_TM + lane]] : 0u;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) m |= __shfl_xor(m, off, 64);
  if (lane == 0) {
    keys[t] = 32u - (uint32_t)__popc(m);     // ascending key = descending weight
    ids[t] = (int32_t)t;
  }
}

__global__ void tile_permute_kernel(const int32_t* __restrict__ order_in, const int32_t* __restrict__ sorted_tiles,
                                    int64_t M, int64_t n_tiles, int band, int32_t* __restrict__ order_out) {
  for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < M; p += (int64_t)gridDim.x * blockDim.x) {
    const int64_t slot = p / SCHED_TM;
    if (slot >= n_tiles) {       // the partial tile at the end keeps its place
      order_out[p] = order_in[p];
      continue;
    }
    // snake: slot -> rank in the weight order
    const int64_t b = slot / band, c = slot - b * band;
    int64_t rank = slot;
    if (b & 1) {
      const int64_t width = min((int64_t)band, n_tiles - b * band);
      rank = b * band + (width - 1 - c);
    }
    order_out[p] = order_in[(int64_t)sorted_tiles[rank] * SCHED_TM + (p - slot * SCHED_TM)];
  }
}

// ---- batched form: the tile orders of ALL gather tables of a pyramid from one sort.  Each table's keys carry the
// table number in their top 4 bits, so one stable sort of the concatenated keys leaves every table's order in its own
// contiguous segment (13 tables of the C2 pyramid: 13 x ~9 merge-sort launches of a few microseconds each -> ~11).
constexpr int TOB_MAX = 16;
struct TileBatch {
  const int32_t* indices[TOB_MAX];
  const uint32_t* mask[TOB_MAX];
  int64_t base[TOB_MAX + 1];      // first element of each table in the concatenated arrays
  int64_t tile_base[TOB_MAX + 1]; // first scheduled tile of each table (tables that are not scheduled: empty range)
  int band[TOB_MAX];
  int n;
  int bs;
};

__device__ __forceinline__ int batch_table_of(const int64_t* base, int n, int64_t e) {
  int t = 0;
#pragma unroll 1
  while (t + 1 < n && base[t + 1] <= e) ++t;
  return t;
}

__global__ void tile_key_batch_kernel(TileBatch b, uint64_t* __restrict__ keys, int32_t* __restrict__ iota) {
  const int64_t total = b.base[b.n];
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int t = batch_table_of(b.base, b.n, e);
    const int64_t r = e - b.base[t];
    const int4 c = reinterpret_cast<const int4*>(b.indices[t])[r];
    const uint32_t mort = spread3((uint32_t)c.y >> b.bs) | (spread3((uint32_t)c.z >> b.bs) << 1) |
                          (spread3((uint32_t)c.w >> b.bs) << 2);
    keys[e] = ((uint64_t)t << 60) | ((uint64_t)((uint32_t)c.x & 0xfu) << 56) | ((uint64_t)(mort & 0xffffffu) << 32) |
              (uint64_t)(b.mask[t] ? b.mask[t][r] : 0u);
    iota[e] = (int32_t)r;
  }
}

__global__ __launch_bounds__(64) void tile_weight_batch_kernel(TileBatch b, const int32_t* __restrict__ order,
                                                               uint32_t* __restrict__ keys,
                                                               int32_t* __restrict__ ids) {
  const int64_t g = blockIdx.x;
  if (g >= b.tile_base[b.n]) return;
  const int t = batch_table_of(b.tile_base, b.n, g);
  const int64_t tile = g - b.tile_base[t];
  const int32_t* ord = order + b.base[t];
  const uint32_t* mask = b.mask[t];
  const int lane = threadIdx.x;
  uint32_t m = lane < SCHED_TM ? mask[ord[tile * SCHED_TM + lane]] : 0u;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) m |= __shfl_xor(m, off, 64);
  if (lane == 0) {
    keys[g] = ((uint32_t)t << 6) | (32u - (uint32_t)__popc(m));   // table-major, then descending weight
    ids[g] = (int32_t)tile;
  }
}

__global__ void tile_permute_batch_kernel(TileBatch b, const int32_t* __restrict__ order_in,
                                          const int32_t* __restrict__ sorted_tiles,
                                          int32_t* __restrict__ order_out) {
  const int64_t total = b.base[b.n];
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int t = batch_table_of(b.base, b.n, e);
    const int64_t n_tiles = b.tile_base[t + 1] - b.tile_base[t];
    const int64_t p = e - b.base[t];
    const int64_t slot = p / SCHED_TM;
    if (slot >= n_tiles) continue;      // table not scheduled / partial tile at the end: order_out already holds it
    const int band = b.band[t];
    const int64_t bb = slot / band, c = slot - bb * band;
    int64_t rank = slot;
    if (bb & 1) {
      const int64_t width = min((int64_t)band, n_tiles - bb * band);
      rank = bb * band + (width - 1 - c);
    }
    order_out[e] = order_in[b.base[t] + (int64_t)sorted_tiles[b.tile_base[t] + rank] * SCHED_TM + (p - slot * SCHED_TM)];
  }
}

int sched_band() {      // snake period of the ORDER: none by default (the launches deal, see above)
  static int band = -1;
  if (band < 0) {
    const char* e = tune_env("WSIS_TILE_BAND");
    band = (e && atoi(e) > 0) ? atoi(e) : 0;
  }
  return band;
}

__global__ void iota_kernel(int32_t* __restrict__ p, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    p[i] = (int32_t)i;
}

int fill_geo(Geo& g, const int32_t* in_shape, const int32_t* out_shape, const int32_t* k,
             const int32_t* s, const int32_t* p) {
  for (int j = 0; j < 3; ++j) {
    g.shape_in[j] = in_shape ? in_shape[j] : 0;
    g.shape_out[j] = out_shape ? out_shape[j] : 0;
    g.k[j] = k ? k[j] : 1;
    g.s[j] = s ? s[j] : 1;
    g.p[j] = p ? p[j] : 0;
    if (g.k[j] < 1 || g.s[j] < 1 || g.p[j] < 0) return -1;
  }
  return 0;
}

bool is_pow2(int64_t x) { return x > 0 && (x & (x - 1)) == 0; }
inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

int bits_for(int64_t max_value) {
  int b = 1;
  while (b < 63 && ((int64_t)1 << b) <= max_value) ++b;
  return b;
}

// several 32-bit pattern fills in ONE launch (the tables and masks of a level were five hipMemsetAsync launches)
constexpr int FILL_MAX = 6;
struct FillBatch {
  uint32_t* p[FILL_MAX];
  uint64_t words[FILL_MAX];
  uint32_t v[FILL_MAX];
  int n = 0;
  void add(void* ptr, uint64_t bytes, uint32_t pattern) {
    if (bytes == 0) return;
    p[n] = static_cast<uint32_t*>(ptr);
    words[n] = bytes / 4;
    v[n] = pattern;
    ++n;
  }
};

__global__ void multi_fill_kernel(FillBatch b) {
  const int s = blockIdx.y;
  uint32_t* __restrict__ p = b.p[s];
  const uint64_t n = b.words[s];
  const uint32_t v = b.v[s];
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    p[i] = v;
}

int run_fills(const FillBatch& b, hipStream_t st) {
  if (b.n == 0) return WSIS_OK;
  uint64_t mx = 0;
  for (int i = 0; i < b.n; ++i) mx = b.words[i] > mx ? b.words[i] : mx;
  hipLaunchKernelGGL(multi_fill_kernel, dim3(grid_for((int64_t)((mx + 3) / 4), 256), b.n), dim3(256), 0, st, b);
  WSIS_LAUNCH_CHECK();
  return WSIS_OK;
}

}  // namespace

extern "C" {

int wsis_hash_build(const int32_t* d_indices, int64_t M, const int32_t* h_shape3, int64_t* d_keys,
                    int32_t* d_vals, int64_t cap, void* stream) {
  WSIS_REQUIRE(M >= 0 && h_shape3 && d_keys && d_vals, "bad args");
  WSIS_REQUIRE(is_pow2(cap) && cap >= 2 * M && cap >= 2, "cap must be a power of two >= 2*M");
  Geo g;
  WSIS_REQUIRE(fill_geo(g, h_shape3, nullptr, nullptr, nullptr, nullptr) == 0, "bad geometry");
  hipStream_t st = as_stream(stream);
  {
    FillBatch fb;
    fb.add(d_keys, sizeof(int64_t) * (uint64_t)cap, 0xFFFFFFFFu);
    fb.add(d_vals, sizeof(int32_t) * (uint64_t)cap, 0x7F7F7F7Fu);
    const int rc = run_fills(fb, st);
    if (rc != WSIS_OK) return rc;
  }
  if (M == 0) return WSIS_OK;
  WSIS_REQUIRE(d_indices, "null indices");
  hipLaunchKernelGGL(hash_insert_kernel, dim3(grid_for(M, 256)), dim3(256), 0, st, d_indices, M, g,
                     d_keys, d_vals, (uint64_t)(cap - 1));
  WSIS_LAUNCH_CHECK();
  return WSIS_OK;
}

int wsis_rulebook_subm(const int32_t* d_indices, int64_t M, const int32_t* h_shape3,
                       const int32_t* h_ksize3, const int32_t* h_pad3, const int64_t* d_keys,
                       const int32_t* d_vals, int64_t cap, int32_t* d_nbr, uint32_t* d_mask,
                       void* stream) {
  WSIS_REQUIRE(M >= 0 && h_shape3 && h_ksize3 && h_pad3, "bad args");
  if (M == 0) return WSIS_OK;
  WSIS_REQUIRE(d_indices && d_keys && d_vals && d_nbr && is_pow2(cap), "null pointer / bad cap");
  Geo g;
  WSIS_REQUIRE(fill_geo(g, h_shape3, h_shape3, h_ksize3, nullptr, h_pad3) == 0, "bad geometry");
  const bool k3 = g.k[0] == 3 && g.k[1] == 3 && g.k[2] == 3 && g.p[0] == 1 && g.p[1] == 1 && g.p[2] == 1 &&
                  cap <= ((int64_t)1 << 32);
  if (k3)
    hipLaunchKernelGGL(subm3_kernel, dim3(grid_for(M, 256)), dim3(256), 0, as_stream(stream), d_indices, M, g, d_keys,
                       d_vals, (uint64_t)(cap - 1), d_nbr, d_mask);
  else
    hipLaunchKernelGGL(subm_kernel, dim3(grid_for(M, 256)), dim3(256), 0, as_stream(stream), d_indices, M,
                       g, d_keys, d_vals, (uint64_t)(cap - 1), d_nbr, d_mask);
  WSIS_LAUNCH_CHECK();
  return WSIS_OK;
}

int64_t wsis_rulebook_down_ncand(int64_t M_in, const int32_t* k, const int32_t* s, const int32_t* p) {
  if (!k || !s || !p || M_in < 0) return -1;
  bool fast = true;
  for (int j = 0; j < 3; ++j) fast = fast && k[j] == s[j] && p[j] == 0;
  return fast ? M_in : M_in * (int64_t)k[0] * k[1] * k[2];
}

int64_t wsis_rulebook_down_workspace_bytes(int64_t n_cand) {
  if (n_cand < 0) return -1;
  if (n_cand == 0) return 256;
  size_t sort_bytes = 0, uniq_bytes = 0;
  int64_t* kp = nullptr;
  int32_t* cp = nullptr;
  if (rocprim::radix_sort_keys(nullptr, sort_bytes, kp, kp, (size_t)n_cand, 0, 64, (hipStream_t)0) !=
      hipSuccess)
    return -1;
  if (rocprim::unique(nullptr, uniq_bytes, kp, kp, cp, (size_t)n_cand, rocprim::equal_to<int64_t>(),
                      (hipStream_t)0) != hipSuccess)
    return -1;
  // layout: [sorted keys n_cand*8][temp max(sort,unique)]
  return (int64_t)(align256((size_t)n_cand * 8) + align256(sort_bytes > uniq_bytes ? sort_bytes : uniq_bytes) +
                   256);
}

namespace {
int down_keys_sorted(const int32_t* d_indices_in, int64_t M_in, const int32_t* h_in_shape3,
                     const int32_t* h_out_shape3, const int32_t* h_ksize3,
                     const int32_t* h_stride3, const int32_t* h_pad3, int64_t* d_cand,
                     int64_t* d_out_keys, int32_t* d_count, void* d_ws, int64_t ws_bytes,
                     void* stream, int32_t batch_limit);
}
int wsis_rulebook_down_keys(const int32_t* d_indices_in, int64_t M_in, const int32_t* h_in_shape3,
                            const int32_t* h_out_shape3, const int32_t* h_ksize3,
                            const int32_t* h_stride3, const int32_t* h_pad3, int64_t* d_cand,
                            int64_t* d_out_keys, int32_t* d_count, void* d_ws, int64_t ws_bytes,
                            void* stream) {
  // (no batch size in this entry point's signature: every row is a candidate; callers validate the batch column)
  return down_keys_sorted(d_indices_in, M_in, h_in_shape3, h_out_shape3, h_ksize3, h_stride3, h_pad3, d_cand, d_out_keys,
                          d_count, d_ws, ws_bytes, stream, 0);
}
namespace {
int down_keys_sorted(const int32_t* d_indices_in, int64_t M_in, const int32_t* h_in_shape3,
                     const int32_t* h_out_shape3, const int32_t* h_ksize3,
                     const int32_t* h_stride3, const int32_t* h_pad3, int64_t* d_cand,
                     int64_t* d_out_keys, int32_t* d_count, void* d_ws, int64_t ws_bytes,
                     void* stream, int32_t batch_limit) {
  WSIS_REQUIRE(M_in >= 0 && h_in_shape3 && h_out_shape3 && h_ksize3 && h_stride3 && h_pad3 && d_count,
               "bad args");
  hipStream_t st = as_stream(stream);
  WSIS_HIP_CHECK(hipMemsetAsync(d_count, 0, sizeof(int32_t), st));
  if (M_in == 0) return WSIS_OK;
  WSIS_REQUIRE(d_indices_in && d_cand && d_out_keys && d_ws, "null pointer");
  Geo g;
  WSIS_REQUIRE(fill_geo(g, h_in_shape3, h_out_shape3, h_ksize3, h_stride3, h_pad3) == 0, "bad geometry");
  const int64_t n_cand = wsis_rulebook_down_ncand(M_in, h_ksize3, h_stride3, h_pad3);
  bool fast_b = true;
  for (int j = 0; j < 3; ++j) fast_b = fast_b && h_ksize3[j] == h_stride3[j] && h_pad3[j] == 0;
  const int fast = fast_b ? 1 : 0;
  // invalid candidates carry the largest int64 so they sort behind every real linear index
  const int64_t invalid = INT64_MAX;
  hipLaunchKernelGGL(down_cand_kernel, dim3(grid_for(M_in, 256)), dim3(256), 0, st, d_indices_in, M_in, g,
                     fast, invalid, d_cand, batch_limit);
  WSIS_LAUNCH_CHECK();
  char* ws = static_cast<char*>(d_ws);
  int64_t* d_sorted = reinterpret_cast<int64_t*>(ws);
  size_t off = align256((size_t)n_cand * 8);
  WSIS_REQUIRE((int64_t)off < ws_bytes, "workspace too small");
  void* d_temp = ws + off;
  size_t temp_bytes = (size_t)ws_bytes - off;
  size_t need = 0;
  WSIS_HIP_CHECK(rocprim::radix_sort_keys(nullptr, need, d_cand, d_sorted, (size_t)n_cand, 0, 64, st));
  WSIS_REQUIRE(need <= temp_bytes, "workspace too small for sort");
  const int end_bit = 64;
  WSIS_HIP_CHECK(rocprim::radix_sort_keys(d_temp, temp_bytes, d_cand, d_sorted, (size_t)n_cand, 0,
                                          end_bit, st));
  need = 0;
  WSIS_HIP_CHECK(rocprim::unique(nullptr, need, d_sorted, d_out_keys, d_count, (size_t)n_cand,
                                 rocprim::equal_to<int64_t>(), st));
  WSIS_REQUIRE(need <= temp_bytes, "workspace too small for unique");
  WSIS_HIP_CHECK(rocprim::unique(d_temp, temp_bytes, d_sorted, d_out_keys, d_count, (size_t)n_cand,
                                 rocprim::equal_to<int64_t>(), st));
  hipLaunchKernelGGL(down_fix_count_kernel, dim3(1), dim3(64), 0, st, d_out_keys, invalid, d_count);
  WSIS_LAUNCH_CHECK();
  return WSIS_OK;
}
}  // namespace

namespace {
// bitmap form of wsis_rulebook_down_keys (see down_mark_kernel); false: does not apply, nothing was issued
bool down_keys_bitmap(const int32_t* d_indices_in, int64_t M_in, const int32_t* h_in_shape3, const int32_t* h_out_shape3,
                      const int32_t* k, const int32_t* s3, const int32_t* p, int32_t batch_size, int64_t* d_out_keys,
                      int32_t* d_count, void* d_ws, int64_t ws_bytes, hipStream_t st, int* rc_out) {
  *rc_out = WSIS_OK;
  for (int j = 0; j < 3; ++j)
    if (k[j] != s3[j] || p[j] != 0) return false;
  if (batch_size < 1 || M_in < 1 || !d_ws) return false;
  const int64_t n_cells = (int64_t)batch_size * h_out_shape3[0] * h_out_shape3[1] * h_out_shape3[2];
  if (n_cells < 1 || n_cells > ((int64_t)1 << 31)) return false;
  const int64_t n_words = (n_cells + 31) >> 5;
  const int64_t n_chunks = (n_words + BM_WORDS_PER_WG - 1) / BM_WORDS_PER_WG;
  const int64_t need = (int64_t)align256((size_t)n_words * 4) + (int64_t)align256((size_t)n_chunks * 4);
  if (need > ws_bytes || n_chunks > (1 << 22)) return false;
  Geo g;
  if (fill_geo(g, h_in_shape3, h_out_shape3, k, s3, p) != 0) return false;
  uint32_t* bitmap = static_cast<uint32_t*>(d_ws);
  int32_t* chunk = reinterpret_cast<int32_t*>(static_cast<char*>(d_ws) + align256((size_t)n_words * 4));
  auto fail_hip = [&](hipError_t e) {
    *rc_out = fail(WSIS_ERR_HIP, "down_keys_bitmap: %s", hipGetErrorString(e));
    return true;
  };
  hipError_t e = hipMemsetAsync(bitmap, 0, (size_t)n_words * 4, st);
  if (e != hipSuccess) return fail_hip(e);
  hipLaunchKernelGGL(down_mark_kernel, dim3(grid_for(M_in, 256)), dim3(256), 0, st, d_indices_in, M_in, g, n_cells, bitmap);
  hipLaunchKernelGGL(bitmap_count_kernel, dim3((unsigned)n_chunks), dim3(256), 0, st, bitmap, n_words, chunk);
  hipLaunchKernelGGL(bitmap_scan_kernel, dim3(1), dim3(256), 0, st, chunk, (int)n_chunks, d_count);
  hipLaunchKernelGGL(bitmap_emit_kernel, dim3((unsigned)n_chunks), dim3(256), 0, st, bitmap, n_words, chunk, d_out_keys);
  e = hipGetLastError();
  if (e != hipSuccess) return fail_hip(e);
  return true;
}
}  // namespace

int wsis_rulebook_down_fill(const int32_t* d_indices_in, int64_t M_in, const int32_t* h_in_shape3,
                            const int32_t* h_out_shape3, const int32_t* h_ksize3,
                            const int32_t* h_stride3, const int32_t* h_pad3,
                            const int64_t* d_out_keys, int64_t M_out, int32_t* d_indices_out,
                            int64_t* d_keys, int32_t* d_vals, int64_t cap, int32_t* d_nbr_down,
                            int32_t* d_nbr_up, uint32_t* d_mask_down, uint32_t* d_mask_up,
                            void* stream) {
  WSIS_REQUIRE(M_in >= 0 && M_out >= 0 && h_in_shape3 && h_out_shape3 && h_ksize3 && h_stride3 && h_pad3,
               "bad args");
  WSIS_REQUIRE(d_keys && d_vals && is_pow2(cap) && cap >= 2 * M_out && cap >= 2,
               "cap must be a power of two >= 2*M_out");
  Geo g;
  WSIS_REQUIRE(fill_geo(g, h_in_shape3, h_out_shape3, h_ksize3, h_stride3, h_pad3) == 0, "bad geometry");
  const int K = g.k[0] * g.k[1] * g.k[2];
  hipStream_t st = as_stream(stream);
  if (M_out > 0) WSIS_REQUIRE(d_out_keys && d_indices_out && d_nbr_down, "null pointer");
  if (M_in > 0) WSIS_REQUIRE(d_indices_in && d_nbr_up, "null pointer");
  {
    FillBatch fb;
    fb.add(d_keys, sizeof(int64_t) * (uint64_t)cap, 0xFFFFFFFFu);
    fb.add(d_vals, sizeof(int32_t) * (uint64_t)cap, 0x7F7F7F7Fu);
    if (M_out > 0) {
      fb.add(d_nbr_down, sizeof(int32_t) * (uint64_t)(K * M_out), 0xFFFFFFFFu);
      if (d_mask_down) fb.add(d_mask_down, sizeof(uint32_t) * (uint64_t)M_out, 0u);
    }
    if (M_in > 0) {
      fb.add(d_nbr_up, sizeof(int32_t) * (uint64_t)(K * M_in), 0xFFFFFFFFu);
      if (M_out == 0 && d_mask_up) fb.add(d_mask_up, sizeof(uint32_t) * (uint64_t)M_in, 0u);
    }
    const int rc = run_fills(fb, st);
    if (rc != WSIS_OK) return rc;
  }
  if (M_out > 0) {
    hipLaunchKernelGGL(down_decode_kernel, dim3(grid_for(M_out, 256)), dim3(256), 0, st, d_out_keys, M_out,
                       g, d_indices_out, d_keys, d_vals, (uint64_t)(cap - 1));
    WSIS_LAUNCH_CHECK();
  }
  if (M_in > 0 && M_out > 0) {
    hipLaunchKernelGGL(down_fill_kernel, dim3(grid_for(M_in, 256)), dim3(256), 0, st, d_indices_in, M_in,
                       M_out, g, d_keys, d_vals, (uint64_t)(cap - 1), d_nbr_down, d_nbr_up, d_mask_down,
                       d_mask_up);
    WSIS_LAUNCH_CHECK();
  }
  return WSIS_OK;
}

int64_t wsis_tile_order_workspace_bytes(int64_t M) {
  if (M < 0) return -1;
  if (M == 0) return 256;
  size_t sort_bytes = 0;
  uint64_t* kp = nullptr;
  int32_t* vp = nullptr;
  if (rocprim::radix_sort_pairs(nullptr, sort_bytes, kp, kp, vp, vp, (size_t)M, 0, 64, (hipStream_t)0) !=
      hipSuccess)
    return -1;
  // layout: [keys M*8][keys_out M*8][iota M*4][temp]; the tile scheduling reuses the two key arrays afterwards
  return (int64_t)(2 * align256((size_t)M * 8) + align256((size_t)M * 4) + align256(sort_bytes) + 256);
}

int wsis_tile_order(const int32_t* d_indices, const uint32_t* d_mask, int64_t M, int32_t block_shift,
                    int32_t* d_order, void* d_ws, int64_t ws_bytes, void* stream) {
  WSIS_REQUIRE(M >= 0 && block_shift >= 0 && block_shift < 16, "bad args");
  if (M == 0) return WSIS_OK;
  WSIS_REQUIRE(d_indices && d_order && d_ws, "null pointer");
  hipStream_t st = as_stream(stream);
  char* ws = static_cast<char*>(d_ws);
  const size_t a8 = align256((size_t)M * 8), a4 = align256((size_t)M * 4);
  WSIS_REQUIRE((int64_t)(2 * a8 + a4) < ws_bytes, "workspace too small");
  uint64_t* keys = reinterpret_cast<uint64_t*>(ws);
  uint64_t* keys_out = reinterpret_cast<uint64_t*>(ws + a8);
  int32_t* iota = reinterpret_cast<int32_t*>(ws + 2 * a8);
  void* temp = ws + 2 * a8 + a4;
  size_t temp_bytes = (size_t)ws_bytes - (2 * a8 + a4);
  hipLaunchKernelGGL(tile_key_kernel, dim3(grid_for(M, 256)), dim3(256), 0, st, d_indices, d_mask, M,
                     (int)block_shift, keys, iota);
  WSIS_LAUNCH_CHECK();
  size_t need = 0;
  WSIS_HIP_CHECK(rocprim::radix_sort_pairs(nullptr, need, keys, keys_out, iota, d_order, (size_t)M, 0, 64, st));
  WSIS_REQUIRE(need <= temp_bytes, "workspace too small for sort");
  WSIS_HIP_CHECK(rocprim::radix_sort_pairs(temp, temp_bytes, keys, keys_out, iota, d_order, (size_t)M, 0, 64, st));

  // ---- tile scheduling (WSIS_TILE_SCHED=0 switches it off; levels below WSIS_TILE_SCHED_MIN full tiles keep the
  //      plain locality order: their offsets are split over blockIdx.z and the dispatch pattern differs)
  static int sched = -1, sched_min = 4;
  if (sched < 0) {
    const char* e = tune_env("WSIS_TILE_SCHED");
    sched = e ? atoi(e) : 1;
    e = tune_env("WSIS_TILE_SCHED_MIN");
    if (e) sched_min = atoi(e);
  }
  const int64_t n_tiles = M / SCHED_TM;
  if (sched && d_mask && n_tiles >= sched_min) {
    // the sort is done: keys (M*8 bytes) holds the old order copy + tile arrays, keys_out the tile sort scratch
    int32_t* ord0 = reinterpret_cast<int32_t*>(keys);                       // M ints (<= half of the region)
    uint32_t* tkeys = reinterpret_cast<uint32_t*>(keys_out);                // n_tiles each, 4 arrays
    uint32_t* tkeys_out = tkeys + n_tiles;
    int32_t* tids = reinterpret_cast<int32_t*>(tkeys_out + n_tiles);
    int32_t* tsorted = tids + n_tiles;
    WSIS_REQUIRE((size_t)n_tiles * 16 <= a8, "workspace too small for the tile schedule");
    WSIS_HIP_CHECK(hipMemcpyAsync(ord0, d_order, (size_t)M * 4, hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(tile_weight_kernel, dim3((unsigned)n_tiles), dim3(64), 0, st, ord0, d_mask, n_tiles, tkeys,
                       tids);
    WSIS_LAUNCH_CHECK();
    size_t need2 = 0;
    WSIS_HIP_CHECK(rocprim::radix_sort_pairs(nullptr, need2, tkeys, tkeys_out, tids, tsorted, (size_t)n_tiles, 0, 6, st));
    WSIS_REQUIRE(need2 <= temp_bytes, "workspace too small for the tile sort");
    WSIS_HIP_CHECK(rocprim::radix_sort_pairs(temp, temp_bytes, tkeys, tkeys_out, tids, tsorted, (size_t)n_tiles, 0, 6, st));
    // Snake period = distance, in tiles, between the consecutive workgroups of one CU.  One workgroup per tile
    // (n_tiles >= n_cu, no offset split): n_cu.  Fewer tiles than CUs: the launch splits the offsets over blockIdx.z
    // with workgroup id = tile + n_tiles * z, so CU c holds tiles c, c + n_cu mod n_tiles, ... of successive slices.
    const int n_cu = sched_band();
    const int64_t grid_tiles = (M + SCHED_TM - 1) / SCHED_TM;
    int band = n_cu <= 0 ? 0x7fffffff : grid_tiles >= n_cu ? n_cu : (int)(n_cu % grid_tiles);
    if (band == 0) band = (int)grid_tiles;
    hipLaunchKernelGGL(tile_permute_kernel, dim3(grid_for(M, 256)), dim3(256), 0, st, ord0, tsorted, M, n_tiles,
                       band, d_order);
    WSIS_LAUNCH_CHECK();
  }
  return WSIS_OK;
}

int64_t wsis_tile_order_batch_workspace_bytes(int64_t N) {
  if (N < 0) return -1;
  if (N == 0) return 256;
  size_t sort_bytes = 0;
  uint64_t* kp = nullptr;
  int32_t* vp = nullptr;
  if (rocprim::radix_sort_pairs(nullptr, sort_bytes, kp, kp, vp, vp, (size_t)N, 0, 64, (hipStream_t)0) !=
      hipSuccess)
    return -1;
  // layout: [keys N*8][keys_out N*8][iota N*4][temp]; the tile schedule reuses keys / keys_out after the sort
  return (int64_t)(2 * align256((size_t)N * 8) + align256((size_t)N * 4) + align256(sort_bytes) + 256);
}

int wsis_tile_order_batch(int32_t n, const void* const* h_indices, const void* const* h_mask, const int64_t* h_M,
                          int32_t block_shift, int32_t batch_size, int32_t* d_order_all, void* d_ws,
                          int64_t ws_bytes, void* stream) {
  WSIS_REQUIRE(n >= 0 && n <= TOB_MAX && block_shift >= 0 && block_shift < 16, "bad args (at most 16 tables)");
  WSIS_REQUIRE(batch_size >= 1 && batch_size <= 16, "the batched tile order packs the batch index into 4 bits");
  if (n == 0) return WSIS_OK;
  WSIS_REQUIRE(h_indices && h_mask && h_M, "null pointer");
  TileBatch b;
  b.n = n;
  b.bs = block_shift;
  b.base[0] = 0;
  for (int t = 0; t < n; ++t) {
    WSIS_REQUIRE(h_M[t] >= 0 && (h_M[t] == 0 || h_indices[t]), "bad table");
    b.indices[t] = static_cast<const int32_t*>(h_indices[t]);
    b.mask[t] = static_cast<const uint32_t*>(h_mask[t]);
    b.base[t + 1] = b.base[t] + h_M[t];
  }
  const int64_t N = b.base[n];
  if (N == 0) return WSIS_OK;
  WSIS_REQUIRE(d_order_all && d_ws, "null pointer");
  hipStream_t st = as_stream(stream);
  char* ws = static_cast<char*>(d_ws);
  const size_t a8 = align256((size_t)N * 8), a4 = align256((size_t)N * 4);
  WSIS_REQUIRE((int64_t)(2 * a8 + a4) < ws_bytes, "workspace too small");
  uint64_t* keys = reinterpret_cast<uint64_t*>(ws);
  uint64_t* keys_out = reinterpret_cast<uint64_t*>(ws + a8);
  int32_t* iota = reinterpret_cast<int32_t*>(ws + 2 * a8);
  void* temp = ws + 2 * a8 + a4;
  size_t temp_bytes = (size_t)ws_bytes - (2 * a8 + a4);
  static int sched = -1, sched_min = 4;
  if (sched < 0) {
    const char* e = tune_env("WSIS_TILE_SCHED");
    sched = e ? atoi(e) : 1;
    e = tune_env("WSIS_TILE_SCHED_MIN");
    if (e) sched_min = atoi(e);
  }
  const int n_cu = sched_band();
  b.tile_base[0] = 0;
  for (int t = 0; t < n; ++t) {
    const int64_t n_tiles = h_M[t] / SCHED_TM;
    const bool on = sched && b.mask[t] && n_tiles >= sched_min;
    b.tile_base[t + 1] = b.tile_base[t] + (on ? n_tiles : 0);
    const int64_t grid_tiles = (h_M[t] + SCHED_TM - 1) / SCHED_TM;
    int band = n_cu <= 0 ? 0x7fffffff : grid_tiles >= n_cu ? n_cu : (grid_tiles > 0 ? (int)(n_cu % grid_tiles) : 1);
    if (band == 0) band = (int)grid_tiles;
    b.band[t] = band > 0 ? band : 1;
  }
  hipLaunchKernelGGL(tile_key_batch_kernel, dim3(grid_for(N, 256)), dim3(256), 0, st, b, keys, iota);
  WSIS_LAUNCH_CHECK();
  size_t need = 0;
  WSIS_HIP_CHECK(rocprim::radix_sort_pairs(nullptr, need, keys, keys_out, iota, d_order_all, (size_t)N, 0, 64, st));
  WSIS_REQUIRE(need <= temp_bytes, "workspace too small for sort");
  WSIS_HIP_CHECK(rocprim::radix_sort_pairs(temp, temp_bytes, keys, keys_out, iota, d_order_all, (size_t)N, 0, 64, st));
  const int64_t T = b.tile_base[n];
  if (T > 0) {
    int32_t* ord0 = reinterpret_cast<int32_t*>(keys);                       // N ints
    uint32_t* tkeys = reinterpret_cast<uint32_t*>(keys_out);                // T each, 4 arrays
    uint32_t* tkeys_out = tkeys + T;
    int32_t* tids = reinterpret_cast<int32_t*>(tkeys_out + T);
    int32_t* tsorted = tids + T;
    WSIS_REQUIRE((size_t)T * 16 <= a8, "workspace too small for the tile schedule");
    WSIS_HIP_CHECK(hipMemcpyAsync(ord0, d_order_all, (size_t)N * 4, hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(tile_weight_batch_kernel, dim3((unsigned)T), dim3(64), 0, st, b, ord0, tkeys, tids);
    WSIS_LAUNCH_CHECK();
    size_t need2 = 0;
    WSIS_HIP_CHECK(rocprim::radix_sort_pairs(nullptr, need2, tkeys, tkeys_out, tids, tsorted, (size_t)T, 0, 10, st));
    WSIS_REQUIRE(need2 <= temp_bytes, "workspace too small for the tile sort");
    WSIS_HIP_CHECK(rocprim::radix_sort_pairs(temp, temp_bytes, tkeys, tkeys_out, tids, tsorted, (size_t)T, 0, 10, st));
    hipLaunchKernelGGL(tile_permute_batch_kernel, dim3(grid_for(N, 256)), dim3(256), 0, st, b, ord0, tsorted,
                       d_order_all);
    WSIS_LAUNCH_CHECK();
  }
  return WSIS_OK;
}

int64_t wsis_mask_order_workspace_bytes(int64_t M) {
  if (M < 0) return -1;
  if (M == 0) return 256;
  size_t sort_bytes = 0;
  uint32_t* kp = nullptr;
  int32_t* vp = nullptr;
  if (rocprim::radix_sort_pairs(nullptr, sort_bytes, kp, kp, vp, vp, (size_t)M, 0, 32, (hipStream_t)0) !=
      hipSuccess)
    return -1;
  // layout: [keys_out M*4][iota M*4][temp]
  return (int64_t)(2 * align256((size_t)M * 4) + align256(sort_bytes) + 256);
}

int wsis_mask_order(const uint32_t* d_mask, int64_t M, int32_t* d_order, void* d_ws, int64_t ws_bytes,
                    void* stream) {
  WSIS_REQUIRE(M >= 0, "bad M");
  if (M == 0) return WSIS_OK;
  WSIS_REQUIRE(d_mask && d_order && d_ws, "null pointer");
  hipStream_t st = as_stream(stream);
  char* ws = static_cast<char*>(d_ws);
  const size_t a = align256((size_t)M * 4);
  WSIS_REQUIRE((int64_t)(2 * a) < ws_bytes, "workspace too small");
  uint32_t* keys_out = reinterpret_cast<uint32_t*>(ws);
  int32_t* iota = reinterpret_cast<int32_t*>(ws + a);
  void* temp = ws + 2 * a;
  size_t temp_bytes = (size_t)ws_bytes - 2 * a;
  hipLaunchKernelGGL(iota_kernel, dim3(grid_for(M, 256)), dim3(256), 0, st, iota, M);
  WSIS_LAUNCH_CHECK();
  size_t need = 0;
  WSIS_HIP_CHECK(rocprim::radix_sort_pairs(nullptr, need, d_mask, keys_out, iota, d_order, (size_t)M, 0, 32, st));
  WSIS_REQUIRE(need <= temp_bytes, "workspace too small for sort");
  WSIS_HIP_CHECK(rocprim::radix_sort_pairs(temp, temp_bytes, d_mask, keys_out, iota, d_order, (size_t)M, 0,
                                           32, st));
  return WSIS_OK;
}


// ---- the whole rulebook pyramid of a UBlock from ONE call -------------------------------------------------------
// SubM k3 p1 table per level + SparseConv3d k2 s2 tables between levels (sparse_unet3d.py:130,261,292; [UPSTREAM
// spconv getIndicePair]) when the row count of every level is known on the host (a loader has the level-0 coordinates
// there: spconv.ops.level_voxel_counts) -- the chain then needs no device read-back, and issuing it from here instead
// of ~25 Python-dispatched calls with ~20 tensor allocations takes the rulebook build out of the issuing thread's
// budget (~1 ms of a 10 ms step).  The SAME entry points in the SAME order as the per-table path (wsis_hash_build,
// wsis_rulebook_subm, wsis_rulebook_down_keys / _fill, wsis_tile_order_batch, wsis_rulebook_pack_batch): identical
// tables.  Everything lives in one caller-allocated arena; the layout (byte offsets per level) is a pure function of
// the row counts.  The device's own output counts stay in the arena (WSIS_PYR_COUNT) for the caller to check.
namespace {
inline int64_t pyr_up(int64_t v) { return (v + 255) / 256 * 256; }
inline int64_t pow2_cap(int64_t m) {
  int64_t cap = 16;
  while (cap < 2 * m) cap <<= 1;
  return cap;
}
struct PyrLayout {
  int64_t off[8][WSIS_PYR_FIELDS];
  int64_t tile_ws, tile_ws_bytes, order_all, total;
};
int pyr_layout(int64_t M0, const int64_t* counts, int n_levels, PyrLayout& L) {
  if (n_levels < 1 || n_levels > 8 || M0 < 0) return -1;
  int64_t M[8];
  M[0] = M0;
  for (int l = 1; l < n_levels; ++l) {
    if (counts[l - 1] < 0) return -1;
    M[l] = counts[l - 1];
  }
  int64_t cur = 0, n_all = 0;
  auto take = [&](int64_t bytes) {
    const int64_t o = cur;
    cur += pyr_up(bytes > 0 ? bytes : 1);
    return o;
  };
  for (int l = 0; l < n_levels; ++l) {
    int64_t* o = L.off[l];
    for (int f = 0; f < WSIS_PYR_FIELDS; ++f) o[f] = -1;
    const int64_t cap = pow2_cap(M[l]);
    o[WSIS_PYR_ROWS] = M[l];
    o[WSIS_PYR_CAP] = cap;
    o[WSIS_PYR_INDICES] = l == 0 ? -1 : take(M[l] * 16);
    o[WSIS_PYR_KEYS] = take(cap * 8);
    o[WSIS_PYR_VALS] = take(cap * 4);
    o[WSIS_PYR_SUBM_NBR] = take(27 * M[l] * 4);
    o[WSIS_PYR_SUBM_MASK] = take(M[l] * 4);
    o[WSIS_PYR_SUBM_NBR_P] = take(27 * M[l] * 4);
    n_all += M[l];
    if (l + 1 < n_levels) {
      const int64_t n_cand = M[l];               // k == s, p == 0: one candidate per input voxel
      o[WSIS_PYR_CAND] = take(n_cand * 8);
      o[WSIS_PYR_OUT_KEYS] = take(n_cand * 8);
      o[WSIS_PYR_COUNT] = take(4);
      o[WSIS_PYR_DOWN_WS] = take(wsis_rulebook_down_workspace_bytes(n_cand));
      o[WSIS_PYR_DOWN_NBR] = take(8 * M[l + 1] * 4);
      o[WSIS_PYR_UP_NBR] = take(8 * M[l] * 4);
      o[WSIS_PYR_DOWN_MASK] = take(M[l + 1] * 4);
      o[WSIS_PYR_UP_MASK] = take(M[l] * 4);
      o[WSIS_PYR_DOWN_NBR_P] = take(8 * M[l + 1] * 4);
      o[WSIS_PYR_UP_NBR_P] = take(8 * M[l] * 4);
      n_all += M[l + 1] + M[l];
    }
  }
  L.order_all = take(n_all * 4);
  L.tile_ws_bytes = wsis_tile_order_batch_workspace_bytes(n_all);
  L.tile_ws = take(L.tile_ws_bytes);
  // tile orders: segments of order_all in the order subm(l), down(l), up(l), subm(l+1), ...
  int64_t seg = 0;
  for (int l = 0; l < n_levels; ++l) {
    L.off[l][WSIS_PYR_SUBM_ORDER] = L.order_all + seg * 4;
    seg += M[l];
    if (l + 1 < n_levels) {
      L.off[l][WSIS_PYR_DOWN_ORDER] = L.order_all + seg * 4;
      seg += M[l + 1];
      L.off[l][WSIS_PYR_UP_ORDER] = L.order_all + seg * 4;
      seg += M[l];
    }
  }
  L.total = cur;
  return 0;
}
}  // namespace

int64_t wsis_rulebook_pyramid_layout(int64_t M0, const int64_t* h_counts, int32_t n_levels, int64_t* h_layout) {
  PyrLayout L;
  if ((n_levels > 1 && !h_counts) || pyr_layout(M0, h_counts, n_levels, L) != 0) return -1;
  if (h_layout)
    for (int l = 0; l < n_levels; ++l)
      for (int f = 0; f < WSIS_PYR_FIELDS; ++f) h_layout[l * WSIS_PYR_FIELDS + f] = L.off[l][f];
  return L.total;
}

int wsis_rulebook_pyramid(const int32_t* d_indices0, int64_t M0, const int32_t* h_shape3, const int64_t* h_counts,
                          int32_t n_levels, int32_t batch_size, int32_t block_shift, void* d_arena, int64_t arena_bytes,
                          void* stream) {
  WSIS_REQUIRE(h_shape3 && d_arena && (M0 == 0 || d_indices0), "null pointer");
  WSIS_REQUIRE(n_levels <= 5 || n_levels * 3 - 2 <= 16, "at most 16 gather tables per batched tile order");
  PyrLayout L;
  WSIS_REQUIRE((n_levels <= 1 || h_counts) && pyr_layout(M0, h_counts, n_levels, L) == 0, "bad level counts");
  WSIS_REQUIRE(arena_bytes >= L.total && (reinterpret_cast<uintptr_t>(d_arena) & 255) == 0, "arena too small or unaligned");
  char* A = static_cast<char*>(d_arena);
  auto at = [&](int l, int f) -> void* { return L.off[l][f] < 0 ? nullptr : A + L.off[l][f]; };
  const int32_t k3[3] = {3, 3, 3}, p1[3] = {1, 1, 1}, k2[3] = {2, 2, 2}, s2[3] = {2, 2, 2}, p0[3] = {0, 0, 0};
  int32_t shape[3] = {h_shape3[0], h_shape3[1], h_shape3[2]};
  const void* t_idx[16];
  const void* t_mask[16];
  int64_t t_M[16];
  const void* pk_nbr[16];
  const void* pk_order[16];
  void* pk_out[16];
  int64_t pk_M[16];
  int32_t pk_K[16];
  int nt = 0;
  const int32_t* indices = d_indices0;
  for (int l = 0; l < n_levels; ++l) {
    const int64_t M = L.off[l][WSIS_PYR_ROWS], cap = L.off[l][WSIS_PYR_CAP];
    int64_t* keys = static_cast<int64_t*>(at(l, WSIS_PYR_KEYS));
    int32_t* vals = static_cast<int32_t*>(at(l, WSIS_PYR_VALS));
    int rc;
    if (l == 0) {        // deeper levels get their hash from the strided build of the level above
      rc = wsis_hash_build(indices, M, shape, keys, vals, cap, stream);
      if (rc != WSIS_OK) return rc;
    }
    rc = wsis_rulebook_subm(indices, M, shape, k3, p1, keys, vals, cap, static_cast<int32_t*>(at(l, WSIS_PYR_SUBM_NBR)),
                            static_cast<uint32_t*>(at(l, WSIS_PYR_SUBM_MASK)), stream);
    if (rc != WSIS_OK) return rc;
    t_idx[nt] = indices;
    t_mask[nt] = at(l, WSIS_PYR_SUBM_MASK);
    t_M[nt] = M;
    pk_nbr[nt] = at(l, WSIS_PYR_SUBM_NBR);
    pk_order[nt] = at(l, WSIS_PYR_SUBM_ORDER);
    pk_out[nt] = at(l, WSIS_PYR_SUBM_NBR_P);
    pk_M[nt] = M;
    pk_K[nt] = 27;
    ++nt;
    if (l + 1 == n_levels) break;
    const int64_t M_out = L.off[l + 1][WSIS_PYR_ROWS];
    int32_t out_shape[3];
    for (int j = 0; j < 3; ++j) out_shape[j] = (shape[j] - 2) / 2 + 1;
    // output cells: bitmap + prefix popcount where it applies (5 launches), candidate keys + sort + unique otherwise (13)
    if (!down_keys_bitmap(indices, M, shape, out_shape, k2, s2, p0, batch_size, static_cast<int64_t*>(at(l, WSIS_PYR_OUT_KEYS)),
                          static_cast<int32_t*>(at(l, WSIS_PYR_COUNT)), at(l, WSIS_PYR_DOWN_WS),
                          wsis_rulebook_down_workspace_bytes(M), as_stream(stream), &rc))
      // (same rows as the bitmap form: batch indices outside [0, batch_size) are not part of the build -- the device
      // count then differs from the caller's hint, which is what the caller checks)
      rc = down_keys_sorted(indices, M, shape, out_shape, k2, s2, p0, static_cast<int64_t*>(at(l, WSIS_PYR_CAND)),
                            static_cast<int64_t*>(at(l, WSIS_PYR_OUT_KEYS)), static_cast<int32_t*>(at(l, WSIS_PYR_COUNT)),
                            at(l, WSIS_PYR_DOWN_WS), wsis_rulebook_down_workspace_bytes(M), stream, batch_size);
    if (rc != WSIS_OK) return rc;
    int32_t* idx_out = static_cast<int32_t*>(at(l + 1, WSIS_PYR_INDICES));
    rc = wsis_rulebook_down_fill(indices, M, shape, out_shape, k2, s2, p0, static_cast<const int64_t*>(at(l, WSIS_PYR_OUT_KEYS)),
                                 M_out, idx_out, static_cast<int64_t*>(at(l + 1, WSIS_PYR_KEYS)),
                                 static_cast<int32_t*>(at(l + 1, WSIS_PYR_VALS)), L.off[l + 1][WSIS_PYR_CAP],
                                 static_cast<int32_t*>(at(l, WSIS_PYR_DOWN_NBR)), static_cast<int32_t*>(at(l, WSIS_PYR_UP_NBR)),
                                 static_cast<uint32_t*>(at(l, WSIS_PYR_DOWN_MASK)), static_cast<uint32_t*>(at(l, WSIS_PYR_UP_MASK)),
                                 stream);
    if (rc != WSIS_OK) return rc;
    t_idx[nt] = idx_out;
    t_mask[nt] = at(l, WSIS_PYR_DOWN_MASK);
    t_M[nt] = M_out;
    pk_nbr[nt] = at(l, WSIS_PYR_DOWN_NBR);
    pk_order[nt] = at(l, WSIS_PYR_DOWN_ORDER);
    pk_out[nt] = at(l, WSIS_PYR_DOWN_NBR_P);
    pk_M[nt] = M_out;
    pk_K[nt] = 8;
    ++nt;
    t_idx[nt] = indices;
    t_mask[nt] = at(l, WSIS_PYR_UP_MASK);
    t_M[nt] = M;
    pk_nbr[nt] = at(l, WSIS_PYR_UP_NBR);
    pk_order[nt] = at(l, WSIS_PYR_UP_ORDER);
    pk_out[nt] = at(l, WSIS_PYR_UP_NBR_P);
    pk_M[nt] = M;
    pk_K[nt] = 8;
    ++nt;
    indices = idx_out;
    for (int j = 0; j < 3; ++j) shape[j] = out_shape[j];
  }
  // all tile orders from one sort, all packed tables from one launch (the tables without rows are skipped inside)
  {
    const void* ti[16];
    const void* tm[16];
    int64_t tM[16];
    int n = 0;
    for (int t = 0; t < nt; ++t)
      if (t_M[t] > 0) {
        ti[n] = t_idx[t];
        tm[n] = t_mask[t];
        tM[n] = t_M[t];
        ++n;
      }
    // (segments of order_all follow the tables WITH rows only when every table has rows; an empty level breaks the
    // contiguity assumption of the layout, so such a pyramid takes the per-table path)
    WSIS_REQUIRE(n == nt, "a level without voxels: use the per-table build");
    const int rc = wsis_tile_order_batch(n, ti, tm, tM, block_shift, batch_size, reinterpret_cast<int32_t*>(A + L.order_all),
                                         A + L.tile_ws, L.tile_ws_bytes, stream);
    if (rc != WSIS_OK) return rc;
  }
  return wsis_rulebook_pack_batch(nt, pk_nbr, pk_order, pk_out, pk_M, pk_K, stream);
}

}  // extern "C"
