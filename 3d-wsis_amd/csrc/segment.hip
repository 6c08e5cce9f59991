// torch_scatter.scatter replacement: sorted-segment wavefront reductions.
//
// torch_scatter 2.0.x [UPSTREAM] does one atomic per element.  Here the index vector is turned
// into a CSR once (stable radix sort => points of a segment stay in ascending position), and one
// wavefront reduces one segment with a fixed summation order: no atomics, deterministic.
// HBM-bound: N*C*4 + N*4 read, S*C*4 written.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_select.hpp>

#include "common.h"

using namespace wsis;

namespace {

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

__global__ void csr_keys_kernel(const int64_t* __restrict__ index, int64_t N, uint32_t* __restrict__ keys,
                                int32_t* __restrict__ iota) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < N;
       i += (int64_t)gridDim.x * blockDim.x) {
    keys[i] = (uint32_t)index[i];
    iota[i] = (int32_t)i;
  }
}

__global__ void csr_offsets_kernel(const uint32_t* __restrict__ sorted, int64_t N, int64_t S,
                                   int32_t* __restrict__ offsets) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < N;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t cur = sorted[i];
    const int64_t prev = i > 0 ? (int64_t)sorted[i - 1] : -1;
    for (int64_t s = prev + 1; s <= cur && s <= S; ++s) offsets[s] = (int32_t)i;
    if (i == N - 1)
      for (int64_t s = cur + 1; s <= S; ++s) offsets[s] = (int32_t)N;
  }
}

// One wave per segment.  Lanes cover G rows x CP channels per step (CP = channels handled per
// pass, power of two <= 64); partial results of the G row groups are combined with a fixed
// butterfly, so the sum order depends only on the segment's (sorted) content.
template <int REDUCE>  // 0 sum, 1 mean, 2 max
__global__ __launch_bounds__(256) void segment_reduce_kernel(
    const float* __restrict__ src, const int32_t* __restrict__ perm, const int32_t* __restrict__ offsets,
    float* __restrict__ out, int32_t* __restrict__ argmax, int64_t S, int C, int CP) {
  const int lane = threadIdx.x & 63;
  const int wave_in_block = threadIdx.x >> 6;
  const int64_t waves_total = (int64_t)gridDim.x * (blockDim.x >> 6);
  const int G = 64 / CP;     // rows in flight per step
  const int c_in = lane % CP;
  const int grp = lane / CP;
  for (int64_t s = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave_in_block; s < S; s += waves_total) {
    const int beg = offsets[s], end = offsets[s + 1];
    for (int c0 = 0; c0 < C; c0 += CP) {
      const int c = c0 + c_in;
      float acc = (REDUCE == 2) ? -INFINITY : 0.0f;
      int32_t arg = -1;
      if (c < C) {
        // eight rows per trip: their perm entries, then their source values, are loaded together (two memory
        // latencies per eight rows instead of per row -- a 168-point superpoint was 84 dependent load pairs);
        // the values are folded in the same order as a row-by-row walk, so the result is unchanged
        constexpr int U = 8;
        int j = beg + grp;
        for (; j + (U - 1) * G < end; j += U * G) {
          int32_t p[U];
          float v[U];
#pragma unroll
          for (int u = 0; u < U; ++u) p[u] = perm[j + u * G];
#pragma unroll
          for (int u = 0; u < U; ++u) v[u] = src[(int64_t)p[u] * C + c];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            if (REDUCE == 2) {
              if (v[u] > acc || arg < 0) {
                acc = v[u];
                arg = p[u];
              }
            } else {
              acc += v[u];
            }
          }
        }
        for (; j < end; j += G) {
          const int32_t p = perm[j];
          const float v = src[(int64_t)p * C + c];
          if (REDUCE == 2) {
            if (v > acc || arg < 0) {  // first occurrence wins inside a group (ascending p)
              acc = v;
              arg = p;
            }
          } else {
            acc += v;
          }
        }
      }
      // combine the G groups: lanes with equal c_in, xor butterfly over the group bits
      for (int off = CP; off < 64; off <<= 1) {
        const float o = __shfl_xor(acc, off, 64);
        const int32_t oa = __shfl_xor(arg, off, 64);
        if (REDUCE == 2) {
          // keep the larger value; ties -> smaller original position (first max)
          const bool take = (oa >= 0) && (arg < 0 || o > acc || (o == acc && oa < arg));
          if (take) {
            acc = o;
            arg = oa;
          }
        } else {
          acc += o;
        }
      }
      if (grp == 0 && c < C) {
        const int cnt = end - beg;
        if (REDUCE == 1) acc = acc / (float)(cnt > 0 ? cnt : 1);
        if (REDUCE == 2) {
          if (cnt == 0) acc = 0.0f;
          argmax[s * C + c] = arg;
        }
        out[s * C + c] = acc;
      }
    }
  }
}

// Many tiny segments (the voxel <- point sum behind the row gather: 153,685 segments of 1.3 rows): eight lanes per
// segment (float4 each), eight segments per wave -- a wave per segment spends three dependent memory latencies on
// every segment it walks (29.7 us for that tensor).  Channels 16 < C <= 32, C % 4 == 0: the wave-per-segment kernel
// then runs two row groups (rows beg, beg + 2, ... and beg + 1, beg + 3, ...) and adds them once; the same two sums
// in the same order here, so the results are bit-identical.
template <int REDUCE>  // 0 sum, 1 mean
__global__ __launch_bounds__(256) void segment_reduce_narrow_kernel(
    const float* __restrict__ src, const int32_t* __restrict__ perm, const int32_t* __restrict__ offsets,
    float* __restrict__ out, int64_t S, int C) {
  const int sub = threadIdx.x & 7;
  const int c = sub * 4;
  const int64_t stride = (int64_t)gridDim.x * (blockDim.x >> 3);
  for (int64_t s = (int64_t)blockIdx.x * (blockDim.x >> 3) + (threadIdx.x >> 3); s < S; s += stride) {
    const int beg = offsets[s], end = offsets[s + 1];
    if (c >= C) continue;
    float4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
    int j = beg;
    for (; j + 3 < end; j += 4) {       // four rows in flight
      const int32_t p0 = perm[j], p1 = perm[j + 1], p2 = perm[j + 2], p3 = perm[j + 3];
      const float4 v0 = *reinterpret_cast<const float4*>(src + (int64_t)p0 * C + c);
      const float4 v1 = *reinterpret_cast<const float4*>(src + (int64_t)p1 * C + c);
      const float4 v2 = *reinterpret_cast<const float4*>(src + (int64_t)p2 * C + c);
      const float4 v3 = *reinterpret_cast<const float4*>(src + (int64_t)p3 * C + c);
      a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
      a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
      a0.x += v2.x; a0.y += v2.y; a0.z += v2.z; a0.w += v2.w;
      a1.x += v3.x; a1.y += v3.y; a1.z += v3.z; a1.w += v3.w;
    }
    for (; j < end; ++j) {
      const float4 v = *reinterpret_cast<const float4*>(src + (int64_t)perm[j] * C + c);
      if (((j - beg) & 1) == 0) {
        a0.x += v.x; a0.y += v.y; a0.z += v.z; a0.w += v.w;
      } else {
        a1.x += v.x; a1.y += v.y; a1.z += v.z; a1.w += v.w;
      }
    }
    float4 r = {a0.x + a1.x, a0.y + a1.y, a0.z + a1.z, a0.w + a1.w};
    if (REDUCE == 1) {
      const float cnt = (float)(end - beg > 0 ? end - beg : 1);
      r.x = r.x / cnt; r.y = r.y / cnt; r.z = r.z / cnt; r.w = r.w / cnt;
    }
    *reinterpret_cast<float4*>(out + s * C + c) = r;
  }
}

// W = 4: four channels per thread (C % 4 == 0, 16-byte aligned rows): a quarter of the index / offset loads and of the
// store instructions -- same values (the division is per element either way)
template <int W>
__global__ void segment_bwd_kernel(const float* __restrict__ dout, const int64_t* __restrict__ index,
                                   const int32_t* __restrict__ offsets, float* __restrict__ dsrc, int64_t N,
                                   int C, int reduce) {
  const int Cw = C / W;
  const int64_t total = N * Cw;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const RowCol rc = row_col(t, Cw, total);
    const int64_t p = rc.row;
    const int c = rc.col * W;
    const int64_t s = index[p];
    float div = 1.0f;
    if (reduce == 1) {
      const int cnt = offsets[s + 1] - offsets[s];
      div = (float)(cnt > 0 ? cnt : 1);
    }
    if (W == 4) {
      float4 g = *reinterpret_cast<const float4*>(dout + s * C + c);
      if (reduce == 1) {
        g.x = g.x / div; g.y = g.y / div; g.z = g.z / div; g.w = g.w / div;
      }
      *reinterpret_cast<float4*>(dsrc + p * C + c) = g;
    } else {
      float g = dout[s * C + c];
      if (reduce == 1) g = g / div;
      dsrc[p * C + c] = g;
    }
  }
}

__global__ void segment_max_bwd_kernel(const float* __restrict__ dout, const int32_t* __restrict__ argmax,
                                       float* __restrict__ dsrc, int64_t S, int C) {
  const int64_t total = S * C;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int32_t a = argmax[t];
    if (a >= 0) dsrc[(int64_t)a * C + (t % C)] = dout[t];  // one writer per (row, c): a row has one segment
  }
}

}  // namespace

extern "C" {

int64_t wsis_segment_csr_workspace_bytes(int64_t N, int64_t S) {
  if (N < 0 || S < 0) return -1;
  if (N == 0) return 256;
  size_t sort_bytes = 0;
  uint32_t* kp = nullptr;
  int32_t* vp = nullptr;
  if (rocprim::radix_sort_pairs(nullptr, sort_bytes, kp, kp, vp, vp, (size_t)N, 0, 32, (hipStream_t)0) !=
      hipSuccess)
    return -1;
  return (int64_t)(3 * align256((size_t)N * 4) + align256(sort_bytes) + 256);
}

int wsis_segment_csr(const int64_t* d_index, int64_t N, int64_t S, int32_t* d_perm, int32_t* d_offsets,
                     void* d_ws, int64_t ws_bytes, void* stream) {
  WSIS_REQUIRE(N >= 0 && S >= 0 && d_offsets, "bad args");
  WSIS_REQUIRE(S < ((int64_t)1 << 31) && N < ((int64_t)1 << 31), "sizes exceed int32");
  hipStream_t st = as_stream(stream);
  if (N == 0) {
    WSIS_HIP_CHECK(hipMemsetAsync(d_offsets, 0, sizeof(int32_t) * (size_t)(S + 1), st));
    return WSIS_OK;
  }
  WSIS_REQUIRE(d_index && d_perm && d_ws, "null pointer");
  char* ws = static_cast<char*>(d_ws);
  const size_t a = align256((size_t)N * 4);
  WSIS_REQUIRE((int64_t)(3 * a) < ws_bytes, "workspace too small");
  uint32_t* keys = reinterpret_cast<uint32_t*>(ws);
  uint32_t* keys_sorted = reinterpret_cast<uint32_t*>(ws + a);
  int32_t* iota = reinterpret_cast<int32_t*>(ws + 2 * a);
  void* temp = ws + 3 * a;
  size_t temp_bytes = (size_t)ws_bytes - 3 * a;
  hipLaunchKernelGGL(csr_keys_kernel, dim3(grid_for(N, 256)), dim3(256), 0, st, d_index, N, keys, iota);
  WSIS_LAUNCH_CHECK();
  int end_bit = 1;
  while (end_bit < 32 && ((int64_t)1 << end_bit) < S + 1) ++end_bit;
  size_t need = 0;
  WSIS_HIP_CHECK(rocprim::radix_sort_pairs(nullptr, need, keys, keys_sorted, iota, d_perm, (size_t)N, 0,
                                           end_bit, st));
  WSIS_REQUIRE(need <= temp_bytes, "workspace too small for sort");
  WSIS_HIP_CHECK(rocprim::radix_sort_pairs(temp, temp_bytes, keys, keys_sorted, iota, d_perm, (size_t)N, 0,
                                           end_bit, st));
  hipLaunchKernelGGL(csr_offsets_kernel, dim3(grid_for(N, 256)), dim3(256), 0, st, keys_sorted, N, S,
                     d_offsets);
  WSIS_LAUNCH_CHECK();
  return WSIS_OK;
}

// ---- the CSRs of up to 8 index vectors from ONE sort (a batch needs six: superpoint ids, point -> voxel map, the two
// directions of the affinity graph and of the ECC graph): keys (table << 32 | index) sorted once with the row number
// local to the table as value, so every table's perm is a contiguous segment of the result and -- the sort being stable
// -- identical to what wsis_segment_csr gives for it; offsets of all tables by one launch.
constexpr int CSRB_MAX = 8;
struct CsrBatch {
  const int64_t* index[CSRB_MAX];
  int64_t base[CSRB_MAX + 1];      // first row of table t in the concatenation
  int64_t S[CSRB_MAX];
  int64_t obase[CSRB_MAX + 1];     // first offset word of table t
  int n;
};
__global__ void csrb_keys_kernel(CsrBatch b, uint64_t* __restrict__ keys, int32_t* __restrict__ iota) {
  const int64_t N = b.base[b.n];
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
    int t = 0;
    while (t + 1 < b.n && i >= b.base[t + 1]) ++t;
    const int64_t r = i - b.base[t];
    keys[i] = ((uint64_t)t << 32) | (uint32_t)b.index[t][r];
    iota[i] = (int32_t)r;
  }
}
__global__ void csrb_offsets_kernel(CsrBatch b, const uint64_t* __restrict__ sorted, int32_t* __restrict__ offsets) {
  const int64_t N = b.base[b.n];
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
    int t = 0;
    while (t + 1 < b.n && i >= b.base[t + 1]) ++t;
    const int64_t r = i - b.base[t], n_t = b.base[t + 1] - b.base[t], S = b.S[t];
    int32_t* off = offsets + b.obase[t];
    const int64_t cur = (int64_t)(uint32_t)sorted[i];
    const int64_t prev = r > 0 ? (int64_t)(uint32_t)sorted[i - 1] : -1;
    for (int64_t s = prev + 1; s <= cur && s <= S; ++s) off[s] = (int32_t)r;
    if (r == n_t - 1)
      for (int64_t s = cur + 1; s <= S; ++s) off[s] = (int32_t)n_t;
  }
}

}  // extern "C"  (a template follows)
namespace {
// ---- the same CSRs WITHOUT a sort (round 6).  The keys are (table, segment id) pairs of bounded range and the wanted order
// inside a segment is the original position, so "stable sort by key" is a counting sort whose segments are then put in
// order: count per (table, segment) -> exclusive scan per table (= the offsets) -> place every row at an atomic cursor of
// its segment (any order) -> order every segment's slice by value (the rows ARE their original positions: ascending
// values = the stable order).  4 launches + 1 fill instead of rocPRIM's merge sort of ~0.5 M 64-bit keys (~23 launches);
// identical perm / offsets.  OPT-IN (WSIS_CSR_COUNTING=1): measured slower than the sort, see wsis_segment_csr_batch.  Out-of-range ids (negative or >= S) share one extra bucket behind the last segment, as the
// sort leaves them behind offsets[S].
struct CsrCount {
  CsrBatch b;
  int64_t cbase[CSRB_MAX + 1];     // first counter word of table t (S_t + 2 words each: buckets 0 .. S_t, one spare)
};
__device__ __forceinline__ int csrc_table(const CsrBatch& b, int64_t i) {
  int t = 0;
  while (t + 1 < b.n && i >= b.base[t + 1]) ++t;
  return t;
}
__global__ void csrc_count_kernel(CsrCount c, int32_t* __restrict__ cnt) {
  const int64_t N = c.b.base[c.b.n];
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
    const int t = csrc_table(c.b, i);
    const int64_t v = c.b.index[t][i - c.b.base[t]];
    const int64_t bucket = (v < 0 || v > c.b.S[t]) ? c.b.S[t] : v;
    atomicAdd(cnt + c.cbase[t] + bucket, 1);
  }
}
// one workgroup per table: exclusive scan of its S + 1 buckets -> offsets [S + 1] and the placement cursors (in place)
__global__ __launch_bounds__(1024) void csrc_scan_kernel(CsrCount c, int32_t* __restrict__ cnt, int32_t* __restrict__ offsets) {
  __shared__ int32_t wsum[16];
  __shared__ int32_t carry_s;
  const int t = blockIdx.x;
  const int64_t n = c.b.S[t] + 1;
  int32_t* cn = cnt + c.cbase[t];
  int32_t* off = offsets + c.b.obase[t];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int64_t i0 = 0; i0 < n; i0 += 1024) {
    const int64_t i = i0 + threadIdx.x;
    const int32_t v = i < n ? cn[i] : 0;
    int32_t x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int32_t y = __shfl_up(x, o, 64);
      if (lane >= o) x += y;
    }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    int32_t base = carry_s;
    for (int w = 0; w < wave; ++w) base += wsum[w];
    const int32_t excl = base + x - v;
    if (i < n) {
      off[i] = excl;
      cn[i] = excl;
    }
    __syncthreads();
    if (threadIdx.x == 1023) carry_s = base + x;
    __syncthreads();
  }
}
__global__ void csrc_place_kernel(CsrCount c, int32_t* __restrict__ cursor, int32_t* __restrict__ perm) {
  const int64_t N = c.b.base[c.b.n];
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
    const int t = csrc_table(c.b, i);
    const int64_t r = i - c.b.base[t];
    const int64_t v = c.b.index[t][r];
    const int64_t bucket = (v < 0 || v > c.b.S[t]) ? c.b.S[t] : v;
    const int32_t pos = atomicAdd(cursor + c.cbase[t] + bucket, 1);
    perm[c.b.base[t] + pos] = (int32_t)r;
  }
}
// one wavefront per (table, bucket): its slice of perm in ascending order.  <= 64 rows: every lane counts the values below
// its own (values are distinct); <= CSRC_LDS rows: bitonic network in the wave's LDS region; more: the same network on
// the slice in memory (a degenerate input -- one segment holding thousands of rows; slow and correct).  All merges are
// "minimum to the lower index", so the virtual +inf padding up to a power of two never moves.
constexpr int CSRC_LDS = 2048;
template <bool GLOBAL>
__device__ __forceinline__ void csrc_bitonic(int32_t* a, int n, int lane) {
  int p2 = 1;
  while (p2 < n) p2 <<= 1;
  for (int k = 2; k <= p2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      const bool flip = j == (k >> 1);
      for (int i = lane; i < p2; i += 64) {
        const int q = flip ? (i ^ (k - 1)) : (i ^ j);
        if (q > i && q < n) {
          const int32_t x = a[i], y = a[q];
          if (x > y) {
            a[i] = y;
            a[q] = x;
          }
        }
      }
      if (GLOBAL) __threadfence();      // (another lane reads these words next: through L2, not a stale L1 line)
      asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
    }
  }
}
__global__ __launch_bounds__(256) void csrc_order_kernel(CsrCount c, const int32_t* __restrict__ offsets, int32_t* __restrict__ perm) {
  __shared__ int32_t stage[4][CSRC_LDS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int64_t total = 0;
  for (int t = 0; t < c.b.n; ++t) total += c.b.S[t] + 1;
  const int64_t nw = (int64_t)gridDim.x * 4;
  for (int64_t u = (int64_t)blockIdx.x * 4 + wave; u < total; u += nw) {
    int t = 0;
    int64_t s = u;
    while (s >= c.b.S[t] + 1) {
      s -= c.b.S[t] + 1;
      ++t;
    }
    const int64_t n_t = c.b.base[t + 1] - c.b.base[t];
    const int32_t* off = offsets + c.b.obase[t];
    const int32_t beg = off[s];
    const int32_t end = s < c.b.S[t] ? off[s + 1] : (int32_t)n_t;      // (the last bucket: out-of-range ids)
    const int n = end - beg;
    if (n < 2) continue;
    int32_t* a = perm + c.b.base[t] + beg;
    if (n <= 64) {
      const int32_t v = lane < n ? a[lane] : 0x7fffffff;
      int rank = 0;
      for (int j = 0; j < n; ++j) rank += __builtin_amdgcn_readlane(v, j) < v ? 1 : 0;
      if (lane < n) a[rank] = v;
    } else if (n <= CSRC_LDS) {
      int32_t* l = stage[wave];
      for (int i = lane; i < n; i += 64) l[i] = a[i];
      asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
      csrc_bitonic<false>(l, n, lane);
      for (int i = lane; i < n; i += 64) a[i] = l[i];
    } else {
      csrc_bitonic<true>(a, n, lane);
    }
  }
}
}  // namespace
extern "C" {

int64_t wsis_segment_csr_batch_workspace_bytes(int64_t N_all) {
  if (N_all < 0) return -1;
  if (N_all == 0) return 256;
  size_t sort_bytes = 0;
  uint64_t* kp = nullptr;
  int32_t* vp = nullptr;
  if (rocprim::radix_sort_pairs(nullptr, sort_bytes, kp, kp, vp, vp, (size_t)N_all, 0, 36, (hipStream_t)0) != hipSuccess)
    return -1;
  return (int64_t)(2 * align256((size_t)N_all * 8) + align256((size_t)N_all * 4) + align256(sort_bytes) + 256);
}

int wsis_segment_csr_batch(int32_t n, const void* const* h_index, const int64_t* h_N, const int64_t* h_S,
                           int32_t* d_perm_all, int32_t* d_offsets_all, void* d_ws, int64_t ws_bytes, void* stream) {
  WSIS_REQUIRE(n >= 1 && n <= CSRB_MAX && h_index && h_N && h_S && d_offsets_all, "bad args (at most 8 index vectors)");
  CsrBatch b;
  b.n = n;
  b.base[0] = 0;
  b.obase[0] = 0;
  for (int t = 0; t < n; ++t) {
    WSIS_REQUIRE(h_N[t] >= 0 && h_S[t] >= 0 && h_S[t] < ((int64_t)1 << 31) && (h_N[t] == 0 || h_index[t]), "bad table");
    b.index[t] = static_cast<const int64_t*>(h_index[t]);
    b.S[t] = h_S[t];
    b.base[t + 1] = b.base[t] + h_N[t];
    b.obase[t + 1] = b.obase[t] + h_S[t] + 1;
  }
  const int64_t N = b.base[n];
  WSIS_REQUIRE(N < ((int64_t)1 << 31), "sizes exceed int32");
  hipStream_t st = as_stream(stream);
  // tables without rows: all their offsets are zero (the offsets kernel never visits them)
  for (int t = 0; t < n; ++t)
    if (h_N[t] == 0)
      WSIS_HIP_CHECK(hipMemsetAsync(d_offsets_all + b.obase[t], 0, sizeof(int32_t) * (size_t)(h_S[t] + 1), st));
  if (N == 0) return WSIS_OK;
  WSIS_REQUIRE(d_perm_all && d_ws, "null pointer");
  char* ws = static_cast<char*>(d_ws);
  {
    // counting form (WSIS_CSR_COUNTING=1; opt-in) where its counters fit the workspace.  Measured (tools/r06_csr.sh,
    // profiles/r06_ab_csr.txt): identical CSRs, 5 launches instead of ~23 -- and 288 us of kernels instead of ~130 (the
    // per-segment ordering pass walks 165 k segments with two dependent loads each: 162 us; count and place serialise on
    // the atomics of 2.3 k superpoint counters: 45 us each): the step 7.81 -> 7.89 ms, four scenes 20.28 -> 20.50.
    const char* ce = getenv("WSIS_CSR_COUNTING");
    CsrCount c;
    c.b = b;
    c.cbase[0] = 0;
    for (int t = 0; t < n; ++t) c.cbase[t + 1] = c.cbase[t] + h_S[t] + 2;
    const int64_t words = c.cbase[n];
    if (ce && atoi(ce) != 0 && words * 4 + 256 <= ws_bytes) {
      int32_t* cnt = reinterpret_cast<int32_t*>(ws);
      WSIS_HIP_CHECK(hipMemsetAsync(cnt, 0, (size_t)words * 4, st));
      hipLaunchKernelGGL(csrc_count_kernel, dim3(grid_for(N, 256)), dim3(256), 0, st, c, cnt);
      hipLaunchKernelGGL(csrc_scan_kernel, dim3((unsigned)n), dim3(1024), 0, st, c, cnt, d_offsets_all);
      hipLaunchKernelGGL(csrc_place_kernel, dim3(grid_for(N, 256)), dim3(256), 0, st, c, cnt, d_perm_all);
      int64_t segs = 0;
      for (int t = 0; t < n; ++t) segs += h_S[t] + 1;
      int64_t g = (segs + 3) / 4;
      if (g > 1024) g = 1024;
      if (g < 1) g = 1;
      hipLaunchKernelGGL(csrc_order_kernel, dim3((unsigned)g), dim3(256), 0, st, c, d_offsets_all, d_perm_all);
      WSIS_LAUNCH_CHECK();
      return WSIS_OK;
    }
  }
  const size_t a8 = align256((size_t)N * 8), a4 = align256((size_t)N * 4);
  WSIS_REQUIRE((int64_t)(2 * a8 + a4) < ws_bytes, "workspace too small");
  uint64_t* keys = reinterpret_cast<uint64_t*>(ws);
  uint64_t* keys_sorted = reinterpret_cast<uint64_t*>(ws + a8);
  int32_t* iota = reinterpret_cast<int32_t*>(ws + 2 * a8);
  void* temp = ws + 2 * a8 + a4;
  size_t temp_bytes = (size_t)ws_bytes - (2 * a8 + a4);
  hipLaunchKernelGGL(csrb_keys_kernel, dim3(grid_for(N, 256)), dim3(256), 0, st, b, keys, iota);
  WSIS_LAUNCH_CHECK();
  int64_t s_max = 0;
  for (int t = 0; t < n; ++t) s_max = h_S[t] > s_max ? h_S[t] : s_max;
  int idx_bits = 1;
  while (idx_bits < 32 && ((int64_t)1 << idx_bits) < s_max + 1) ++idx_bits;
  // the table number sits at bit 32: sort the low index bits, then the table bits (stable radix sort: two ranges of
  // one key would need two sorts, so the whole span up to the table bits is sorted -- 36 bits at most)
  const int end_bit = 32 + 3;
  (void)idx_bits;
  size_t need = 0;
  WSIS_HIP_CHECK(rocprim::radix_sort_pairs(nullptr, need, keys, keys_sorted, iota, d_perm_all, (size_t)N, 0, end_bit, st));
  WSIS_REQUIRE(need <= temp_bytes, "workspace too small for sort");
  WSIS_HIP_CHECK(rocprim::radix_sort_pairs(temp, temp_bytes, keys, keys_sorted, iota, d_perm_all, (size_t)N, 0, end_bit, st));
  hipLaunchKernelGGL(csrb_offsets_kernel, dim3(grid_for(N, 256)), dim3(256), 0, st, b, keys_sorted, d_offsets_all);
  WSIS_LAUNCH_CHECK();
  return WSIS_OK;
}

int wsis_segment_reduce_fwd(const float* d_src, const int32_t* d_perm, const int32_t* d_offsets,
                            float* d_out, int32_t* d_argmax, int64_t N, int64_t S, int32_t C,
                            int32_t reduce, void* stream) {
  WSIS_REQUIRE(N >= 0 && S >= 0 && C >= 1 && reduce >= 0 && reduce <= 2, "bad args");
  if (S == 0) return WSIS_OK;
  WSIS_REQUIRE(d_offsets && d_out && (N == 0 || (d_src && d_perm)), "null pointer");
  WSIS_REQUIRE(reduce != 2 || d_argmax, "max needs an argmax buffer");
  int CP = 1;
  while (CP < C && CP < 64) CP <<= 1;
  const int block = 256;
  hipStream_t st = as_stream(stream);
  const char* narrow_env = getenv("WSIS_SEGMENT_NARROW");      // (read per call: 0 keeps the wave-per-segment kernel)
  if ((!narrow_env || atoi(narrow_env) != 0) && reduce != 2 && CP == 32 && (C & 3) == 0 && N < 4 * S &&
      (reinterpret_cast<uintptr_t>(d_src) & 15) == 0 &&
      (reinterpret_cast<uintptr_t>(d_out) & 15) == 0) {
    int64_t gn = ceil_div(S, block / 8);
    if (gn > 256 * 16) gn = 256 * 16;
    if (reduce == 0)
      hipLaunchKernelGGL(segment_reduce_narrow_kernel<0>, dim3((unsigned)gn), dim3(block), 0, st, d_src, d_perm,
                         d_offsets, d_out, S, C);
    else
      hipLaunchKernelGGL(segment_reduce_narrow_kernel<1>, dim3((unsigned)gn), dim3(block), 0, st, d_src, d_perm,
                         d_offsets, d_out, S, C);
    WSIS_LAUNCH_CHECK();
    return WSIS_OK;
  }
  int64_t g = ceil_div(S, block / 64);
  if (g > 256 * 16) g = 256 * 16;
  if (reduce == 0)
    hipLaunchKernelGGL(segment_reduce_kernel<0>, dim3((unsigned)g), dim3(block), 0, st, d_src, d_perm,
                       d_offsets, d_out, d_argmax, S, C, CP);
  else if (reduce == 1)
    hipLaunchKernelGGL(segment_reduce_kernel<1>, dim3((unsigned)g), dim3(block), 0, st, d_src, d_perm,
                       d_offsets, d_out, d_argmax, S, C, CP);
  else
    hipLaunchKernelGGL(segment_reduce_kernel<2>, dim3((unsigned)g), dim3(block), 0, st, d_src, d_perm,
                       d_offsets, d_out, d_argmax, S, C, CP);
  WSIS_LAUNCH_CHECK();
  return WSIS_OK;
}

int wsis_segment_reduce_bwd(const float* d_dout, const int64_t* d_index, const int32_t* d_offsets,
                            const int32_t* d_argmax, float* d_dsrc, int64_t N, int64_t S, int32_t C,
                            int32_t reduce, void* stream) {
  WSIS_REQUIRE(N >= 0 && S >= 0 && C >= 1 && reduce >= 0 && reduce <= 2, "bad args");
  if (N == 0 || S == 0) return WSIS_OK;
  WSIS_REQUIRE(d_dout && d_dsrc, "null pointer");
  hipStream_t st = as_stream(stream);
  if (reduce == 2) {
    WSIS_REQUIRE(d_argmax, "max backward needs argmax");
    hipLaunchKernelGGL(segment_max_bwd_kernel, dim3(grid_for(S * C, 256)), dim3(256), 0, st, d_dout, d_argmax,
                       d_dsrc, S, C);
  } else {
    WSIS_REQUIRE(d_index && d_offsets, "null pointer");
    const bool vec = C % 4 == 0 && ((reinterpret_cast<uintptr_t>(d_dout) | reinterpret_cast<uintptr_t>(d_dsrc)) & 15) == 0;
    if (vec)
      hipLaunchKernelGGL(segment_bwd_kernel<4>, dim3(grid_for(N * (C / 4), 256)), dim3(256), 0, st, d_dout, d_index,
                         d_offsets, d_dsrc, N, C, reduce);
    else
      hipLaunchKernelGGL(segment_bwd_kernel<1>, dim3(grid_for(N * C, 256)), dim3(256), 0, st, d_dout, d_index,
                         d_offsets, d_dsrc, N, C, reduce);
  }
  WSIS_LAUNCH_CHECK();
  return WSIS_OK;
}

}  // extern "C"
