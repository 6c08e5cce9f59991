// Weight gradient of the sparse convolutions, wave-autonomous form:
//   dW[k][ci][co] = sum over rows r with nbr[k][r] >= 0 of  X[nbr[k][r]][ci] * dY[r][co]
// for layers whose channel counts are multiples of 32 and K <= 32 (every UNet layer but the 6-channel input conv).
//
// One wave = one worker with a FIXED (offset group, 32-channel chunk of Cin, 32-channel block of Cout): it owns up to
// 8 kernel offsets (k = og, og + NOG, ...) whose 32x32 accumulators stay in registers for the whole launch (8 x 16
// AGPRs), and streams through its share of the 32-row slices of the output (tile order, strided so that every worker
// samples the whole scene).  Per slice: the dY rows of the slice (the B operand, 32 rows x 128 B) are gathered ONCE into
// LDS by LDS-DMA and kept as 16 fragment registers for all offsets of the slice; per active offset the X rows it pairs
// with (the A operand) are gathered the same way (full 128-byte lines, 8 rows per instruction -- the round-1 kernel
// fetched both operands with 4-byte loads, two rows per instruction) and read back transposed (lane = input channel)
// with conflict-free ds_read_b32; 16 v_mfma_f32_32x32x2_f32 per (slice, offset).  The next step's gather flies while
// the current step computes.  The 4 waves of a workgroup share a combination and add their accumulators through LDS in
// wave order; workgroup slabs are added in workgroup order by dw2_reduce_kernel: no atomics, bit-reproducible.
#include <cstdlib>

#include "common.h"

using namespace wsis;

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int GS = 8;                 // offsets per worker
constexpr int TILE = 32 * 128;        // bytes of a 32-row x 32-channel image
constexpr int HDR = (GS * 32 + 32) * 4;                  // nb[8][32] + row[32]
constexpr int WAVE_LDS = 2 * HDR + 2 * TILE + 2 * TILE;  // headers, dY tiles, X tiles (double-buffered)

__device__ __attribute__((aligned(256))) float g_dw_zero_row[64];

__device__ __forceinline__ void dma16(const void* src, void* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

template <bool DIAG>
__global__ __launch_bounds__(256, 2) void spconv_dw2_kernel(const float* __restrict__ X, const int32_t* __restrict__ nbrS,
                                                         const int32_t* __restrict__ order, const float* __restrict__ dY,
                                                         float* __restrict__ partial, int64_t M_out, int K, int Cin,
                                                         int Cout, int NOG, unsigned long long* dbg) {
  unsigned long long t_start = 0, t_loop = 0, t_end_loop = 0;
  unsigned n_steps = 0, n_slices_done = 0;
  if (DIAG) t_start = __builtin_readcyclecounter();
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r31 = lane & 31, half = lane >> 5;
  unsigned char* const my = lds + wave * WAVE_LDS;
  int32_t* const hdr0 = reinterpret_cast<int32_t*>(my);
  unsigned char* const Bt = my + 2 * HDR;
  unsigned char* const At = Bt + 2 * TILE;

  const int nchunk = Cin >> 5, nblk = Cout >> 5;
  int combo = blockIdx.y;
  const int cb = combo % nblk;
  combo /= nblk;
  const int c = combo % nchunk;
  const int og = combo / nchunk;
  const int64_t n_slices = (M_out + 31) >> 5;
  // strided share of the slices: every worker samples the whole scene (contiguous runs measured 20-30 % slower: the
  // pair density varies along the tile order)
  const int64_t stride = (int64_t)gridDim.x * 4;
  const int64_t first = (int64_t)blockIdx.x * 4 + wave;
  const int64_t last = n_slices;

  f32x16 acc0, acc1, acc2, acc3, acc4, acc5, acc6, acc7;
#pragma unroll
  for (int i = 0; i < 16; ++i)
    acc0[i] = acc1[i] = acc2[i] = acc3[i] = acc4[i] = acc5[i] = acc6[i] = acc7[i] = 0.0f;

  const char* const Xb = reinterpret_cast<const char*>(X) + c * 128;
  const char* const Yb = reinterpret_cast<const char*>(dY) + cb * 128;
  const int64_t x_pitch = (int64_t)Cin * 4, y_pitch = (int64_t)Cout * 4;
  const int d_row = lane >> 3, d_piece = lane & 7;
  const char* const zrow = reinterpret_cast<const char*>(g_dw_zero_row) + d_piece * 16;

  // header of a slice in registers: table entries of this worker's offsets (lane: offset slot lane >> 3, rows
  // (lane & 7) * 4 .. + 3) and the slice's output rows
  struct Hreg {
    int32_t nb[4];
    int32_t row;
  };
  auto load_hdr = [&](int64_t s, Hreg& h) {
    const int k = og + (lane >> 3) * NOG;
    const int64_t t0 = s * 32 + (lane & 7) * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int64_t t = t0 + q;
      const bool ok = k < K && t < M_out;
      int32_t v;
      if (nbrS)
        v = nbrS[ok ? (int64_t)k * M_out + t : 0];
      else
        v = order ? order[ok ? t : 0] : (int32_t)t;        // dense 1x1: the row pairs with itself
      h.nb[q] = ok ? v : -1;
    }
    const int64_t tr = s * 32 + r31;
    h.row = tr < M_out ? (order ? order[tr] : (int32_t)tr) : -1;
  };
  // registers -> LDS; returns the mask of offset slots with at least one pair in the slice
  auto store_hdr = [&](int32_t* hb, const Hreg& h) -> uint32_t {
    int4 v = make_int4(h.nb[0], h.nb[1], h.nb[2], h.nb[3]);
    *reinterpret_cast<int4*>(hb + lane * 4) = v;            // nb[slot][row]: slot = lane >> 3, rows (lane & 7) * 4 ..
    if (lane < 32) hb[GS * 32 + lane] = h.row;
    const bool any = (h.nb[0] & h.nb[1] & h.nb[2] & h.nb[3]) >= 0;    // some entry non-negative
    const unsigned long long b = __ballot(any);
    uint32_t m = 0u;
#pragma unroll
    for (int j = 0; j < GS; ++j)
      if ((b >> (8 * j)) & 0xffull) m |= 1u << j;
    return m;
  };
  auto issueB = [&](const int32_t* hb, unsigned char* dst) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int32_t r = hb[GS * 32 + i * 8 + d_row];
      const char* src = r >= 0 ? Yb + (int64_t)r * y_pitch + d_piece * 16 : zrow;
      dma16(src, dst + i * 1024);
    }
  };
  auto issueA = [&](const int32_t* hb, int j, unsigned char* dst) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int32_t g = hb[j * 32 + i * 8 + d_row];
      const char* src = g >= 0 ? Xb + (int64_t)g * x_pitch + d_piece * 16 : zrow;
      dma16(src, dst + i * 1024);
    }
  };
  // MFMA operands: lane (channel r31, half) takes rows 2s + half of a 32-row image, s = 0 .. 15
  auto readfrag = [&](const unsigned char* img, float (&f)[16]) {
    const float* p = reinterpret_cast<const float*>(img) + half * 32 + r31;
#pragma unroll
    for (int s = 0; s < 16; ++s) f[s] = p[s * 64];
  };
  auto mfma16 = [&](f32x16& acc, const float (&a)[16], const float (&b)[16]) {
#pragma unroll
    for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
  };

  if (first < last) {
    Hreg h, hn;
    int64_t cur = first;
    uint32_t mask = 0u;
    int hb = 0;
    // first non-empty slice of this worker (a slice without any pair in the worker's offsets is skipped)
    for (;;) {
      load_hdr(cur, h);
      mask = store_hdr(hdr0 + hb * (HDR / 4), h);
      if (mask || cur + stride >= last) break;
      cur += stride;
    }
    if (mask) {
      int bb = 0, ab = 0;
      issueB(hdr0 + hb * (HDR / 4), Bt);
      issueA(hdr0 + hb * (HDR / 4), __builtin_ctz(mask), At);
      int64_t nxt = cur + stride;
      bool have_next = nxt < last;
      if (have_next) load_hdr(nxt, hn);
      bool new_slice = true;
      float bfr[16], afr[16];
      if (DIAG) t_loop = __builtin_readcyclecounter();
      uint32_t mask_n = 0u;
      bool advance_slice = false;
      // one (slice, offset slot) step; jslot is a compile-time constant at every call site so that each accumulator is
      // touched by exactly one MFMA chain
      auto step = [&](f32x16& acc, int jslot) {
        // the step's images (and, at a slice change, the next header registers) have landed
        asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
        if (new_slice) {
          readfrag(Bt + bb * TILE, bfr);
          new_slice = false;
        }
        readfrag(At + ab * TILE, afr);
        // bring in the next step while this one computes
        const uint32_t rem = mask & ~((2u << jslot) - 1u);
        if (rem) {
          issueA(hdr0 + hb * (HDR / 4), __builtin_ctz(rem), At + (ab ^ 1) * TILE);
        } else {
          while (have_next) {       // next slice with work for this worker
            mask_n = store_hdr(hdr0 + (hb ^ 1) * (HDR / 4), hn);
            if (mask_n) break;
            nxt += stride;
            have_next = nxt < last;
            if (have_next) load_hdr(nxt, hn);     // rare path (slice without pairs): the load is waited for in place
          }
          advance_slice = have_next;
          if (have_next) {
            issueB(hdr0 + (hb ^ 1) * (HDR / 4), Bt + (bb ^ 1) * TILE);
            issueA(hdr0 + (hb ^ 1) * (HDR / 4), __builtin_ctz(mask_n), At + (ab ^ 1) * TILE);
          }
        }
        mfma16(acc, afr, bfr);
        ab ^= 1;
        if (DIAG) ++n_steps;
      };
      for (;;) {
        if (mask & 1u) step(acc0, 0);
        if (mask & 2u) step(acc1, 1);
        if (mask & 4u) step(acc2, 2);
        if (mask & 8u) step(acc3, 3);
        if (mask & 16u) step(acc4, 4);
        if (mask & 32u) step(acc5, 5);
        if (mask & 64u) step(acc6, 6);
        if (mask & 128u) step(acc7, 7);
        if (DIAG) ++n_slices_done;
        if (!advance_slice) break;
        advance_slice = false;
        cur = nxt;
        mask = mask_n;
        bb ^= 1;
        hb ^= 1;
        new_slice = true;
        nxt = cur + stride;
        have_next = nxt < last;
        if (have_next) load_hdr(nxt, hn);        // prefetch: consumed at the end of the slice just entered
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
  if (DIAG) t_end_loop = __builtin_readcyclecounter();
  __syncthreads();

  // ---- the 4 waves' accumulators of one offset slot at a time: LDS, added in wave order, one slab per workgroup.
  // C/D map: col (co) = lane & 31, row (ci) = (reg & 3) + 8 * (reg >> 2) + 4 * half
  float* const red = reinterpret_cast<float*>(lds);          // [4][32 ci][32 co], stride WAVE_LDS / 4 floats per wave
  float* const slab = partial + (int64_t)blockIdx.x * K * Cin * Cout;
  auto flush = [&](const f32x16& acc, int jslot) {
    const int k = og + jslot * NOG;
    if (k >= K) return;               // uniform over the workgroup
    float* mine = red + wave * (WAVE_LDS / 4);
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int ci = (reg & 3) + 8 * (reg >> 2) + 4 * half;
      mine[ci * 32 + r31] = acc[reg];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 1024; e += 256) {
      const float v = ((red[e] + red[(WAVE_LDS / 4) + e]) + red[2 * (WAVE_LDS / 4) + e]) + red[3 * (WAVE_LDS / 4) + e];
      const int ci = e >> 5, co = e & 31;
      slab[((int64_t)k * Cin + c * 32 + ci) * Cout + cb * 32 + co] = v;
    }
    __syncthreads();
  };
  flush(acc0, 0);
  flush(acc1, 1);
  flush(acc2, 2);
  flush(acc3, 3);
  flush(acc4, 4);
  flush(acc5, 5);
  flush(acc6, 6);
  flush(acc7, 7);
  if (DIAG && lane == 0) {
    unsigned long long* d = dbg + ((int64_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * 6;
    d[0] = t_start; d[1] = t_loop; d[2] = t_end_loop; d[3] = __builtin_readcyclecounter(); d[4] = n_steps; d[5] = n_slices_done;
  }
}

// dW = sum over workgroup slabs, slab order fixed; four floats per thread
__global__ void dw2_reduce_kernel(const float4* __restrict__ partial, float4* __restrict__ dW, int64_t total4, int P) {
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total4; t += (int64_t)gridDim.x * blockDim.x) {
    float4 s = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll 8
    for (int p = 0; p < P; ++p) {
      const float4 v = partial[(int64_t)p * total4 + t];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    dW[t] = s;
  }
}

int dw2_env(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}

// workgroups per combination (= slabs): ~target waves over the launch, never more than there are slices
int dw2_P(int64_t M_out, int K, int Cin, int Cout) {
  static int target = -1;
  if (target < 0) target = dw2_env("WSIS_DW2_WAVES", 2048);
  const int NOG = (K + GS - 1) / GS;
  const int64_t combos = (int64_t)NOG * (Cin / 32) * (Cout / 32);
  const int64_t n_slices = (M_out + 31) / 32;
  int64_t P = (target / 4 + combos - 1) / combos;
  const int64_t cap = (n_slices + 3) / 4;
  if (P > cap) P = cap;
  if (P < 1) P = 1;
  return (int)P;
}

}  // namespace

namespace wsis {

bool dw2_supported(int K, int Cin, int Cout) {
  static int on = -1;
  if (on < 0) on = dw2_env("WSIS_DW2", 1);
  return on && K >= 1 && K <= 32 && Cin >= 32 && Cin % 32 == 0 && Cout >= 32 && Cout % 32 == 0;
}

int64_t dw2_workspace_bytes(int64_t M_out, int K, int Cin, int Cout) {
  return (int64_t)dw2_P(M_out, K, Cin, Cout) * K * Cin * Cout * (int64_t)sizeof(float) + 256;
}

int dw2_launch(const float* d_X, const int32_t* d_nbr, const int32_t* d_order, const float* d_dY, float* d_dW,
               int64_t M_out, int K, int Cin, int Cout, void* d_ws, hipStream_t st) {
  const int NOG = (K + GS - 1) / GS;
  const int P = dw2_P(M_out, K, Cin, Cout);
  float* partial = static_cast<float*>(d_ws);
  const dim3 grid((unsigned)P, (unsigned)(NOG * (Cin / 32) * (Cout / 32)), 1);
  const size_t ldsb = (size_t)WAVE_LDS * 4;
  static bool attr_set = false;
  if (!attr_set) {
    WSIS_HIP_CHECK(hipFuncSetAttribute((const void*)spconv_dw2_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)ldsb));
    attr_set = true;
  }
  ProfScope prof(1, st);
  hipLaunchKernelGGL(spconv_dw2_kernel<false>, grid, dim3(256), ldsb, st, d_X, d_nbr, d_order, d_dY, partial, M_out, K,
                     Cin, Cout, NOG, (unsigned long long*)nullptr);
  prof.stop();
  WSIS_LAUNCH_CHECK();
  const int64_t total4 = (int64_t)K * Cin * Cout / 4;
  hipLaunchKernelGGL(dw2_reduce_kernel, dim3(grid_for(total4, 256)), dim3(256), 0, st,
                     reinterpret_cast<const float4*>(partial), reinterpret_cast<float4*>(d_dW), total4, P);
  WSIS_LAUNCH_CHECK();
  return WSIS_OK;
}

}  // namespace wsis

// diagnostic build of the kernel: per-wave cycle stamps (start, loop begin, loop end, end) and step / slice counts,
// 6 x u64 per wave in launch order; *n_waves receives the wave count.  tools/dw2_stamps.py reads it.
extern "C" int wsis_debug_dw2_diag(const void* d_X, const void* d_nbr, const void* d_order, const void* d_dY,
                                   int64_t M_out, int K, int Cin, int Cout, void* d_ws, void* d_dbg, int64_t dbg_bytes,
                                   int64_t* n_waves, void* stream) {
  WSIS_REQUIRE(wsis::dw2_supported(K, Cin, Cout), "shape not supported by the dw2 kernel");
  const int NOG = (K + GS - 1) / GS;
  const int P = dw2_P(M_out, K, Cin, Cout);
  const dim3 grid((unsigned)P, (unsigned)(NOG * (Cin / 32) * (Cout / 32)), 1);
  *n_waves = (int64_t)grid.x * grid.y * 4;
  WSIS_REQUIRE(dbg_bytes >= *n_waves * 48, "stamp buffer too small");
  const size_t ldsb = (size_t)WAVE_LDS * 4;
  WSIS_HIP_CHECK(hipFuncSetAttribute((const void*)spconv_dw2_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)ldsb));
  hipLaunchKernelGGL(spconv_dw2_kernel<true>, grid, dim3(256), ldsb, wsis::as_stream(stream),
                     static_cast<const float*>(d_X), static_cast<const int32_t*>(d_nbr),
                     static_cast<const int32_t*>(d_order), static_cast<const float*>(d_dY), static_cast<float*>(d_ws),
                     M_out, K, Cin, Cout, NOG, static_cast<unsigned long long*>(d_dbg));
  WSIS_LAUNCH_CHECK();
  return WSIS_OK;
}
