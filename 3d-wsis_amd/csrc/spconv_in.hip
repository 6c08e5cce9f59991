// The 6-channel input convolution (backbone_3D_WSIS.py:43: SubMConv3d(6 -> 32, k3) on [rgb | xyz] voxel features) as an
// im2col product on the fp32 matrix cores, operands straight in registers.
//
//   out[r, :] = sum_k X[nbr[k][r], 0:6] @ W[k]           W [27, 6, 32] (the reference's state-dict layout, no copy)
//
// 27 offsets x 6 channels = 162 values per output row: two lanes per row (lane = row r31 + 32 * half) hold the row's
// im2col vector -- half 0 the offsets 0..13 (84 values), half 1 the offsets 14..26 (78 values + 6 zeros) -- which is
// exactly the A operand of 84 v_mfma_f32_32x32x2_f32 (k index = half); the B operand of MFMA s is W's flattened row
// half * 84 + s at column r31: 84 registers per lane that a wave loads ONCE and keeps while it walks its slices.  No LDS,
// no staging: per 32-row slice 14 coalesced table loads and 42 eight-byte row loads per lane, all in flight together,
// then one 84-deep MFMA chain (exact fp32, fixed order).  The generic kernels pad the 6 channels to a 32-wide chunk
// (27 steps of 16 MFMAs for 6/32 useful work: 72.9 us per launch on the C2 scene).
#include <algorithm>
#include <cstdlib>

#include "common.h"

using namespace wsis;

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int IN_C = 6;           // input channels
constexpr int IN_K = 27;          // offsets
constexpr int IN_H = 14;          // offsets per half (half 1: 13 real ones)
constexpr int IN_S = IN_H * IN_C; // 84 MFMAs per slice

__global__ __launch_bounds__(256, 2) void spconv_in_kernel(const float* __restrict__ X, const int32_t* __restrict__ nbrP,
                                                           const int32_t* __restrict__ order, const float* __restrict__ W,
                                                           const float* __restrict__ bias,
                                                           const float* __restrict__ residual, float* __restrict__ out,
                                                           int64_t M_out, int64_t n_slices) {
  const int lane = threadIdx.x & 63;
  const int r31 = lane & 31, half = lane >> 5;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (int64_t)gridDim.x * 4;
  // ---- B operand: W[half * 14 + j][c][r31], j = 0..13, c = 0..5 (zeros for the 28th offset), kept for every slice
  float b[IN_S];
#pragma unroll
  for (int j = 0; j < IN_H; ++j) {
    const int k = half * IN_H + j;
#pragma unroll
    for (int c = 0; c < IN_C; ++c) b[j * IN_C + c] = k < IN_K ? W[(k * IN_C + c) * 32 + r31] : 0.0f;
  }
  const float bv = bias ? bias[r31] : 0.0f;
  for (int64_t s = wave; s < n_slices; s += n_waves) {
    const int64_t t = s * 32 + r31;
    const bool in = t < M_out;
    const int32_t my_row = in ? (order ? order[t] : (int32_t)t) : -1;
    int32_t idx[IN_H];
#pragma unroll
    for (int j = 0; j < IN_H; ++j) {
      const int k = half * IN_H + j;
      idx[j] = (k < IN_K && in) ? nbrP[(int64_t)k * M_out + t] : -1;
    }
    float a[IN_S];
#pragma unroll
    for (int j = 0; j < IN_H; ++j) {      // branch-free: a missing pair reads row 0 and is zeroed
      const float2* p = reinterpret_cast<const float2*>(X + (int64_t)(idx[j] >= 0 ? idx[j] : 0) * IN_C);
      const float2 v0 = p[0], v1 = p[1], v2 = p[2];
      const bool ok = idx[j] >= 0;
      a[j * IN_C + 0] = ok ? v0.x : 0.0f;
      a[j * IN_C + 1] = ok ? v0.y : 0.0f;
      a[j * IN_C + 2] = ok ? v1.x : 0.0f;
      a[j * IN_C + 3] = ok ? v1.y : 0.0f;
      a[j * IN_C + 4] = ok ? v2.x : 0.0f;
      a[j * IN_C + 5] = ok ? v2.y : 0.0f;
    }
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
#pragma unroll
    for (int i = 0; i < IN_S; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[i], acc, 0, 0, 0);
    // C/D map: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * half; the row id sits in lane `row`
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int rr = (reg & 3) + 8 * (reg >> 2) + 4 * half;
      const int32_t row = __shfl(my_row, rr, 64);
      if (row >= 0) {
        float v = acc[reg] + bv;
        if (residual) v += residual[(int64_t)row * 32 + r31];
        out[(int64_t)row * 32 + r31] = v;
      }
    }
  }
}

// ---- weight gradient of the same layer: dW[k][c][o] = sum_r X[nbr[k][r]][c] dY[r][o] -------------------------------------
// im2col again: dW as a [162 x 32] matrix = Xim^T [162 x rows] @ dY [rows x 32].  A wave loads a slice's im2col rows exactly
// as the forward kernel does (lane = (row, half): 14 table loads, 42 eight-byte row loads), turns them through a 22 KB LDS
// tile so that a lane holds one im2col COLUMN over the slice's rows (the A operand, k index = row), takes dY[row][o] in
// the same row order as B, and runs 6 blocks x 16 MFMAs per slice; the 6 x 16 accumulators stay in registers over all
// slices of the wave.  The next slice's loads are issued before the MFMA phase.  The two waves of a workgroup add through
// LDS in wave order, workgroup slabs are added in workgroup order by spconv_in_dw_reduce_kernel: fixed order, no atomics.
constexpr int IN_MB = 6;                  // 32-column blocks of the 162 (+ padding) im2col columns
constexpr int IN_TP = 172;                // tile row pitch in floats (conflict-free 16-byte row writes and column reads)
constexpr int IN_DW_WAVES = 2;
constexpr int IN_DW_FLOATS = IN_K * IN_C * 32;      // 5184

struct InRows {
  float a[IN_S];
  int32_t my_row;
};

__device__ __forceinline__ void in_load_idx(const int32_t* __restrict__ nbrP, const int32_t* __restrict__ order, int64_t M_out,
                                            int64_t s, int r31, int half, int32_t (&idx)[IN_H], int32_t& my_row) {
  const int64_t t = s * 32 + r31;
  const bool in = t < M_out;
  my_row = in ? (order ? order[t] : (int32_t)t) : -1;
#pragma unroll
  for (int j = 0; j < IN_H; ++j) {
    const int k = half * IN_H + j;
    idx[j] = (k < IN_K && in) ? nbrP[(int64_t)k * M_out + t] : -1;
  }
}

__device__ __forceinline__ void in_load_rows(const float* __restrict__ X, const int32_t (&idx)[IN_H], float (&a)[IN_S]) {
#pragma unroll
  for (int j = 0; j < IN_H; ++j) {
    const float2* p = reinterpret_cast<const float2*>(X + (int64_t)(idx[j] >= 0 ? idx[j] : 0) * IN_C);
    const float2 v0 = p[0], v1 = p[1], v2 = p[2];
    const bool ok = idx[j] >= 0;
    a[j * IN_C + 0] = ok ? v0.x : 0.0f;
    a[j * IN_C + 1] = ok ? v0.y : 0.0f;
    a[j * IN_C + 2] = ok ? v1.x : 0.0f;
    a[j * IN_C + 3] = ok ? v1.y : 0.0f;
    a[j * IN_C + 4] = ok ? v2.x : 0.0f;
    a[j * IN_C + 5] = ok ? v2.y : 0.0f;
  }
}

__global__ __launch_bounds__(64 * IN_DW_WAVES) void spconv_in_dw_kernel(const float* __restrict__ X, const int32_t* __restrict__ nbrP,
                                                                         const int32_t* __restrict__ order,
                                                                         const float* __restrict__ dY, float* __restrict__ partial,
                                                                         int64_t M_out, int64_t n_slices) {
  extern __shared__ __attribute__((aligned(16))) float in_lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r31 = lane & 31, half = lane >> 5;
  float* tile = in_lds + wave * 32 * IN_TP;
  const int64_t w0 = (int64_t)blockIdx.x * IN_DW_WAVES + wave, n_waves = (int64_t)gridDim.x * IN_DW_WAVES;
  f32x16 acc[IN_MB];
#pragma unroll
  for (int mb = 0; mb < IN_MB; ++mb)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[mb][i] = 0.0f;
  int32_t idx[IN_H], my_row = -1;
  float a[IN_S];
  if (w0 < n_slices) {
    in_load_idx(nbrP, order, M_out, w0, r31, half, idx, my_row);
    in_load_rows(X, idx, a);
  }
  for (int64_t s = w0; s < n_slices; s += n_waves) {
    // the slice's rows of dY in the MFMA's k order (row of step i = (i & 3) + 8 (i >> 2) + 4 half)
    float b[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int rr = (i & 3) + 8 * (i >> 2) + 4 * half;
      const int32_t row = __shfl(my_row, rr, 64);
      b[i] = row >= 0 ? dY[(int64_t)row * 32 + r31] : 0.0f;
    }
    // im2col rows -> LDS (lane = row: 84 contiguous floats at column half * 84)
#pragma unroll
    for (int q = 0; q < IN_S / 4; ++q)
      *reinterpret_cast<float4*>(tile + r31 * IN_TP + half * IN_S + q * 4) =
          make_float4(a[q * 4 + 0], a[q * 4 + 1], a[q * 4 + 2], a[q * 4 + 3]);
    // the next slice's loads fly during the MFMA phase
    const int64_t sn = s + n_waves;
    if (sn < n_slices) {
      in_load_idx(nbrP, order, M_out, sn, r31, half, idx, my_row);
      in_load_rows(X, idx, a);
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int mb = 0; mb < IN_MB; ++mb) {
      const int m = mb * 32 + r31;
      float av[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int rr = (i & 3) + 8 * (i >> 2) + 4 * half;
        av[i] = m < 2 * IN_S ? tile[rr * IN_TP + m] : 0.0f;
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], b[i], acc[mb], 0, 0, 0);
    }
    __builtin_amdgcn_wave_barrier();
  }
  // workgroup slab [162][32]: waves add in wave order
  __syncthreads();
  float* slab = in_lds;
  for (int w = 0; w < IN_DW_WAVES; ++w) {
    if (wave == w) {
#pragma unroll
      for (int mb = 0; mb < IN_MB; ++mb)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int m = mb * 32 + (i & 3) + 8 * (i >> 2) + 4 * half;
          if (m < IN_K * IN_C) {
            float* d = slab + m * 32 + r31;
            *d = w == 0 ? acc[mb][i] : *d + acc[mb][i];
          }
        }
    }
    __syncthreads();
  }
  float* dst = partial + (int64_t)blockIdx.x * IN_DW_FLOATS;
  for (int f = threadIdx.x; f < IN_DW_FLOATS / 4; f += blockDim.x)
    reinterpret_cast<float4*>(dst)[f] = reinterpret_cast<const float4*>(slab)[f];
}

// dW = sum of the workgroup slabs in workgroup order: eight lanes per float4 split the slabs, added in lane order
__global__ __launch_bounds__(256) void spconv_in_dw_reduce_kernel(const float4* __restrict__ partial, float4* __restrict__ dW, int P) {
  __shared__ float4 red[8][32];
  const int el = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int t = blockIdx.x * 32 + el;
  constexpr int total4 = IN_DW_FLOATS / 4;
  float4 s = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  if (t < total4) {
#pragma unroll 4
    for (int p = sl; p < P; p += 8) {
      const float4 v = partial[(int64_t)p * total4 + t];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  red[sl][el] = s;
  __syncthreads();
  if (sl == 0 && t < total4) {
#pragma unroll
    for (int l = 1; l < 8; ++l) {
      const float4 v = red[l][el];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    dW[t] = s;
  }
}

}  // namespace

namespace wsis {

bool spconv_in_supported(int K, int Cin, int Cout) {
  const char* e = getenv("WSIS_IN_CONV");      // (read per call)
  return (!e || atoi(e) != 0) && K == IN_K && Cin == IN_C && Cout == 32;
}

int spconv_in_launch(const float* d_X, const int32_t* d_nbr, const int32_t* d_order, const float* d_W, const float* d_bias,
                     const float* d_residual, float* d_out, int64_t M_out, hipEvent_t ka, hipEvent_t kb, hipStream_t st) {
  WSIS_REQUIRE((reinterpret_cast<uintptr_t>(d_X) & 7) == 0, "X must be 8-byte aligned");
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    WSIS_HIP_CHECK(hipGetDevice(&dev));
    WSIS_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  }
  const int64_t n_slices = ceil_div(M_out, 32);
  int64_t wgs = ceil_div(n_slices, 4);
  if (wgs > (int64_t)cus * 2) wgs = (int64_t)cus * 2;      // two 4-wave workgroups per CU (~200 VGPRs per wave)
  hipExtLaunchKernelGGL(spconv_in_kernel, dim3((unsigned)wgs), dim3(256), 0u, st, ka, kb, 0u, d_X, d_nbr, d_order, d_W, d_bias,
                        d_residual, d_out, M_out, n_slices);
  WSIS_LAUNCH_CHECK();
  return WSIS_OK;
}

static int in_dw_wgs(int64_t M_out) {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
      cus = 256;
  }
  const int64_t n_slices = ceil_div(M_out, 32);
  int64_t wgs = ceil_div(n_slices, IN_DW_WAVES);
  if (wgs > (int64_t)cus * 2) wgs = (int64_t)cus * 2;        // two 2-wave workgroups per CU (288 registers per lane: one wave per SIMD)
  return (int)(wgs < 1 ? 1 : wgs);
}

int64_t spconv_in_dw_workspace_bytes(int64_t M_out) { return (int64_t)in_dw_wgs(M_out) * IN_DW_FLOATS * (int64_t)sizeof(float) + 256; }

int spconv_in_dw_launch(const float* d_X, const int32_t* d_nbr, const int32_t* d_order, const float* d_dY, float* d_dW,
                        int64_t M_out, void* d_ws, hipEvent_t ka, hipEvent_t kb, hipStream_t st) {
  WSIS_REQUIRE((reinterpret_cast<uintptr_t>(d_X) & 7) == 0 && (reinterpret_cast<uintptr_t>(d_ws) & 15) == 0 &&
                   (reinterpret_cast<uintptr_t>(d_dW) & 15) == 0,
               "alignment");
  const int wgs = in_dw_wgs(M_out);
  const size_t lds = (size_t)IN_DW_WAVES * 32 * IN_TP * sizeof(float);
  static_assert((size_t)IN_DW_WAVES * 32 * IN_TP >= (size_t)IN_DW_FLOATS, "the slab reuses the tiles");
  float* partial = static_cast<float*>(d_ws);
  hipExtLaunchKernelGGL(spconv_in_dw_kernel, dim3((unsigned)wgs), dim3(64 * IN_DW_WAVES), (unsigned)lds, st, ka, kb, 0u, d_X, d_nbr,
                        d_order, d_dY, partial, M_out, ceil_div(M_out, 32));
  WSIS_LAUNCH_CHECK();
  hipLaunchKernelGGL(spconv_in_dw_reduce_kernel, dim3((IN_DW_FLOATS / 4 + 31) / 32), dim3(256), 0, st,
                     reinterpret_cast<const float4*>(partial), reinterpret_cast<float4*>(d_dW), wgs);
  WSIS_LAUNCH_CHECK();
  return WSIS_OK;
}

}  // namespace wsis
