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

}  // namespace wsis
