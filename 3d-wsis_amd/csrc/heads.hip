// Prediction heads over the superpoint rows (backbone_3D_WSIS.py:59-96 `head(cin, cout)` = Linear(64,64) -> BatchNorm1d ->
// ReLU -> Linear(64,cout); :195-204 the four heads on the GNN output, :210-216 the bias-free q / k / v layers on the same
// rows, :253 the discriminative-feature head).  The reference runs every layer as its own chain of launches (forward
// 4 per head, backward ~12 with the bias / weight gradients and the gradient accumulation of the shared input); on
// S ~ 2,300 rows each of those is pure launch latency.  Here ALL blocks that read the same input x [S,64] are
//
//   forward  3 launches   lin1: one wave per (32-row slice, block): H_p = x W1_p^T + b1_p on the fp32 matrix cores
//                               (32 v_mfma_f32_32x32x2_f32 per 32-column half, operands straight from global in
//                               64-byte pieces) + the BatchNorm partials of the slice (sum, centred sum of squares);
//                         fin:  the slice partials of a head combined by one workgroup (fp64, Chan's form as bn.hip)
//                         lin2: BatchNorm + ReLU on the wave's fragment, then times W2_p^T
//   backward 4 launches   bwd1: dA = dY_p W2_p, dz = dA * [relu'], slice partials (sum dz, sum dz x^), dW2 / db2 partials
//                         fin:  the global sums of the BatchNorm backward (+ d gamma, d beta)
//                         bwd2: BatchNorm backward, dx = sum_p dH_p W1_p (blocks of a
//                               slice on 8 waves, added through LDS in block order), dW1 / db1 partials
//                         bwd3: fixed-order sum of the partials into the caller's gradient tensors
//
// No atomics; every sum has a fixed order (run-to-run identical).  Exact fp32 products (no reduced-precision MFMA).
#include <cstring>

#include "common.h"

using namespace wsis;

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int HC = 64;                    // width of the input, of the hidden layer and of the plain layers
constexpr int HB = WSIS_HEADS_MAX;        // blocks per call
constexpr int W1P = HC * HC + HC;         // floats of a dW1 / db1 partial
constexpr int W2P = 32 * HC + 32;         // floats of a dW2 / db2 partial (cout <= 32)
constexpr int MAX_PART = 128;             // gradient partials per call (slice sets)

__device__ __forceinline__ int row_of(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

__device__ __forceinline__ float half_sum(float v) { return v + __shfl_xor(v, 32, 64); }

// fragment of a row-major [rows, 64] matrix for the A side (lane = row r31) or, of a [cols, 64] matrix, for the B^T side
// (lane = column r31): the lane's 16 values of chunk c are channels c*32 + half*16 .. +15 (k index (i, half))
__device__ __forceinline__ void load_frag(const float* base, bool ok, int half, float (&f)[2][16]) {
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ok) v = *reinterpret_cast<const float4*>(base + c * 32 + half * 16 + q * 4);
      f[c][q * 4 + 0] = v.x; f[c][q * 4 + 1] = v.y; f[c][q * 4 + 2] = v.z; f[c][q * 4 + 3] = v.w;
    }
}

__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.0f;
  return z;
}

__device__ __forceinline__ f32x16 mma64(const float (&a)[2][16], const float (&b)[2][16], f32x16 acc) {
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c][i], b[c][i], acc, 0, 0, 0);
  return acc;
}

// ---- forward 1: H_p = x W1_p^T + b1_p, BatchNorm slice partials ---------------------------------------------------
__global__ __launch_bounds__(256) void heads_lin1_kernel(const wsis_heads h, const float* __restrict__ X, int64_t S,
                                                         float* __restrict__ part, int training) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r31 = lane & 31, half = lane >> 5;
  const int p = blockIdx.y * 4 + wave, nb = h.n_heads + h.n_lin;
  if (p >= nb) return;
  const int64_t s = blockIdx.x, row = s * 32 + r31;
  float xa[2][16];
  load_frag(X + row * HC, row < S, half, xa);
  const float* W1 = h.W1[p];
  const float* b1 = p < h.n_heads ? h.b1[p] : nullptr;
  float* H = h.hidden[p];
  const int64_t left = S - s * 32;
  const float n = (float)(left < 32 ? left : 32);
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    float wb[2][16];
    load_frag(W1 + (cb * 32 + r31) * HC, true, half, wb);
    f32x16 acc = mma64(xa, wb, zero16());
    const int col = cb * 32 + r31;
    const float bv = b1 ? b1[col] : 0.0f;
    float v[16], sa = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int64_t g = s * 32 + row_of(i, half);
      v[i] = acc[i] + bv;
      if (g < S) {
        H[g * HC + col] = v[i];
        sa += v[i];
      }
    }
    if (p < h.n_heads && training) {
      sa = half_sum(sa);
      const float mean_s = sa / n;
      float sb = 0.0f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float d = (s * 32 + row_of(i, half) < S) ? v[i] - mean_s : 0.0f;
        sb += d * d;
      }
      sb = half_sum(sb);
      if (half == 0) {
        float* dst = part + ((s * h.n_heads + p) * 2) * HC + col;
        dst[0] = sa;
        dst[HC] = sb;
      }
    }
  }
}

// ---- statistics finish: one workgroup per head, thread (g, c) adds the slice partials g, g + 16, ... of column c (8 in
// flight), the sixteen group sums are added in group order (fp64).  Forward: S = sum of slice sums, Q = sum of centred
// squares, W = sum S_i^2 / n_i -> mean, biased variance (Chan's combination, bn.hip: bn_finish_centred), running
// statistics; backward: sum dz, sum dz x^.  (The first version had every wave of lin2 / bwd2 combine the partials itself:
// n_slices^2 loads of the same lines -- 31 us at 72 slices, 99 us at 287.)
constexpr int FIN_G = 16;

template <int NV>
__device__ __forceinline__ void fin_sums(const float* __restrict__ part, int n_heads, int p, int64_t S, double (&out)[NV],
                                         double (*red)[NV][HC]) {
  const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int64_t n_slices = (S + 31) / 32;
  const double inv_last = 1.0 / (double)(S - (n_slices - 1) * 32);
  double acc[3] = {0.0, 0.0, 0.0};
  constexpr int NB = 8;
  for (int64_t j0 = g; j0 < n_slices; j0 += FIN_G * NB) {
    float a[NB], b[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const int64_t j = j0 + (int64_t)u * FIN_G;
      const float* src = part + ((j * n_heads + p) * 2) * HC + c;
      a[u] = j < n_slices ? src[0] : 0.0f;
      b[u] = j < n_slices ? src[HC] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const int64_t j = j0 + (int64_t)u * FIN_G;
      if (j < n_slices) {
        acc[0] += (double)a[u];
        acc[1] += (double)b[u];
        if (NV == 3) acc[2] += (double)a[u] * (double)a[u] * (j + 1 < n_slices ? 1.0 / 32.0 : inv_last);
      }
    }
  }
#pragma unroll
  for (int q = 0; q < NV; ++q) red[g][q][c] = acc[q];
  __syncthreads();
#pragma unroll
  for (int q = 0; q < NV; ++q) {
    double t = red[0][q][c];
#pragma unroll
    for (int w = 1; w < FIN_G; ++w) t += red[w][q][c];
    out[q] = t;
  }
}

__global__ __launch_bounds__(64 * FIN_G) void heads_fin_fwd_kernel(const wsis_heads h, int64_t S, const float* __restrict__ part,
                                                                    float* __restrict__ saved, float eps, float momentum,
                                                                    int training) {
  __shared__ double red[FIN_G][3][HC];
  const int p = blockIdx.x, c = threadIdx.x & 63;
  float mean, var;
  if (training) {
    double t[3];
    fin_sums<3>(part, h.n_heads, p, S, t, red);
    const double n = (double)S, mu = t[0] / n;
    double v = (t[1] + (t[2] - n * mu * mu)) / n;
    if (v < 0.0) v = 0.0;
    mean = (float)mu;
    var = (float)v;
    if (threadIdx.x < HC && h.running_mean[p]) {
      const double unb = n > 1 ? v * n / (n - 1) : v;
      h.running_mean[p][c] = (float)((1.0 - momentum) * h.running_mean[p][c] + momentum * mu);
      h.running_var[p][c] = (float)((1.0 - momentum) * h.running_var[p][c] + momentum * unb);
    }
  } else {
    mean = h.running_mean[p][c];
    var = h.running_var[p][c];
  }
  if (threadIdx.x < HC) {
    saved[(p * 2 + 0) * HC + c] = mean;
    saved[(p * 2 + 1) * HC + c] = 1.0f / sqrtf(var + eps);
  }
}

// sums[p][0][c] = mean of dz, sums[p][1][c] = mean of dz x^ (zeros in eval mode: running statistics are constants);
// d beta = sum dz, d gamma = sum dz x^ in both modes
__global__ __launch_bounds__(64 * FIN_G) void heads_fin_bwd_kernel(const wsis_heads h, int64_t S, const float* __restrict__ partb,
                                                                    float* __restrict__ sums, int training) {
  __shared__ double red[FIN_G][2][HC];
  const int p = blockIdx.x, c = threadIdx.x & 63;
  double t[2];
  fin_sums<2>(partb, h.n_heads, p, S, t, red);
  if (threadIdx.x < HC) {
    if (h.dbeta[p]) h.dbeta[p][c] = (float)t[0];
    if (h.dgamma[p]) h.dgamma[p][c] = (float)t[1];
    sums[(p * 2 + 0) * HC + c] = training ? (float)(t[0] / (double)S) : 0.0f;
    sums[(p * 2 + 1) * HC + c] = training ? (float)(t[1] / (double)S) : 0.0f;
  }
}

// ---- forward 2: BatchNorm (+ running statistics) + ReLU + second Linear -------------------------------------------
__global__ __launch_bounds__(256) void heads_lin2_kernel(const wsis_heads h, int64_t S, const float* __restrict__ saved) {
  __shared__ float coef[4][3][HC];      // mean, gamma * rstd, beta per wave
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r31 = lane & 31, half = lane >> 5;
  const int p = blockIdx.y * 4 + wave;
  if (p >= h.n_heads) return;
  const int64_t s = blockIdx.x;
  {   // lane = column
    const int c = lane;
    coef[wave][0][c] = saved[(p * 2 + 0) * HC + c];
    coef[wave][1][c] = h.gamma[p][c] * saved[(p * 2 + 1) * HC + c];
    coef[wave][2][c] = h.beta[p][c];
  }
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const int64_t row = s * 32 + r31;
  float a[2][16];
  load_frag(h.hidden[p] + row * HC, row < S, half, a);
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int ch = c * 32 + half * 16 + i;
      const float z = __builtin_fmaf(a[c][i] - coef[wave][0][ch], coef[wave][1][ch], coef[wave][2][ch]);
      a[c][i] = row < S ? fmaxf(z, 0.0f) : 0.0f;
    }
  const int cout = h.cout[p];
  float wb[2][16];
  load_frag(h.W2[p] + r31 * HC, r31 < cout, half, wb);
  f32x16 acc = mma64(a, wb, zero16());
  if (r31 < cout) {
    const float bv = h.b2[p] ? h.b2[p][r31] : 0.0f;
    float* Y = h.out[p];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int64_t g = s * 32 + row_of(i, half);
      if (g < S) Y[g * cout + r31] = acc[i] + bv;
    }
  }
}

// ---- backward 1: through the second Linear and the ReLU; BatchNorm-backward slice partials; dW2 / db2 partials ------
__global__ __launch_bounds__(256) void heads_bwd1_kernel(const wsis_heads h, int64_t S, const float* __restrict__ saved,
                                                         float* __restrict__ dZ, float* __restrict__ partb,
                                                         float* __restrict__ partW2, int n_part) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r31 = lane & 31, half = lane >> 5;
  const int p = blockIdx.y * 4 + wave;
  if (p >= h.n_heads) return;
  const int cout = h.cout[p];
  const float* dY = h.dout[p];
  const float* H = h.hidden[p];
  const float* W2 = h.W2[p];
  float* dz_out = dZ + (int64_t)p * S * HC;
  const int64_t n_slices = (S + 31) / 32;
  // column constants of this lane (columns r31 and 32 + r31)
  float mean[2], rstd[2], sc[2], bt[2];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    const int col = cb * 32 + r31;
    mean[cb] = saved[(p * 2 + 0) * HC + col];
    rstd[cb] = saved[(p * 2 + 1) * HC + col];
    sc[cb] = h.gamma[p][col] * rstd[cb];
    bt[cb] = h.beta[p][col];
  }
  // B side of dA = dY W2: b_i[cb] = W2[k = half*16 + i][cb*32 + r31]
  float w2b[2][16];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int k = half * 16 + i;
      w2b[cb][i] = (dY && k < cout) ? W2[k * HC + cb * 32 + r31] : 0.0f;
    }
  f32x16 accw[2] = {zero16(), zero16()};
  float db = 0.0f;
  for (int64_t s = blockIdx.x; s < n_slices; s += n_part) {
    const int64_t row = s * 32 + r31;
    float ya[16], yt[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int k = half * 16 + i;
      ya[i] = (dY && row < S && k < cout) ? dY[row * cout + k] : 0.0f;            // A side: lane = row
      const int64_t g = s * 32 + row_of(i, half);
      yt[i] = (dY && g < S && r31 < cout) ? dY[g * cout + r31] : 0.0f;            // transposed: lane = output column
      db += yt[i];
    }
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      f32x16 dA = zero16();
#pragma unroll
      for (int i = 0; i < 16; ++i) dA = __builtin_amdgcn_mfma_f32_32x32x2f32(ya[i], w2b[cb][i], dA, 0, 0, 0);
      const int col = cb * 32 + r31;
      float act[16], sa = 0.0f, sb = 0.0f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int64_t g = s * 32 + row_of(i, half);
        const bool ok = g < S;
        const float hv = ok ? H[g * HC + col] : 0.0f;
        const float z = __builtin_fmaf(hv - mean[cb], sc[cb], bt[cb]);      // the forward's expression: same mask
        const float xh = (hv - mean[cb]) * rstd[cb];
        const float dz = (ok && z > 0.0f) ? dA[i] : 0.0f;
        act[i] = ok ? fmaxf(z, 0.0f) : 0.0f;
        if (ok) dz_out[g * HC + col] = dz;
        sa += dz;
        sb += dz * xh;
      }
      sa = half_sum(sa);
      sb = half_sum(sb);
      if (half == 0) {
        float* dst = partb + ((s * h.n_heads + p) * 2) * HC + col;
        dst[0] = sa;
        dst[HC] = sb;
      }
      // dW2[m][col] += sum_rows dY[row][m] * act[row][col]: the activations in the C/D register order ARE the B side
#pragma unroll
      for (int i = 0; i < 16; ++i) accw[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(yt[i], act[i], accw[cb], 0, 0, 0);
    }
  }
  float* dst = partW2 + ((int64_t)blockIdx.x * h.n_heads + p) * W2P;
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
#pragma unroll
    for (int i = 0; i < 16; ++i) dst[row_of(i, half) * HC + cb * 32 + r31] = accw[cb][i];
  db = half_sum(db);
  if (half == 0) dst[32 * HC + r31] = db;
}

// ---- backward 2: BatchNorm backward, dx, dW1 / db1 partials -------------------------------------------------------------
constexpr int TP = HC + 4;       // row pitch of the LDS tiles (floats)
struct Bwd2Lds {
  float tile[HB][32][TP];        // dH of a block (column layout in, row layout out), then its dx partial
};

__global__ __launch_bounds__(64 * HB) void heads_bwd2_kernel(const wsis_heads h, const float* __restrict__ X, int64_t S,
                                                              const float* __restrict__ saved, const float* __restrict__ dZ,
                                                              const float* __restrict__ sums, float* __restrict__ dX,
                                                              float* __restrict__ partW1, int n_part) {
  extern __shared__ unsigned char lds_raw[];
  Bwd2Lds& L = *reinterpret_cast<Bwd2Lds*>(lds_raw);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r31 = lane & 31, half = lane >> 5;
  const int nb = h.n_heads + h.n_lin;
  const int64_t n_slices = (S + 31) / 32;
  const int p = wave;                       // one block per wave
  const bool live = p < nb, is_head = p < h.n_heads;
  float mean[2] = {0.f, 0.f}, rstd[2] = {0.f, 0.f}, gs[2] = {0.f, 0.f}, m1[2] = {0.f, 0.f}, m2[2] = {0.f, 0.f};
  if (live && is_head) {
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      const int col = cb * 32 + r31;
      mean[cb] = saved[(p * 2 + 0) * HC + col];
      rstd[cb] = saved[(p * 2 + 1) * HC + col];
      gs[cb] = h.gamma[p][col] * rstd[cb];
      m1[cb] = sums[(p * 2 + 0) * HC + col];
      m2[cb] = sums[(p * 2 + 1) * HC + col];
    }
  }
  const float* src = live ? (is_head ? dZ + (int64_t)p * S * HC : h.dout[p]) : nullptr;
  const float* H = (live && is_head) ? h.hidden[p] : nullptr;
  const float* W1 = live ? h.W1[p] : nullptr;
  f32x16 accw[2][2] = {{zero16(), zero16()}, {zero16(), zero16()}};       // dW1[kb*32 + row][cb*32 + col]
  float db[2] = {0.0f, 0.0f};
  for (int64_t s = blockIdx.x; s < n_slices; s += n_part) {
    if (live) {
      // dH of the slice in the column layout (lane = column, registers = rows), kept for dW1 and written to LDS
      float dh[2][16], xt[2][16];
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        const int col = cb * 32 + r31;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int64_t g = s * 32 + row_of(i, half);
          const bool ok = g < S;
          float v = (ok && src) ? src[g * HC + col] : 0.0f;
          if (is_head) {
            const float hv = ok ? H[g * HC + col] : 0.0f;
            const float xh = (hv - mean[cb]) * rstd[cb];
            v = ok ? gs[cb] * ((v - m1[cb]) - xh * m2[cb]) : 0.0f;      // (eval mode: m1 = m2 = 0)
          }
          dh[cb][i] = v;
          db[cb] += v;
          L.tile[wave][row_of(i, half)][col] = v;
          xt[cb][i] = ok ? X[g * HC + col] : 0.0f;
        }
      }
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
          for (int i = 0; i < 16; ++i)
            accw[kb][cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(dh[kb][i], xt[cb][i], accw[kb][cb], 0, 0, 0);
      // dx partial of this block: dH (row layout from LDS) @ W1 (k = hidden index = W1 row)
      __builtin_amdgcn_wave_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      float a[2][16];
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 v = *reinterpret_cast<const float4*>(&L.tile[wave][r31][c * 32 + half * 16 + q * 4]);
          a[c][q * 4 + 0] = v.x; a[c][q * 4 + 1] = v.y; a[c][q * 4 + 2] = v.z; a[c][q * 4 + 3] = v.w;
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
#pragma unroll 1
      for (int cb = 0; cb < 2; ++cb) {      // (not unrolled: both halves' weight loads hoisted together spill)
        float wb[2][16];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int i = 0; i < 16; ++i) wb[c][i] = W1[(c * 32 + half * 16 + i) * HC + cb * 32 + r31];
        const f32x16 dxp = mma64(a, wb, zero16());
#pragma unroll
        for (int i = 0; i < 16; ++i) L.tile[wave][row_of(i, half)][cb * 32 + r31] = dxp[i];
      }
    }
    __syncthreads();
    // dx of the slice: the blocks' partials added in block order
    for (int e = threadIdx.x; e < 32 * HC; e += blockDim.x) {
      const int rr = e / HC, col = e % HC;
      const int64_t g = s * 32 + rr;
      float v = 0.0f;
      for (int q = 0; q < nb; ++q) v += L.tile[q][rr][col];
      if (g < S) dX[g * HC + col] = v;
    }
    __syncthreads();
  }
  if (live) {
    float* dst = partW1 + ((int64_t)blockIdx.x * nb + p) * W1P;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int i = 0; i < 16; ++i) dst[(kb * 32 + row_of(i, half)) * HC + cb * 32 + r31] = accw[kb][cb][i];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      const float v = half_sum(db[cb]);
      if (half == 0) dst[HC * HC + cb * 32 + r31] = v;
    }
  }
}

// ---- backward 3: the partials of every slice set, added in set order, into the gradient tensors -----------------------
// eight lanes per output float split the sets (lane l: l, l + 8, ...), their sums are added in lane order
__global__ __launch_bounds__(256) void heads_bwd3_kernel(const wsis_heads h, const float* __restrict__ partW1,
                                                         const float* __restrict__ partW2, int n_part) {
  __shared__ float red[8][32];
  const int el = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int nb = h.n_heads + h.n_lin;
  const int n1 = nb * W1P, n2 = h.n_heads * W2P;
  const int e = blockIdx.x * 32 + el;
  const bool ok = e < n1 + n2;
  const float* src = e < n1 ? partW1 + e : partW2 + (e - n1);
  const int64_t stride = e < n1 ? (int64_t)n1 : (int64_t)n2;
  float s = 0.0f;
  if (ok)
    for (int q = sl; q < n_part; q += 8) s += src[q * stride];
  red[sl][el] = s;
  __syncthreads();
  if (sl == 0 && ok) {
#pragma unroll
    for (int l = 1; l < 8; ++l) s += red[l][el];
    if (e < n1) {
      const int p = e / W1P, r = e % W1P;
      if (r < HC * HC) {
        if (h.dW1[p]) h.dW1[p][r] = s;
      } else if (p < h.n_heads && h.db1[p]) {
        h.db1[p][r - HC * HC] = s;
      }
    } else {
      const int p = (e - n1) / W2P, r = (e - n1) % W2P, cout = h.cout[p];
      if (r < 32 * HC) {
        if (r / HC < cout && h.dW2[p]) h.dW2[p][r] = s;          // [cout, 64] row-major: the same offset
      } else if (r - 32 * HC < cout && h.db2[p]) {
        h.db2[p][r - 32 * HC] = s;
      }
    }
  }
}

int n_part_of(int64_t S) {
  const int64_t n_slices = (S + 31) / 32;
  return (int)(n_slices < MAX_PART ? n_slices : MAX_PART);
}

int check_desc(const wsis_heads* h, bool backward) {
  WSIS_REQUIRE(h != nullptr, "descriptor is NULL");
  WSIS_REQUIRE(h->n_heads >= 0 && h->n_lin >= 0 && h->n_heads + h->n_lin >= 1 && h->n_heads + h->n_lin <= HB,
               "1..WSIS_HEADS_MAX blocks");
  for (int p = 0; p < h->n_heads + h->n_lin; ++p) {
    WSIS_REQUIRE(h->W1[p] && h->hidden[p], "W1 / hidden buffer missing");
    if (p < h->n_heads) {
      WSIS_REQUIRE(h->cout[p] >= 1 && h->cout[p] <= 32, "1 <= cout <= 32");
      WSIS_REQUIRE(h->gamma[p] && h->beta[p] && h->W2[p], "head parameter missing");
      WSIS_REQUIRE(backward || h->out[p], "output buffer missing");
    }
  }
  return WSIS_OK;
}

}  // namespace

extern "C" int64_t wsis_heads_workspace_bytes(int64_t S, int32_t n_heads, int32_t n_lin) {
  if (S < 1 || n_heads < 0 || n_lin < 0 || n_heads + n_lin > HB) return -1;
  const int64_t n_slices = (S + 31) / 32, np = n_part_of(S);
  const int64_t part = n_slices * n_heads * 2 * HC;                  // forward partials / backward partials
  const int64_t dz = (int64_t)n_heads * S * HC;
  const int64_t pw = np * ((int64_t)(n_heads + n_lin) * W1P + (int64_t)n_heads * W2P);
  return (part + dz + pw + (int64_t)n_heads * 2 * HC) * (int64_t)sizeof(float) + 1024;
}

extern "C" int wsis_heads_fwd(const wsis_heads* h, const float* d_x, int64_t S, float eps, float momentum, int32_t training,
                              float* d_saved, void* d_ws, int64_t ws_bytes, void* stream) {
  if (check_desc(h, false) != WSIS_OK) return WSIS_ERR_ARG;
  WSIS_REQUIRE(d_x && S >= 1 && S < ((int64_t)1 << 24), "x missing or S out of range");
  WSIS_REQUIRE(h->n_heads == 0 || d_saved, "saved statistics buffer missing");
  WSIS_REQUIRE(ws_bytes >= wsis_heads_workspace_bytes(S, h->n_heads, h->n_lin) && d_ws, "workspace too small");
  for (int p = 0; p < h->n_heads; ++p)
    WSIS_REQUIRE(training || (h->running_mean[p] && h->running_var[p]), "eval mode needs running statistics");
  hipStream_t st = as_stream(stream);
  const int nb = h->n_heads + h->n_lin;
  const unsigned n_slices = (unsigned)((S + 31) / 32);
  float* part = static_cast<float*>(d_ws);
  hipLaunchKernelGGL(heads_lin1_kernel, dim3(n_slices, (unsigned)((nb + 3) / 4)), dim3(256), 0, st, *h, d_x, S, part,
                     (int)training);
  WSIS_LAUNCH_CHECK();
  if (h->n_heads > 0) {
    hipLaunchKernelGGL(heads_fin_fwd_kernel, dim3((unsigned)h->n_heads), dim3(64 * FIN_G), 0, st, *h, S, part, d_saved, eps,
                       momentum, (int)training);
    WSIS_LAUNCH_CHECK();
    hipLaunchKernelGGL(heads_lin2_kernel, dim3(n_slices, (unsigned)((h->n_heads + 3) / 4)), dim3(256), 0, st, *h, S, d_saved);
    WSIS_LAUNCH_CHECK();
  }
  return WSIS_OK;
}

extern "C" int wsis_heads_bwd(const wsis_heads* h, const float* d_x, int64_t S, int32_t training, const float* d_saved,
                              float* d_dx, void* d_ws, int64_t ws_bytes, void* stream) {
  if (check_desc(h, true) != WSIS_OK) return WSIS_ERR_ARG;
  WSIS_REQUIRE(d_x && d_dx && S >= 1 && S < ((int64_t)1 << 24), "x / dx missing or S out of range");
  WSIS_REQUIRE(h->n_heads == 0 || d_saved, "saved statistics buffer missing");
  WSIS_REQUIRE(ws_bytes >= wsis_heads_workspace_bytes(S, h->n_heads, h->n_lin) && d_ws, "workspace too small");
  hipStream_t st = as_stream(stream);
  const int nb = h->n_heads + h->n_lin, np = n_part_of(S);
  const int64_t n_slices = (S + 31) / 32;
  float* partb = static_cast<float*>(d_ws);
  float* dZ = partb + n_slices * h->n_heads * 2 * HC;
  float* partW1 = dZ + (int64_t)h->n_heads * S * HC;
  float* partW2 = partW1 + (int64_t)np * nb * W1P;
  float* sums = partW2 + (int64_t)np * h->n_heads * W2P;
  if (h->n_heads > 0) {
    hipLaunchKernelGGL(heads_bwd1_kernel, dim3((unsigned)np, (unsigned)((h->n_heads + 3) / 4)), dim3(256), 0, st, *h, S,
                       d_saved, dZ, partb, partW2, np);
    WSIS_LAUNCH_CHECK();
    hipLaunchKernelGGL(heads_fin_bwd_kernel, dim3((unsigned)h->n_heads), dim3(64 * FIN_G), 0, st, *h, S, partb, sums,
                       (int)training);
    WSIS_LAUNCH_CHECK();
  }
  static bool attr_set = false;       // (one process per GPU: include/wsis_hip.h)
  if (!attr_set) {
    WSIS_HIP_CHECK(hipFuncSetAttribute((const void*)heads_bwd2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)sizeof(Bwd2Lds)));
    attr_set = true;
  }
  hipLaunchKernelGGL(heads_bwd2_kernel, dim3((unsigned)np), dim3(64 * HB), sizeof(Bwd2Lds), st, *h, d_x, S, d_saved, dZ, sums,
                     d_dx, partW1, np);
  WSIS_LAUNCH_CHECK();
  const int total = nb * W1P + h->n_heads * W2P;
  hipLaunchKernelGGL(heads_bwd3_kernel, dim3((unsigned)((total + 31) / 32)), dim3(256), 0, st, *h, partW1, partW2, np);
  WSIS_LAUNCH_CHECK();
  return WSIS_OK;
}
