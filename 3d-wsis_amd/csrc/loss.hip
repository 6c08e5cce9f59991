// Point-level semantic loss of MultiTaskLoss (losses_3D_WSIS.py:52-67 of the reference: CrossEntropyLoss with
// ignore_index + dice_loss_multi_classes on the softmax of the kept rows) as two passes over the [N, C] scores:
//   loss = sum_valid(-log_softmax(x)[y]) / n_valid + mean_c(1 - (2 A_c + eps) / (B_c + K_c + 1e-4 + eps))
//   A_c = sum_valid p_c [y = c],  B_c = sum_valid p_c^2,  K_c = #valid rows of class c,  eps = 1e-5.
// The torch evaluation is ~40 launches over [N, 20] tensors (log_softmax, nll_loss with a one-workgroup reduction,
// softmax, a 32-MB int64 one_hot, two maskings, three column sums, and their backward nodes): ~0.7 ms per step.
// Forward: one thread per row keeps the per-class sums in registers (static indices), a workgroup reduces them with
// wave shuffles + a 4-entry LDS stage, one partial row per workgroup; a single-workgroup final kernel sums the partials
// in fp64 in a fixed order and evaluates the loss -> deterministic.  Backward: one pass, analytic gradient.
#include <cstdlib>

#include "common.h"

namespace wsis {
namespace {

constexpr int SL_CMAX = 32;                  // classes held in registers
constexpr int SL_THREADS = 256;
constexpr int SL_VALS = 3 * SL_CMAX + 2;     // A[32] B[32] K[32] ce n
constexpr int SL_TP = SL_THREADS + 4;        // row pitch of the reduction tile (floats)
constexpr int SL_MAX_BLOCKS = 512;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__global__ __launch_bounds__(SL_THREADS) void sem_loss_fwd_kernel(const float* __restrict__ x,
                                                                  const int64_t* __restrict__ y, int64_t N, int C,
                                                                  int64_t ignore, float* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) float sl_tile[];
  float A[SL_CMAX], B[SL_CMAX], K[SL_CMAX];
#pragma unroll
  for (int c = 0; c < SL_CMAX; ++c) A[c] = B[c] = K[c] = 0.0f;
  float ce = 0.0f, n = 0.0f;
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < N; r += (int64_t)gridDim.x * blockDim.x) {
    const int64_t lab = y[r];
    if (lab == ignore) continue;
    const float* row = x + r * C;
    float v[SL_CMAX];
    float m = -INFINITY, xl = 0.0f;
    // (branch-free loads -- clamped column, masked value: behind a conditional load hipcc waits for each one before it
    // issues the next)
#pragma unroll
    for (int c = 0; c < SL_CMAX; ++c) v[c] = row[c < C ? c : C - 1];
#pragma unroll
    for (int c = 0; c < SL_CMAX; ++c) {
      v[c] = c < C ? v[c] : -INFINITY;
      m = fmaxf(m, v[c]);
      if ((int64_t)c == lab && c < C) xl = v[c];
    }
    float Z = 0.0f;
#pragma unroll
    for (int c = 0; c < SL_CMAX; ++c) {
      v[c] = c < C ? expf(v[c] - m) : 0.0f;
      Z += v[c];
    }
    const float inv = 1.0f / Z, logZ = logf(Z);
#pragma unroll
    for (int c = 0; c < SL_CMAX; ++c) {
      const float p = v[c] * inv;
      const bool hit = (int64_t)c == lab;
      B[c] += p * p;
      A[c] += hit ? p : 0.0f;
      K[c] += hit ? 1.0f : 0.0f;
    }
    ce += (m + logZ) - xl;      // -log_softmax(x)[y]
    n += 1.0f;
  }
  // the 98 sums of the workgroup through an LDS tile [98][256]: every thread stores its terms in its column, thread v
  // adds row v in thread order (16-byte reads).  (98 wave butterflies were 588 ds_bpermute per wave: ~15 us of LDS
  // crossbar time per workgroup.)
#pragma unroll
  for (int c = 0; c < SL_CMAX; ++c) {
    sl_tile[c * SL_TP + threadIdx.x] = A[c];
    sl_tile[(SL_CMAX + c) * SL_TP + threadIdx.x] = B[c];
    sl_tile[(2 * SL_CMAX + c) * SL_TP + threadIdx.x] = K[c];
  }
  sl_tile[(3 * SL_CMAX) * SL_TP + threadIdx.x] = ce;
  sl_tile[(3 * SL_CMAX + 1) * SL_TP + threadIdx.x] = n;
  __syncthreads();
  if (threadIdx.x < SL_VALS) {
    const float4* row = reinterpret_cast<const float4*>(sl_tile + threadIdx.x * SL_TP);
    float s = 0.0f;
#pragma unroll 8
    for (int q = 0; q < SL_THREADS / 4; ++q) {
      const float4 v = row[q];
      s += v.x;
      s += v.y;
      s += v.z;
      s += v.w;
    }
    partial[(int64_t)blockIdx.x * SL_VALS + threadIdx.x] = s;
  }
}

// out[0] = loss, out[1] = n_valid; saved[0..C) = Num_c, saved[C..2C) = Den_c, saved[2C] = n_valid
__global__ __launch_bounds__(1024) void sem_loss_final_kernel(const float* __restrict__ partial, int nblk, int C,
                                                              float* __restrict__ out, float* __restrict__ saved) {
  __shared__ double sums[128];
  __shared__ double psub[8][128];
  // value v of partial row b: a wave reads 64 consecutive values of one row (the layout with eight consecutive threads
  // on eight different rows touched eight cache lines per load: 19 us for 200 KB).  Same partition (rows b = sub mod 8,
  // ascending) and the same pairing of the eight sub-sums as the shuffle butterfly it replaces: identical result.
  const int v = threadIdx.x & 127, sub = threadIdx.x >> 7;
  double s = 0.0;
  if (v < SL_VALS)
    for (int b = sub; b < nblk; b += 64) {      // eight loads in flight per trip, added in the same (ascending) order
      float t[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int bb = b + 8 * j;
        t[j] = bb < nblk ? partial[(int64_t)bb * SL_VALS + v] : 0.0f;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) s += (double)t[j];
    }
  psub[sub][v] = s;
  __syncthreads();
  if (sub == 0)
    sums[v] = ((psub[0][v] + psub[1][v]) + (psub[2][v] + psub[3][v])) + ((psub[4][v] + psub[5][v]) + (psub[6][v] + psub[7][v]));
  __syncthreads();
  if (threadIdx.x == 0) {
    const double eps = 1e-5;
    const double n = sums[3 * SL_CMAX + 1];
    double dice_sum = 0.0;
    for (int c = 0; c < C; ++c) {
      const double num = 2.0 * sums[c] + eps;
      const double den = sums[SL_CMAX + c] + sums[2 * SL_CMAX + c] + 1e-4 + eps;
      saved[c] = (float)num;
      saved[C + c] = (float)den;
      dice_sum += 1.0 - num / den;
    }
    saved[2 * C] = (float)n;
    out[0] = (float)(sums[3 * SL_CMAX] / n + dice_sum / (double)C);   // n == 0 -> nan, as the reference's CE
    out[1] = (float)n;
  }
}

__global__ __launch_bounds__(SL_THREADS) void sem_loss_bwd_kernel(const float* __restrict__ x,
                                                                  const int64_t* __restrict__ y, int64_t N, int C,
                                                                  int64_t ignore, const float* __restrict__ saved,
                                                                  const float* __restrict__ gout,
                                                                  float* __restrict__ dx) {
  __shared__ float s_num[SL_CMAX], s_den[SL_CMAX];
  if (threadIdx.x < SL_CMAX) {
    s_num[threadIdx.x] = threadIdx.x < C ? saved[threadIdx.x] : 0.0f;
    s_den[threadIdx.x] = threadIdx.x < C ? saved[C + threadIdx.x] : 1.0f;
  }
  __syncthreads();
  const float g = gout[0];
  const float inv_n = 1.0f / saved[2 * C];
  const float inv_c = 1.0f / (float)C;
  // a trip of the workgroup covers SL_THREADS consecutive rows; their gradient rows are written to an LDS tile of odd pitch
  // and leave as consecutive floats (with every thread storing its own row -- 80-byte stride -- a store instruction
  // touched 40 cache lines with 4 bytes each)
  __shared__ float tile[SL_THREADS * (SL_CMAX + 1)];
  const int pitch = C | 1;
  for (int64_t r0 = blockIdx.x * (int64_t)blockDim.x; r0 < N; r0 += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = r0 + threadIdx.x;
    float* const mine = tile + threadIdx.x * pitch;
    const int64_t lab = r < N ? y[r] : ignore;
    if (lab == ignore) {
      for (int c = 0; c < C; ++c) mine[c] = 0.0f;
    } else {
      const float* row = x + r * C;
      float v[SL_CMAX];
      float m = -INFINITY;
#pragma unroll
      for (int c = 0; c < SL_CMAX; ++c) v[c] = row[c < C ? c : C - 1];      // (branch-free loads, see the forward kernel)
#pragma unroll
      for (int c = 0; c < SL_CMAX; ++c) {
        v[c] = c < C ? v[c] : -INFINITY;
        m = fmaxf(m, v[c]);
      }
      float Z = 0.0f;
#pragma unroll
      for (int c = 0; c < SL_CMAX; ++c) {
        v[c] = c < C ? expf(v[c] - m) : 0.0f;
        Z += v[c];
      }
      const float inv = 1.0f / Z;
      // dL/dp_c = (1/C) (2 p_c Num_c / Den_c^2 - 2 [y = c] / Den_c);  dx_j = p_j (gp_j - sum_c gp_c p_c) + (p_j - [y=j]) / n
      float gp[SL_CMAX];
      float dot = 0.0f;
#pragma unroll
      for (int c = 0; c < SL_CMAX; ++c) {
        const float p = v[c] * inv;
        v[c] = p;
        const float den = s_den[c];
        gp[c] = c < C ? inv_c * (2.0f * p * s_num[c] / (den * den) - (((int64_t)c == lab) ? 2.0f / den : 0.0f)) : 0.0f;
        dot += gp[c] * p;
      }
#pragma unroll
      for (int c = 0; c < SL_CMAX; ++c)
        if (c < C) mine[c] = g * (v[c] * (gp[c] - dot) + (v[c] - (((int64_t)c == lab) ? 1.0f : 0.0f)) * inv_n);
    }
    __syncthreads();
    const int64_t rows = N - r0 < (int64_t)blockDim.x ? N - r0 : (int64_t)blockDim.x;
    float* const dst = dx + r0 * C;
    for (int64_t f = threadIdx.x; f < rows * C; f += blockDim.x) {
      const int rr = (int)(f / C), cc = (int)(f - (int64_t)rr * C);
      dst[f] = tile[rr * pitch + cc];
    }
    __syncthreads();
  }
}

// ---- superpoint regression terms of MultiTaskLoss (losses_3D_WSIS.py:79-96 offset L1 + cosine, :113-127 occupancy and
// instance-size L1): rows count when both superpoint labels differ from ignore_label.  One workgroup walks the S rows
// (S ~ 1-10 k), fp32 per-thread sums folded in fp64 in a fixed order; ~80 small torch launches otherwise.
//   out[0] = sum_valid |p - g|_1 / (n + 1e-6)          out[1] = sum_valid -(g/(|g|+1e-8)) . (p/(|p|+1e-8)) / (n + 1e-6)
//   out[2] = sum_valid |occ_p - occ_g| / n              out[3] = sum_valid |size_p - size_g| / n      out[4] = n
// Dropped rows are skipped by selection, not multiplied by 0 (the log voxel count of an unlabelled superpoint is -inf).
constexpr int SR_THREADS = 1024;

__global__ __launch_bounds__(SR_THREADS) void sp_reg_fwd_kernel(
    const float* __restrict__ p_off, const float* __restrict__ g_off, const float* __restrict__ p_occ,
    const float* __restrict__ g_occ, const float* __restrict__ p_size, const float* __restrict__ g_size,
    const int64_t* __restrict__ sem, const int64_t* __restrict__ ins, int64_t S, int64_t ignore,
    float* __restrict__ out) {
  __shared__ double sh[SR_THREADS / 64][5];
  float a[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  for (int64_t r = threadIdx.x; r < S; r += SR_THREADS) {
    if (sem[r] == ignore || ins[r] == ignore) continue;
    float l1 = 0.f, pp = 0.f, gg = 0.f;
    float p[3], g[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      p[k] = p_off[r * 3 + k];
      g[k] = g_off[r * 3 + k];
      l1 += fabsf(p[k] - g[k]);
      pp += p[k] * p[k];
      gg += g[k] * g[k];
    }
    const float ip = 1.0f / (sqrtf(pp) + 1e-8f), ig = 1.0f / (sqrtf(gg) + 1e-8f);
    float dot = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) dot += (g[k] * ig) * (p[k] * ip);
    a[0] += l1;
    a[1] += -dot;
    a[2] += fabsf(p_occ[r] - g_occ[r]);
    a[3] += fabsf(p_size[r] - g_size[r]);
    a[4] += 1.0f;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < 5; ++q) {
    double v = (double)a[q];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    if (lane == 0) sh[wave][q] = v;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double t[5] = {0, 0, 0, 0, 0};
    for (int w = 0; w < SR_THREADS / 64; ++w)
      for (int q = 0; q < 5; ++q) t[q] += sh[w][q];
    const double n = t[4];
    out[0] = (float)(t[0] / (n + 1e-6));
    out[1] = (float)(t[1] / (n + 1e-6));
    out[2] = (float)(t[2] / n);          // n == 0 -> nan, as nn.L1Loss on an empty selection
    out[3] = (float)(t[3] / n);
    out[4] = (float)n;
  }
}

__device__ __forceinline__ float sgn(float v) { return (v > 0.f) ? 1.f : ((v < 0.f) ? -1.f : 0.f); }

// d_off [S,3], d_occ [S], d_size [S] from the four upstream gradients (device scalars) and n = out[4]
__global__ void sp_reg_bwd_kernel(const float* __restrict__ p_off, const float* __restrict__ g_off,
                                  const float* __restrict__ p_occ, const float* __restrict__ g_occ,
                                  const float* __restrict__ p_size, const float* __restrict__ g_size,
                                  const int64_t* __restrict__ sem, const int64_t* __restrict__ ins, int64_t S,
                                  int64_t ignore, const float* __restrict__ out, const float* __restrict__ g0,
                                  const float* __restrict__ g1, const float* __restrict__ g2,
                                  const float* __restrict__ g3, float* __restrict__ d_off,
                                  float* __restrict__ d_occ, float* __restrict__ d_size) {
  const float n = out[4];
  const float w01 = 1.0f / (n + 1e-6f), w23 = 1.0f / n;
  const float c0 = g0[0] * w01, c1 = g1[0] * w01, c2 = g2[0] * w23, c3 = g3[0] * w23;
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < S; r += (int64_t)gridDim.x * blockDim.x) {
    if (sem[r] == ignore || ins[r] == ignore) {
      d_off[r * 3] = d_off[r * 3 + 1] = d_off[r * 3 + 2] = 0.f;
      d_occ[r] = 0.f;
      d_size[r] = 0.f;
      continue;
    }
    float p[3], g[3], pp = 0.f, gg = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      p[k] = p_off[r * 3 + k];
      g[k] = g_off[r * 3 + k];
      pp += p[k] * p[k];
      gg += g[k] * g[k];
    }
    const float np_ = sqrtf(pp), ip = 1.0f / (np_ + 1e-8f), ig = 1.0f / (sqrtf(gg) + 1e-8f);
    float gdp = 0.f;                               // (g/(|g|+eps)) . p
#pragma unroll
    for (int k = 0; k < 3; ++k) gdp += g[k] * ig * p[k];
    // d/dp_k [ p_k/(|p|+eps) ] = 1/(|p|+eps) - p p^T / (|p| (|p|+eps)^2); torch's norm backward is 0 at |p| = 0
    const float tail = np_ > 0.f ? gdp * ip * ip / np_ : 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k)
      d_off[r * 3 + k] = c0 * sgn(p[k] - g[k]) - c1 * (g[k] * ig * ip - tail * p[k]);
    d_occ[r] = c2 * sgn(p_occ[r] - g_occ[r]);
    d_size[r] = c3 * sgn(p_size[r] - g_size[r]);
  }
}

// ---- superpoint semantic term (losses_3D_WSIS.py:72-74: CrossEntropyLoss(ignore_index) on the [S, C] superpoint scores,
// logged with scores.sum()) and the weighted sum of all terms (:130-151).  torch runs log_softmax + nll_loss + sum forward
// and two launches backward on 2,289 rows, and one launch per `loss = loss + term`; here one workgroup walks the rows
// (fp32 per row, folded in fp64 in a fixed order) and one thread adds the terms in the reference's order.
//   out[0] = sum_kept (logsumexp(row) - row[label]) / n_kept     out[1] = sum of all scores     out[2] = n_kept
// one row per thread, 256 rows per workgroup; workgroup sums (fp64, fixed order) go to `partial` write-through, the last
// workgroup to arrive (ticket in the caller's sync slot, self-resetting) adds them in workgroup order.  As ONE workgroup
// walking all rows the kernel was bound by one compute unit's expf rate: 2,289 x 20 calls = 23 us.
constexpr int CE_THREADS = 256;

__global__ __launch_bounds__(CE_THREADS) void sp_ce_fwd_kernel(const float* __restrict__ scores, const int64_t* __restrict__ labels,
                                                                int64_t S, int C, int64_t ignore, double* __restrict__ partial,
                                                                unsigned* __restrict__ ticket, float* __restrict__ out) {
  __shared__ double sh[CE_THREADS / 64][3];
  __shared__ int s_last;
  float a[3] = {0.f, 0.f, 0.f};
  const int64_t r = (int64_t)blockIdx.x * CE_THREADS + threadIdx.x;
  if (r < S) {
    const float* row = scores + r * C;
    float v[32];                           // the row once, all loads in flight (branch-free: clamped column, masked value)
#pragma unroll
    for (int c = 0; c < 32; ++c) v[c] = row[c < C ? c : C - 1];
#pragma unroll
    for (int c = 0; c < 32; ++c) v[c] = c < C ? v[c] : -INFINITY;
    float mx = v[0], sum = 0.f;
#pragma unroll
    for (int c = 1; c < 32; ++c) mx = fmaxf(mx, v[c]);
    float se = 0.f;
#pragma unroll
    for (int c = 0; c < 32; ++c) {
      se += c < C ? expf(v[c] - mx) : 0.0f;
      sum += c < C ? v[c] : 0.0f;
    }
    a[1] = sum;
    const int64_t lab = labels[r];
    if (lab != ignore && lab >= 0 && lab < C) {
      a[0] = (logf(se) + mx) - row[lab];
      a[2] = 1.0f;
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    double v = (double)a[q];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    if (lane == 0) sh[wave][q] = v;
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    double t = 0.0;
    for (int w = 0; w < CE_THREADS / 64; ++w) t += sh[w][threadIdx.x];
    st_sc1(partial + (int64_t)blockIdx.x * 3 + threadIdx.x, t);
  }
  wait_stores_left();
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned t = atomicAdd(ticket, 1u);
    s_last = t == gridDim.x - 1;
    if (s_last) *ticket = 0u;              // self-cleaning: ready for the next launch on this stream
  }
  __syncthreads();
  if (!s_last || threadIdx.x != 0) return;
  double t[3] = {0, 0, 0};
  for (unsigned g = 0; g < gridDim.x; ++g)
    for (int q = 0; q < 3; ++q) t[q] += ld_sc1(partial + (int64_t)g * 3 + q);
  out[0] = (float)(t[0] / t[2]);        // n_kept == 0 -> nan, as torch
  out[1] = (float)t[1];
  out[2] = (float)t[2];
}

// d scores[r, c] = g * (softmax(row)[c] - [c == label]) / n_kept for kept rows, 0 for the others
__global__ void sp_ce_bwd_kernel(const float* __restrict__ scores, const int64_t* __restrict__ labels, int64_t S, int C,
                                 int64_t ignore, const float* __restrict__ out, const float* __restrict__ g,
                                 float* __restrict__ d) {
  const float w = g[0] / out[2];
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < S; r += (int64_t)gridDim.x * blockDim.x) {
    const float* row = scores + r * C;
    const int64_t lab = labels[r];
    if (lab == ignore || lab < 0 || lab >= C) {
      for (int c = 0; c < C; ++c) d[r * C + c] = 0.f;
      continue;
    }
    float v[32];                           // the row once (branch-free loads, C <= 32)
#pragma unroll
    for (int c = 0; c < 32; ++c) v[c] = row[c < C ? c : C - 1];
    float mx = v[0];
#pragma unroll
    for (int c = 1; c < 32; ++c) mx = fmaxf(mx, c < C ? v[c] : v[0]);
    float se = 0.f;
#pragma unroll
    for (int c = 0; c < 32; ++c) {
      v[c] = c < C ? expf(v[c] - mx) : 0.0f;
      se += v[c];
    }
    const float inv = 1.0f / se;
#pragma unroll
    for (int c = 0; c < 32; ++c)
      if (c < C) d[r * C + c] = w * (v[c] * inv - ((int64_t)c == lab ? 1.0f : 0.0f));
  }
}

// loss = t0 + t1 + ... in order; a term with `paired` bit i set is first added to its successor: (t_i + t_{i+1})
__global__ void loss_sum_kernel(const float* t0, const float* t1, const float* t2, const float* t3, const float* t4,
                                const float* t5, const float* t6, const float* t7, int n, unsigned paired, float* out) {
  const float* t[8] = {t0, t1, t2, t3, t4, t5, t6, t7};
  float acc = 0.0f;
  bool first = true;
  for (int i = 0; i < n; ++i) {
    float v = t[i][0];
    if ((paired >> i) & 1u) {
      v = v + t[i + 1][0];
      ++i;
    }
    acc = first ? v : acc + v;
    first = false;
  }
  out[0] = acc;
}

// ---- discriminative (pull / push / regularisation) loss of one scene's superpoint embeddings
// (losses_3D_WSIS.py:157-230): instances in n_slots <= 64 fixed slots (slot = instance id, bound known on the host),
// S <= 4096 rows of D = 7 features, everything in one workgroup: rows and slots staged in LDS, instance sums by
// (row chunk, slot, feature) threads walking their chunk in order + a chunk-ordered combine (deterministic), the three terms folded in fp64.
//   loss = l_var + l_dist + 0.001 l_reg,  l_var = (1/n) sum_a (1/c_a) sum_{i in a} max(|x_i - mu_a|_2 - 0.1, 0)^2,
//   l_dist = sum_{a != b} max(3 - |mu_a - mu_b|_1, 0)^2 / (n (n - 1)),  l_reg = sum_a |mu_a|_2.
// saved: mu [64][8] (column 7 = member count), k [S] = 2 max(t - dv, 0) / t per row, n.  The backward kernel is the
// analytic gradient (checked against autograd in fp64 on the host before it was written).
constexpr int DL_D = 7, DL_SLOTS = 64, DL_ROWS = 4096, DL_THREADS = 1024;
// the whole scene is staged in LDS: xs (112 KB) + kk (16 KB) + slot (8 KB) + per-slot sums -- this needs gfx950's
// 160 KB of LDS per CU (a 64 KB-LDS target would have to stream the rows instead)
#if !defined(__gfx950__) && defined(__HIP_DEVICE_COMPILE__)
#error "csrc/loss.hip is written for gfx950 (160 KB LDS per CU)"
#endif
static_assert(DL_ROWS * DL_D * 4 + DL_ROWS * 4 + DL_ROWS * 2 + DL_SLOTS * 8 * 4 * 2 <= 160 * 1024,
              "disc_loss kernels stage the scene in LDS: the bound follows the 160 KB of a gfx950 CU");

// row chunks of the per-instance walks: as many as fit 1024 threads at I*8 threads per chunk (<= 16)
__device__ __forceinline__ int dl_chunks(int I) {
  const int c = DL_THREADS / (I * 8);
  return c > 16 ? 16 : (c < 1 ? 1 : c);
}

struct DlParams {
  float delta_v, delta_d, p_var, p_dist, p_reg;
};

__device__ __forceinline__ void dl_stage(const float* __restrict__ x, const int64_t* __restrict__ ins,
                                         const int64_t* __restrict__ sem, int S, int I, int64_t ignore, float* xs,
                                         short* slot) {
  // every load of the thread in flight before its first LDS store (a load -> store loop runs one memory round trip per
  // iteration: 16 of them for 2,289 rows, most of the kernel's 30 us); branch-free: clamped index, masked store
  constexpr int NX = DL_ROWS * DL_D / DL_THREADS, NR = DL_ROWS / DL_THREADS;
  const int n = S * DL_D;
  float v[NX];
#pragma unroll
  for (int u = 0; u < NX; ++u) {
    const int t = threadIdx.x + u * DL_THREADS;
    v[u] = x[t < n ? t : 0];
  }
  int64_t a[NR], b[NR];
#pragma unroll
  for (int u = 0; u < NR; ++u) {
    const int r = threadIdx.x + u * DL_THREADS;
    a[u] = ins[r < S ? r : 0];
    b[u] = sem[r < S ? r : 0];
  }
#pragma unroll
  for (int u = 0; u < NX; ++u) {
    const int t = threadIdx.x + u * DL_THREADS;
    if (t < n) xs[t] = v[u];
  }
#pragma unroll
  for (int u = 0; u < NR; ++u) {
    const int r = threadIdx.x + u * DL_THREADS;
    if (r < S) slot[r] = (a[u] != ignore && b[u] != ignore && a[u] >= 0 && a[u] < I) ? (short)a[u] : (short)-1;
  }
}

__device__ __forceinline__ double dl_block_sum(double v, double* sh) {   // sh: [DL_THREADS / 64]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  double s = 0.0;
  for (int w = 0; w < DL_THREADS / 64; ++w) s += sh[w];
  return s;
}

__global__ __launch_bounds__(DL_THREADS) void disc_loss_fwd_kernel(const float* __restrict__ x,
                                                                   const int64_t* __restrict__ ins,
                                                                   const int64_t* __restrict__ sem, int S, int I,
                                                                   int64_t ignore, DlParams P, float* __restrict__ out,
                                                                   float* __restrict__ saved) {
  __shared__ float xs[DL_ROWS * DL_D];
  __shared__ short slot[DL_ROWS];
  __shared__ float mu[DL_SLOTS][8];
  __shared__ float psum[DL_THREADS];
  __shared__ double red[DL_THREADS / 64];
  dl_stage(x, ins, sem, S, I, ignore, xs, slot);
  __syncthreads();
  const int nch = dl_chunks(I), per = I * 8;
  if (threadIdx.x < nch * per) {
    const int ch = threadIdx.x / per, u = threadIdx.x - ch * per;
    const int a = u >> 3, d = u & 7;
    const int r0 = (int)((int64_t)S * ch / nch), r1 = (int)((int64_t)S * (ch + 1) / nch);
    float s = 0.f;
    // eight rows per trip, their slot / value reads issued together (a read, a compare and a dependent read per row was
    // ~150 cycles per row: 250 rows per thread = most of the kernel's 44 us); same additions in the same order
    for (int r = r0; r < r1; r += 8) {
      short sl[8];
      float v[8];
#pragma unroll
      for (int u8 = 0; u8 < 8; ++u8) {
        const int rr = r + u8 < r1 ? r + u8 : r1 - 1;
        sl[u8] = slot[rr];
        v[u8] = d < DL_D ? xs[rr * DL_D + d] : 1.0f;
      }
#pragma unroll
      for (int u8 = 0; u8 < 8; ++u8)
        if (r + u8 < r1 && sl[u8] == a) s += v[u8];
    }
    psum[threadIdx.x] = s;
  }
  __syncthreads();
  if (threadIdx.x < per) {
    float s = 0.f;
    for (int ch = 0; ch < nch; ++ch) s += psum[ch * per + threadIdx.x];
    mu[threadIdx.x >> 3][threadIdx.x & 7] = s;
  }
  __syncthreads();
  {
    const bool act = threadIdx.x < I * 8;
    const int a = threadIdx.x >> 3, d = threadIdx.x & 7;
    const float c = act ? mu[a][7] : 1.0f, sv = act ? mu[a][d] : 0.0f;
    __syncthreads();
    if (act && d < DL_D) mu[a][d] = sv / fmaxf(c, 1.0f);
  }
  __syncthreads();
  double lvar = 0.0, ldist = 0.0, lreg = 0.0, n = 0.0;
  float* kk = saved + DL_SLOTS * 8;
  for (int r = threadIdx.x; r < S; r += DL_THREADS) {
    const int a = slot[r];
    float k = 0.f;
    if (a >= 0) {
      float t2 = 0.f;
#pragma unroll
      for (int d = 0; d < DL_D; ++d) {
        const float e = xs[r * DL_D + d] - mu[a][d];
        t2 += e * e;
      }
      const float t = sqrtf(t2), h = fmaxf(t - P.delta_v, 0.f);
      lvar += (double)(h * h / mu[a][7]);
      k = t > 0.f ? 2.0f * h / t : 0.f;
    }
    kk[r] = k;
  }
  for (int q = threadIdx.x; q < I * I; q += DL_THREADS) {
    const int a = q / I, b = q - a * I;
    if (a != b && mu[a][7] > 0.f && mu[b][7] > 0.f) {
      float l1 = 0.f;
#pragma unroll
      for (int d = 0; d < DL_D; ++d) l1 += fabsf(mu[a][d] - mu[b][d]);
      const float h = fmaxf(2.0f * P.delta_d - l1, 0.f);
      ldist += (double)(h * h);
    }
  }
  if (threadIdx.x < I && mu[threadIdx.x][7] > 0.f) {
    float t2 = 0.f;
#pragma unroll
    for (int d = 0; d < DL_D; ++d) t2 += mu[threadIdx.x][d] * mu[threadIdx.x][d];
    lreg = (double)sqrtf(t2);
    n = 1.0;
  }
  lvar = dl_block_sum(lvar, red);
  ldist = dl_block_sum(ldist, red);
  lreg = dl_block_sum(lreg, red);
  n = dl_block_sum(n, red);
  for (int t = threadIdx.x; t < DL_SLOTS * 8; t += DL_THREADS) saved[t] = (t >> 3) < I ? mu[t >> 3][t & 7] : 0.f;
  if (threadIdx.x == 0) {
    const double den = n * (n - 1.0) > 1.0 ? n * (n - 1.0) : 1.0;
    out[0] = (float)(P.p_var * (lvar / n) + P.p_dist * (ldist / den) + P.p_reg * lreg);   // n == 0 -> nan (0/0), as torch
    saved[DL_SLOTS * 8 + DL_ROWS] = (float)n;
  }
}

__global__ __launch_bounds__(DL_THREADS) void disc_loss_bwd_kernel(const float* __restrict__ x,
                                                                   const int64_t* __restrict__ ins,
                                                                   const int64_t* __restrict__ sem, int S, int I,
                                                                   int64_t ignore, DlParams P,
                                                                   const float* __restrict__ saved,
                                                                   const float* __restrict__ gout,
                                                                   float* __restrict__ dx) {
  __shared__ float xs[DL_ROWS * DL_D];
  __shared__ short slot[DL_ROWS];
  __shared__ float mu[DL_SLOTS][8];
  __shared__ float gmu[DL_SLOTS][8];
  __shared__ float psum[DL_THREADS];
  __shared__ float kk[DL_ROWS];
  dl_stage(x, ins, sem, S, I, ignore, xs, slot);
  for (int t = threadIdx.x; t < DL_SLOTS * 8; t += DL_THREADS) mu[t >> 3][t & 7] = saved[t];
  {
    float kv[DL_ROWS / DL_THREADS];      // (loads first, then the LDS stores: see dl_stage)
#pragma unroll
    for (int u = 0; u < DL_ROWS / DL_THREADS; ++u) {
      const int r = threadIdx.x + u * DL_THREADS;
      kv[u] = saved[DL_SLOTS * 8 + (r < S ? r : 0)];
    }
#pragma unroll
    for (int u = 0; u < DL_ROWS / DL_THREADS; ++u) {
      const int r = threadIdx.x + u * DL_THREADS;
      if (r < S) kk[r] = kv[u];
    }
  }
  __syncthreads();
  const float n = saved[DL_SLOTS * 8 + DL_ROWS];
  const float den = n * (n - 1.0f) > 1.0f ? n * (n - 1.0f) : 1.0f;
  // pull term through the mean: sum_{i in a} k_i (x_i - mu_a), chunked over the rows like the forward sums
  const int nch = dl_chunks(I), per = I * 8;
  if (threadIdx.x < nch * per) {
    const int ch = threadIdx.x / per, u = threadIdx.x - ch * per;
    const int a = u >> 3, d = u & 7;
    const int r0 = (int)((int64_t)S * ch / nch), r1 = (int)((int64_t)S * (ch + 1) / nch);
    float via = 0.f;
    if (d < DL_D && mu[a][7] > 0.f) {
      const float m = mu[a][d];
      // eight rows per trip, their LDS reads issued together (as the forward sums: a read, a compare and two dependent
      // reads per row left the loop latency-bound); same additions in the same order
      for (int r = r0; r < r1; r += 8) {
        short sl[8];
        float kv[8], xv[8];
#pragma unroll
        for (int u8 = 0; u8 < 8; ++u8) {
          const int rr = r + u8 < r1 ? r + u8 : r1 - 1;
          sl[u8] = slot[rr];
          kv[u8] = kk[rr];
          xv[u8] = xs[rr * DL_D + d];
        }
#pragma unroll
        for (int u8 = 0; u8 < 8; ++u8)
          if (r + u8 < r1 && sl[u8] == a) via += kv[u8] * (xv[u8] - m);
      }
    }
    psum[threadIdx.x] = via;
  }
  __syncthreads();
  if (threadIdx.x < I * 8) {
    const int a = threadIdx.x >> 3, d = threadIdx.x & 7;
    float gsum = 0.f;
    const float c = mu[a][7];
    if (d < DL_D && c > 0.f) {
      // push term: every unordered pair appears twice in the ordered sum
      for (int b = 0; b < I; ++b) {
        if (b == a || !(mu[b][7] > 0.f)) continue;
        float l1 = 0.f;
#pragma unroll
        for (int e = 0; e < DL_D; ++e) l1 += fabsf(mu[a][e] - mu[b][e]);
        const float h = fmaxf(2.0f * P.delta_d - l1, 0.f);
        const float df = mu[a][d] - mu[b][d];
        const float sg = df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f);
        gsum += (P.p_dist / den) * 4.0f * h * (-sg);
      }
      float t2 = 0.f;
#pragma unroll
      for (int e = 0; e < DL_D; ++e) t2 += mu[a][e] * mu[a][e];
      const float nm = sqrtf(t2);
      if (nm > 0.f) gsum += P.p_reg * mu[a][d] / nm;
      float via = 0.f;
      for (int ch = 0; ch < nch; ++ch) via += psum[ch * per + threadIdx.x];
      gsum += -(P.p_var / (n * c)) * via;
    }
    gmu[a][d] = gsum;
  }
  __syncthreads();
  const float g = gout[0];
  for (int t = threadIdx.x; t < S * DL_D; t += DL_THREADS) {
    const int r = t / DL_D, d = t - r * DL_D;
    const int a = slot[r];
    float v = 0.f;
    if (a >= 0) {
      const float c = mu[a][7];
      v = g * ((P.p_var / (n * c)) * kk[r] * (xs[t] - mu[a][d]) + gmu[a][d] / c);
    }
    dx[t] = v;
  }
}

int sl_blocks(int64_t N) {
  const char* e = tune_env("WSIS_SL_BLOCKS");          // (tuning knob, read per call; <= SL_MAX_BLOCKS)
  int cap = e ? atoi(e) : 256;      // (one workgroup per CU: 22 + 19 us forward + finish at 512 workgroups, 27 + 10.5 at 256, 33.5 + 7 at 128)
  if (cap < 1 || cap > SL_MAX_BLOCKS) cap = SL_MAX_BLOCKS;
  int64_t b = ceil_div(N > 0 ? N : 1, SL_THREADS);
  if (b > cap) b = cap;
  return (int)b;
}

}  // namespace
}  // namespace wsis

using namespace wsis;

extern "C" {

int64_t wsis_semantic_loss_workspace_bytes(int64_t N) {
  if (N < 0) return -1;
  return (int64_t)sl_blocks(N) * SL_VALS * (int64_t)sizeof(float) + 256;
}

int wsis_semantic_loss_fwd(const float* d_scores, const int64_t* d_labels, int64_t N, int32_t C, int64_t ignore_label,
                           float* d_out2, float* d_saved, void* d_ws, int64_t ws_bytes, void* stream) {
  WSIS_REQUIRE(N >= 0 && C >= 1 && C <= SL_CMAX, "1 <= classes <= 32");
  WSIS_REQUIRE(d_out2 && d_saved && d_ws, "null pointer");
  WSIS_REQUIRE(N == 0 || (d_scores && d_labels), "null input");
  WSIS_REQUIRE(ws_bytes >= wsis_semantic_loss_workspace_bytes(N), "workspace too small");
  hipStream_t st = as_stream(stream);
  const int nblk = sl_blocks(N);
  float* partial = static_cast<float*>(d_ws);
  const size_t ldsb = (size_t)SL_VALS * SL_TP * sizeof(float);      // 102 KB
  static bool attr_set = false;       // (one process per GPU: include/wsis_hip.h)
  if (!attr_set) {
    WSIS_HIP_CHECK(hipFuncSetAttribute((const void*)sem_loss_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
    attr_set = true;
  }
  hipLaunchKernelGGL(sem_loss_fwd_kernel, dim3(nblk), dim3(SL_THREADS), ldsb, st, d_scores, d_labels, N, (int)C,
                     ignore_label, partial);
  WSIS_LAUNCH_CHECK();
  hipLaunchKernelGGL(sem_loss_final_kernel, dim3(1), dim3(1024), 0, st, partial, nblk, (int)C, d_out2, d_saved);
  WSIS_LAUNCH_CHECK();
  return WSIS_OK;
}

int wsis_semantic_loss_bwd(const float* d_scores, const int64_t* d_labels, int64_t N, int32_t C, int64_t ignore_label,
                           const float* d_saved, const float* d_grad_loss, float* d_dscores, void* stream) {
  WSIS_REQUIRE(N >= 0 && C >= 1 && C <= SL_CMAX, "1 <= classes <= 32");
  if (N == 0) return WSIS_OK;
  WSIS_REQUIRE(d_scores && d_labels && d_saved && d_grad_loss && d_dscores, "null pointer");
  hipLaunchKernelGGL(sem_loss_bwd_kernel, dim3(grid_for(N, SL_THREADS)), dim3(SL_THREADS), 0, as_stream(stream),
                     d_scores, d_labels, N, (int)C, ignore_label, d_saved, d_grad_loss, d_dscores);
  WSIS_LAUNCH_CHECK();
  return WSIS_OK;
}

int wsis_sp_regression_loss_fwd(const float* d_pred_off, const float* d_gt_off, const float* d_pred_occ,
                                const float* d_gt_occ, const float* d_pred_size, const float* d_gt_size,
                                const int64_t* d_sem_label, const int64_t* d_ins_label, int64_t S,
                                int64_t ignore_label, float* d_out5, void* stream) {
  WSIS_REQUIRE(S >= 0 && d_out5, "bad args");
  WSIS_REQUIRE(S == 0 || (d_pred_off && d_gt_off && d_pred_occ && d_gt_occ && d_pred_size && d_gt_size &&
                          d_sem_label && d_ins_label), "null pointer");
  hipLaunchKernelGGL(sp_reg_fwd_kernel, dim3(1), dim3(SR_THREADS), 0, as_stream(stream), d_pred_off, d_gt_off,
                     d_pred_occ, d_gt_occ, d_pred_size, d_gt_size, d_sem_label, d_ins_label, S, ignore_label, d_out5);
  WSIS_LAUNCH_CHECK();
  return WSIS_OK;
}

int wsis_sp_regression_loss_bwd(const float* d_pred_off, const float* d_gt_off, const float* d_pred_occ,
                                const float* d_gt_occ, const float* d_pred_size, const float* d_gt_size,
                                const int64_t* d_sem_label, const int64_t* d_ins_label, int64_t S,
                                int64_t ignore_label, const float* d_out5, const float* d_g_norm,
                                const float* d_g_dir, const float* d_g_occ, const float* d_g_size, float* d_doff,
                                float* d_docc, float* d_dsize, void* stream) {
  WSIS_REQUIRE(S >= 0, "bad args");
  if (S == 0) return WSIS_OK;
  WSIS_REQUIRE(d_pred_off && d_gt_off && d_pred_occ && d_gt_occ && d_pred_size && d_gt_size && d_sem_label &&
               d_ins_label && d_out5 && d_g_norm && d_g_dir && d_g_occ && d_g_size && d_doff && d_docc && d_dsize,
               "null pointer");
  hipLaunchKernelGGL(sp_reg_bwd_kernel, dim3(grid_for(S, 256)), dim3(256), 0, as_stream(stream), d_pred_off, d_gt_off,
                     d_pred_occ, d_gt_occ, d_pred_size, d_gt_size, d_sem_label, d_ins_label, S, ignore_label, d_out5,
                     d_g_norm, d_g_dir, d_g_occ, d_g_size, d_doff, d_docc, d_dsize);
  WSIS_LAUNCH_CHECK();
  return WSIS_OK;
}

int64_t wsis_sp_ce_loss_workspace_bytes(int64_t S) {
  if (S < 0) return -1;
  return ceil_div(S > 0 ? S : 1, CE_THREADS) * 3 * (int64_t)sizeof(double) + 256;
}

int wsis_sp_ce_loss_fwd(const float* d_scores, const int64_t* d_labels, int64_t S, int32_t C, int64_t ignore_label,
                        float* d_out3, void* d_ws, int64_t ws_bytes, void* d_sync, void* stream) {
  WSIS_REQUIRE(S >= 1 && C >= 1 && C <= 32 && d_out3, "bad args (S >= 1, C <= 32)");
  WSIS_REQUIRE(d_scores && d_labels && d_ws && d_sync, "null pointer");
  WSIS_REQUIRE(ws_bytes >= wsis_sp_ce_loss_workspace_bytes(S), "workspace too small");
  double* partial = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(d_ws) + 255) & ~(uintptr_t)255);
  hipLaunchKernelGGL(sp_ce_fwd_kernel, dim3((unsigned)ceil_div(S, CE_THREADS)), dim3(CE_THREADS), 0, as_stream(stream), d_scores,
                     d_labels, S, (int)C, ignore_label, partial, static_cast<SyncSlot*>(d_sync)->ticket, d_out3);
  WSIS_LAUNCH_CHECK();
  return WSIS_OK;
}

int wsis_sp_ce_loss_bwd(const float* d_scores, const int64_t* d_labels, int64_t S, int32_t C, int64_t ignore_label,
                        const float* d_out3, const float* d_grad_loss, float* d_dscores, void* stream) {
  WSIS_REQUIRE(S >= 0 && C >= 1 && C <= 32, "bad args (C <= 32)");
  if (S == 0) return WSIS_OK;
  WSIS_REQUIRE(d_scores && d_labels && d_out3 && d_grad_loss && d_dscores, "null pointer");
  hipLaunchKernelGGL(sp_ce_bwd_kernel, dim3(grid_for(S, 256)), dim3(256), 0, as_stream(stream), d_scores, d_labels, S, (int)C,
                     ignore_label, d_out3, d_grad_loss, d_dscores);
  WSIS_LAUNCH_CHECK();
  return WSIS_OK;
}

int wsis_loss_sum(const float* const* d_terms, int32_t n, uint32_t paired, float* d_out, void* stream) {
  WSIS_REQUIRE(d_terms && d_out && n >= 1 && n <= 8, "1..8 terms");
  const float* t[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  for (int i = 0; i < n; ++i) {
    WSIS_REQUIRE(d_terms[i], "null term");
    t[i] = d_terms[i];
  }
  WSIS_REQUIRE((paired >> (n - 1)) == 0, "a paired term needs a successor");
  hipLaunchKernelGGL(loss_sum_kernel, dim3(1), dim3(1), 0, as_stream(stream), t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7], (int)n,
                     (unsigned)paired, d_out);
  WSIS_LAUNCH_CHECK();
  return WSIS_OK;
}

int32_t wsis_disc_loss_saved_floats(void) { return DL_SLOTS * 8 + DL_ROWS + 1; }

int wsis_disc_loss_fwd(const float* d_x, const int64_t* d_ins_label, const int64_t* d_sem_label, int64_t S,
                       int32_t D, int32_t n_slots, int64_t ignore_label, float delta_v, float delta_d, float p_var,
                       float p_dist, float p_reg, float* d_out1, float* d_saved, void* stream) {
  WSIS_REQUIRE(S >= 1 && S <= DL_ROWS && D == DL_D && n_slots >= 1 && n_slots <= DL_SLOTS,
               "1 <= rows <= 4096, 7 features, 1 <= slots <= 64");
  WSIS_REQUIRE(d_x && d_ins_label && d_sem_label && d_out1 && d_saved, "null pointer");
  const DlParams P = {delta_v, delta_d, p_var, p_dist, p_reg};
  hipLaunchKernelGGL(disc_loss_fwd_kernel, dim3(1), dim3(DL_THREADS), 0, as_stream(stream), d_x, d_ins_label,
                     d_sem_label, (int)S, (int)n_slots, ignore_label, P, d_out1, d_saved);
  WSIS_LAUNCH_CHECK();
  return WSIS_OK;
}

int wsis_disc_loss_bwd(const float* d_x, const int64_t* d_ins_label, const int64_t* d_sem_label, int64_t S,
                       int32_t D, int32_t n_slots, int64_t ignore_label, float delta_v, float delta_d, float p_var,
                       float p_dist, float p_reg, const float* d_saved, const float* d_grad_loss, float* d_dx,
                       void* stream) {
  WSIS_REQUIRE(S >= 1 && S <= DL_ROWS && D == DL_D && n_slots >= 1 && n_slots <= DL_SLOTS,
               "1 <= rows <= 4096, 7 features, 1 <= slots <= 64");
  WSIS_REQUIRE(d_x && d_ins_label && d_sem_label && d_saved && d_grad_loss && d_dx, "null pointer");
  const DlParams P = {delta_v, delta_d, p_var, p_dist, p_reg};
  hipLaunchKernelGGL(disc_loss_bwd_kernel, dim3(1), dim3(DL_THREADS), 0, as_stream(stream), d_x, d_ins_label,
                     d_sem_label, (int)S, (int)n_slots, ignore_label, P, d_saved, d_grad_loss, d_dx);
  WSIS_LAUNCH_CHECK();
  return WSIS_OK;
}

}  // extern "C"
