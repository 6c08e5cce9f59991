// BatchNorm1d (+ReLU) over the active voxels: modules/model/sparse_unet3d.py:128-137,
// modules/model/backbone_3D_WSIS.py:47,52-55.  The reference runs BN and ReLU as separate torch ops
// (several elementwise passes + atomics-free but multi-kernel reductions); here
//   forward  = stats (one read of x, fixed-order tree)  + apply(+ReLU) (one read, one write)
//   backward = reduce (dgamma, dbeta; reads x, dy)      + apply (reads x, dy; writes dx)
// All reductions use per-workgroup partials combined in a fixed order (Chan's formula for mean/M2), so the
// result is run-to-run deterministic.  HBM-bound: 2*M*C*4 bytes per pass.
#include <cstdlib>

#include "common.h"

using namespace wsis;

namespace {

constexpr int BN_ROWS_PER_THREAD = 8;   // rows one thread accumulates: all 8 (x, dy) loads are in flight at once
// rows reduced by one workgroup: 256 threads = G channel groups x R row lanes, 8 rows per lane.  (A fixed 256 rows
// made a thread of the wide levels walk 26-32 rows four at a time: 7-8 memory latencies per launch.)
__host__ __device__ inline int bn_rows_per_wg(int Cp) {
  const int G = Cp >> 2;
  const int R = 256 / G > 1 ? 256 / G : 1;
  return R * BN_ROWS_PER_THREAD;
}
constexpr int BN_THREADS = 256;

struct f4 {
  float v[4];
};
__device__ __forceinline__ f4 ld4(const float* p, bool vec) {
  f4 r;
  if (vec) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w;
  } else {
    r.v[0] = p[0]; r.v[1] = p[1]; r.v[2] = p[2]; r.v[3] = p[3];
  }
  return r;
}

// Thread layout of the reductions: channels are handled in groups of 4 (one 16-byte load per row);
// 256 threads = G channel groups x R row lanes.  C is padded to C4*4 logically; tail channels are masked.
// pivot of channel c: the mean of its first (up to) 8 rows, fp32, fixed order.  Every sum of the statistics pass is
// taken of x - K: fp32 sums of the raw x and x^2 cancel in var = E[x^2] - mean^2 when |mean| >> sigma (|mean| = 1000
// sigma: 10 % off); a pivot within sigma / sqrt(8) of the mean keeps sum (x - K)^2 within 12 % of the centred sum (a
// one-row pivot doubles it, and with it the rounding error: median gradient error against the fp64 oracle 3.7e-4 ->
// 6.6e-4)
__device__ __forceinline__ float bn_pivot(const float* __restrict__ x, int64_t M, int C, int c) {
  const int n = M < 8 ? (int)M : 8;
  float s = 0.0f;
  for (int r = 0; r < n; ++r) s += x[(int64_t)r * C + c];
  return s / (float)n;
}

// pass 1 (forward): per-workgroup (sum, sumsq) per channel from ONE read of x.  partial [nblk][2][Cp].
// Both sums are taken of x - K with the pivot K[c] = x[0][c] (one sample of the channel): fp32 sums of the raw x and x^2
// cancel in var = E[x^2] - mean^2 when |mean| >> sigma (|mean| = 1000 sigma: 10 % off), the shifted ones do not.
__global__ __launch_bounds__(BN_THREADS) void bn_stats_partial_kernel(const float* __restrict__ x, int64_t M, int C,
                                                                      int Cp, float* __restrict__ partial) {
  __shared__ float s_a[BN_THREADS * 4];
  __shared__ float s_b[BN_THREADS * 4];
  const int G = Cp >> 2;                       // channel groups
  const int R = max(BN_THREADS / G, 1);        // row lanes
  const bool vec = (C & 3) == 0;
  const int64_t r0 = (int64_t)blockIdx.x * bn_rows_per_wg(Cp);
  const int64_t r1 = min(M, r0 + bn_rows_per_wg(Cp));
  for (int g0 = 0; g0 < G; g0 += BN_THREADS) {  // G <= 256 in practice: one trip
    const int g = g0 + (threadIdx.x % min(G, BN_THREADS));
    const int rl = threadIdx.x / min(G, BN_THREADS);
    float sa[4] = {0.f, 0.f, 0.f, 0.f}, sb[4] = {0.f, 0.f, 0.f, 0.f};
    if (g < G && rl < R) {
      const int c = g * 4;
      float K[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) K[e] = (c + e < C) ? bn_pivot(x, M, C, c + e) : 0.0f;
#pragma unroll 8
      for (int64_t r = r0 + rl; r < r1; r += R) {
        float v[4];
        if (vec) {
          const f4 t = ld4(x + r * C + c, true);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = t.v[e] - K[e];
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (c + e < C) ? x[r * C + c + e] - K[e] : 0.0f;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          sa[e] += v[e];
          sb[e] += v[e] * v[e];
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      s_a[threadIdx.x * 4 + e] = sa[e];
      s_b[threadIdx.x * 4 + e] = sb[e];
    }
    __syncthreads();
    if (g < G && rl == 0) {
      const int gw = min(G, BN_THREADS);
      float ta[4] = {0.f, 0.f, 0.f, 0.f}, tb[4] = {0.f, 0.f, 0.f, 0.f};
      for (int j = 0; j < R; ++j)   // fixed order
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          ta[e] += s_a[(j * gw + (threadIdx.x % gw)) * 4 + e];
          tb[e] += s_b[(j * gw + (threadIdx.x % gw)) * 4 + e];
        }
      float* p = partial + (int64_t)blockIdx.x * 2 * Cp;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        p[g * 4 + e] = ta[e];
        p[Cp + g * 4 + e] = tb[e];
      }
    }
    __syncthreads();
  }
}

// pass 2: one wavefront per channel: lanes stride over the partials (fp64), fixed butterfly => deterministic.
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__global__ __launch_bounds__(64) void bn_stats_final_kernel(const float* __restrict__ partial, int nblk, int C,
                                                            int Cp, int64_t M, const float* __restrict__ x0,
                                                            float* __restrict__ mean,
                                                            float* __restrict__ var,
                                                            float* __restrict__ running_mean,
                                                            float* __restrict__ running_var, float momentum) {
  const int c = blockIdx.x;
  double s = 0.0, q = 0.0;
#pragma unroll 8
  for (int b = threadIdx.x; b < nblk; b += 64) {
    s += partial[(int64_t)b * 2 * Cp + c];
    q += partial[(int64_t)b * 2 * Cp + Cp + c];
  }
  const double S = wave_sum_f64(s), Q = wave_sum_f64(q);
  if (threadIdx.x == 0) {
    const double n = (double)M;
    const double ms = S / n;              // mean of x - K, K = x0[c] (bn_stats_partial_kernel)
    const double mu = (double)bn_pivot(x0, M, C, c) + ms;
    double v = Q / n - ms * ms;
    if (v < 0.0) v = 0.0;
    mean[c] = (float)mu;
    var[c] = (float)v;
    if (running_mean) {
      const double unb = n > 1 ? v * n / (n - 1) : v;
      running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mu);
      running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unb);
    }
  }
}

// the same from CENTRED partials of 32-row slices (written by the convolution epilogues, csrc/spconv2.hip): slice i
// holds S_i = sum and Q_i = sum of squared deviations from ITS mean over n_i = min(32, M - 32 i) rows;
// mean = sum S_i / M, var = (sum Q_i + sum S_i^2 / n_i - M mean^2) / M, all in fp64 (Chan's pairwise combination).
// Two levels so that the 1.2 MB of partials of a 150k-row level is read by up to 64 workgroups per 32 channels instead of
// one: chunk sums (fixed order inside a chunk) -> [G][3][C] doubles, then one thread per channel adds the chunks in order.
// arrival tickets of the two-level reductions: one counter per channel group in the caller's sync slot (common.h
// SyncSlot; zero between launches: the last workgroup resets its counter).
constexpr int BN_FIN_CHUNKS = kBnFinChunks;      // <= 64: the finish kernels hold one chunk per lane

// chunk g of G for channel group cgi: 256 threads = 32 channel lanes x 8 partial lanes; G == 1 finishes in place.
// Returns true in the workgroup that wrote the final mean / var of the channel group (G == 1, or the last arrival).
__device__ __forceinline__ bool bn_chunk_centred_stage(const float* __restrict__ partial, int nblk, int C, int64_t M,
                                                       double* __restrict__ chunk, float* __restrict__ mean,
                                                       float* __restrict__ var, float* __restrict__ running_mean,
                                                       float* __restrict__ running_var, float momentum,
                                                       unsigned* __restrict__ ticket, int g0, int G, int cgi,
                                                       double (&red)[3][8][33], int& s_last) {
  const int cl = threadIdx.x & 31, pl = threadIdx.x >> 5;
  const int c = cgi * 32 + cl;
  const int per = (nblk + G - 1) / G;
  const int lo = g0 * per;
  const int hi = lo + per < nblk ? lo + per : nblk;
  double s = 0.0, q = 0.0, w = 0.0;
  if (c < C) {
#pragma unroll 4
    for (int b = lo + pl; b < hi; b += 8) {
      const float sf = partial[(int64_t)b * 2 * C + c];
      const float qf = partial[(int64_t)b * 2 * C + C + c];
      const int64_t left = M - (int64_t)b * 32;
      const double si = sf;
      s += si;
      q += qf;
      w += si * si * (left < 32 ? 1.0 / (double)left : 0.03125);
    }
  }
  red[0][pl][cl] = s;
  red[1][pl][cl] = q;
  red[2][pl][cl] = w;
  __syncthreads();
  if (pl == 0 && c < C) {
    double S = 0.0, Q = 0.0, W = 0.0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {      // fixed order
      S += red[0][j][cl];
      Q += red[1][j][cl];
      W += red[2][j][cl];
    }
    if (G == 1) {
      bn_finish_centred(S, Q, W, M, c, mean, var, running_mean, running_var, momentum);
    } else {
      double* o = chunk + (int64_t)g0 * 3 * C;
      st_sc1(o + c, S);
      st_sc1(o + C + c, Q);
      st_sc1(o + 2 * C + c, W);
    }
  }
  if (G == 1) return true;
  if (!ticket) return false;
  // ---- the workgroup that arrives last (per channel group) adds the chunks, always in chunk order: one launch instead
  // of two.  The chunk rows cross XCDs as sc1 stores / sc1 loads around the ticket (common.h): no fences, each of which
  // costs more than the rest of this stage.
  wait_stores_left();
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned t = atomicAdd(ticket + cgi, 1u);
    s_last = t == (unsigned)G - 1;
    if (s_last) ticket[cgi] = 0u;          // self-cleaning: ready for the next launch on this stream
  }
  __syncthreads();
  if (!s_last) return false;
  double s2 = 0.0, q2 = 0.0, w2 = 0.0;
  if (c < C)
    for (int g = pl; g < G; g += 8) {
      const double* o = chunk + (int64_t)g * 3 * C;
      s2 += ld_sc1(o + c);
      q2 += ld_sc1(o + C + c);
      w2 += ld_sc1(o + 2 * C + c);
    }
  __syncthreads();
  red[0][pl][cl] = s2;
  red[1][pl][cl] = q2;
  red[2][pl][cl] = w2;
  __syncthreads();
  if (pl == 0 && c < C) {
    double S = 0.0, Q = 0.0, W = 0.0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      S += red[0][j][cl];
      Q += red[1][j][cl];
      W += red[2][j][cl];
    }
    bn_finish_centred(S, Q, W, M, c, mean, var, running_mean, running_var, momentum);
  }
  return true;
}

// grid (G, ceil(C / 32))
__global__ __launch_bounds__(256) void bn_stats_chunk_centred_kernel(const float* __restrict__ partial, int nblk, int C,
                                                                     int64_t M, double* __restrict__ chunk,
                                                                     float* __restrict__ mean, float* __restrict__ var,
                                                                     float* __restrict__ running_mean,
                                                                     float* __restrict__ running_var, float momentum,
                                                                     unsigned* __restrict__ ticket) {
  __shared__ double red[3][8][33];
  __shared__ int s_last;
  bn_chunk_centred_stage(partial, nblk, C, M, chunk, mean, var, running_mean, running_var, momentum, ticket,
                         (int)blockIdx.x, (int)gridDim.x, (int)blockIdx.y, red, s_last);
}

// one wavefront per channel: lane g holds chunk g (G <= 64), fixed butterfly
__global__ __launch_bounds__(256) void bn_stats_final_centred_kernel(const double* __restrict__ chunk, int G, int C,
                                                                     int64_t M, float* __restrict__ mean,
                                                                     float* __restrict__ var,
                                                                     float* __restrict__ running_mean,
                                                                     float* __restrict__ running_var, float momentum) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= C) return;
  const bool in = lane < G;
  const double S = wave_sum_f64(in ? chunk[(int64_t)lane * 3 * C + c] : 0.0);
  const double Q = wave_sum_f64(in ? chunk[(int64_t)lane * 3 * C + C + c] : 0.0);
  const double W = wave_sum_f64(in ? chunk[(int64_t)lane * 3 * C + 2 * C + c] : 0.0);
  if (lane == 0) bn_finish_centred(S, Q, W, M, c, mean, var, running_mean, running_var, momentum);
}

// COHERENT: mean / var were written by other workgroups of THIS launch (bn_finalize_apply_kernel): they are read with
// agent-scope loads, which do not hit stale lines of this XCD's L2 -- cheaper than an acquire fence, whose L2 invalidate
// is serialised over the waiting workgroups of an XCD (measured: + 50 ns per waiting workgroup)
template <bool COHERENT>
__device__ __forceinline__ float bn_ld_stat(const float* p) {
  if (COHERENT) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return *p;
}

template <bool COHERENT>
__device__ __forceinline__ void bn_apply_body(const float* __restrict__ x, const float* mean,
                                              const float* var, const float* __restrict__ gamma,
                                              const float* __restrict__ beta, float eps, int relu,
                                              float* __restrict__ y, int64_t M, int C) {
  const int64_t total = M * C;
  if ((C & 3) == 0) {
    const int64_t total4 = total >> 2;
    const unsigned C4 = (unsigned)(C >> 2);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const bool fixed_c = (stride % C4) == 0;      // then the channel group of a thread never changes
    unsigned cg = (unsigned)(t % C4);
    float sc[4], mu[4], bt[4];
    auto coef = [&](unsigned g) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = (int)g * 4 + e;
        sc[e] = (gamma ? gamma[c] : 1.0f) * rsqrtf(bn_ld_stat<COHERENT>(var + c) + eps);
        mu[e] = bn_ld_stat<COHERENT>(mean + c);
        bt[e] = beta ? beta[c] : 0.0f;
      }
    };
    coef(cg);
    // fma(x - mean, sc, beta): x - mean first (x*sc + (beta - mean*sc) cancels when |mean| >> sigma), one rounding for
    // the scale-and-shift; the convolutions that apply the BatchNorm while they read their input (csrc/spconv2.hip
    // bnfrag, csrc/spconv_dw2.hip) use the same expression, so fused and unfused passes agree bit for bit
    auto body = [&](const float4 v) {
      float o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float z = __builtin_fmaf(o[e] - mu[e], sc[e], bt[e]);
        if (relu) z = fmaxf(z, 0.0f);
        o[e] = z;
      }
      return make_float4(o[0], o[1], o[2], o[3]);
    };
    if (fixed_c) {   // two float4 per trip in flight (the launcher rounds the grid so that this branch is taken)
      for (; t + stride < total4; t += 2 * stride) {
        const float4 v0 = reinterpret_cast<const float4*>(x)[t];
        const float4 v1 = reinterpret_cast<const float4*>(x)[t + stride];
        reinterpret_cast<float4*>(y)[t] = body(v0);
        reinterpret_cast<float4*>(y)[t + stride] = body(v1);
      }
      if (t < total4) reinterpret_cast<float4*>(y)[t] = body(reinterpret_cast<const float4*>(x)[t]);
    } else {
      for (; t < total4; t += stride) {
        cg = (unsigned)(t % C4);
        coef(cg);
        reinterpret_cast<float4*>(y)[t] = body(reinterpret_cast<const float4*>(x)[t]);
      }
    }
  } else {
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
      const int c = (int)(t % C);
      const float sc = (gamma ? gamma[c] : 1.0f) * rsqrtf(bn_ld_stat<COHERENT>(var + c) + eps);
      float z = __builtin_fmaf(x[t] - bn_ld_stat<COHERENT>(mean + c), sc, beta ? beta[c] : 0.0f);
      if (relu) z = fmaxf(z, 0.0f);
      y[t] = z;
    }
  }
}

__global__ void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                const float* __restrict__ var, const float* __restrict__ gamma,
                                const float* __restrict__ beta, float eps, int relu, float* __restrict__ y,
                                int64_t M, int C) {
  bn_apply_body<false>(x, mean, var, gamma, beta, eps, relu, y, M, C);
}

// ---- statistics finish + apply in ONE launch.  The first G * ceil(C/32) workgroups run the chunk stage of
// bn_stats_chunk_centred_kernel (tickets included); the workgroup that completes the last channel group publishes the
// launch's epoch in `flag`; every workgroup waits for it and then applies.  The grid is at most 512 workgroups of 256
// threads (two per CU): all of them are resident, the waiting ones cannot keep the working ones off the machine.
// Same arithmetic as the two launches (bn_stats_chunk_centred_kernel, bn_apply_kernel): identical results.
constexpr int BN_POLL_SLEEP = 12;      // x 64 cycles
constexpr unsigned long long BN_SPIN_LIMIT = 200000000ull;      // s_memrealtime ticks (100 MHz): 2 s

// every workgroup of a producer / consumer launch: wait until the producers have published (flag != 0), bounded -- a
// launch whose producers never become resident sets `err` and goes on instead of hanging the device --, then the
// workgroup that leaves the wait last puts flag and counter back to zero for the next launch on this slot
__device__ __forceinline__ void bn_wait_published(SyncSlot* s) {
  if (threadIdx.x == 0) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__hip_atomic_load(&s->flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
      __builtin_amdgcn_s_sleep(BN_POLL_SLEEP);
      if (__builtin_amdgcn_s_memrealtime() - t0 > BN_SPIN_LIMIT) {
        __hip_atomic_store(&s->err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
    const unsigned l = atomicAdd(&s->left, 1u);
    if (l == gridDim.x - 1) {
      __hip_atomic_store(&s->left, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&s->flag, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __syncthreads();
}
__device__ __forceinline__ void bn_publish(SyncSlot* s, int CG) {      // by thread 0 of a workgroup that finished a group
  const unsigned d = atomicAdd(&s->done, 1u);
  if (d == (unsigned)CG - 1) {
    __hip_atomic_store(&s->done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();
    __hip_atomic_store(&s->flag, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}

__global__ __launch_bounds__(256) void bn_finalize_apply_kernel(
    const float* __restrict__ partial, int nblk, int C, int64_t M, double* __restrict__ chunk, int G,
    float* mean, float* var, float* __restrict__ running_mean,
    float* __restrict__ running_var, float momentum, SyncSlot* __restrict__ sync, const float* __restrict__ x,
    const float* __restrict__ gamma, const float* __restrict__ beta, float eps, int relu, float* __restrict__ y) {
  __shared__ double red[3][8][33];
  __shared__ int s_last;
  const int CG = (C + 31) / 32;
  if ((int)blockIdx.x < G * CG) {
    const bool fin = bn_chunk_centred_stage(partial, nblk, C, M, chunk, mean, var, running_mean, running_var, momentum,
                                            sync->ticket, (int)blockIdx.x % G, G, (int)blockIdx.x / G, red, s_last);
    if (fin) {               // this workgroup wrote mean / var of one channel group; the last such group publishes
      __threadfence();
      __syncthreads();
      if (threadIdx.x == 0) bn_publish(sync, CG);
    }
  }
  bn_wait_published(sync);    // relaxed polls a few hundred ns apart (hundreds of pollers on one word)
  bn_apply_body<true>(x, mean, var, gamma, beta, eps, relu, y, M, C);
}

// backward pass 1: per-workgroup partial (sum dz, sum dz*xhat) per channel.  partial [nblk][2][Cp]
__global__ __launch_bounds__(BN_THREADS) void bn_bwd_partial_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ mean,
    const float* __restrict__ var, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
    int relu, int64_t M, int C, int Cp, float* __restrict__ partial) {
  __shared__ float s_a[BN_THREADS * 4];
  __shared__ float s_b[BN_THREADS * 4];
  const int G = Cp >> 2;
  const int gw = min(G, BN_THREADS);
  const int R = max(BN_THREADS / G, 1);
  const bool vec = (C & 3) == 0;
  const int64_t r0 = (int64_t)blockIdx.x * bn_rows_per_wg(Cp);
  const int64_t r1 = min(M, r0 + bn_rows_per_wg(Cp));
  for (int g0 = 0; g0 < G; g0 += BN_THREADS) {
    const int g = g0 + (threadIdx.x % gw);
    const int rl = threadIdx.x / gw;
    float sa[4] = {0.f, 0.f, 0.f, 0.f}, sb[4] = {0.f, 0.f, 0.f, 0.f};
    if (g < G && rl < R) {
      const int c = g * 4;
      float mu[4], rstd[4], gm[4], bt[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int cc = min(c + e, C - 1);
        mu[e] = mean[cc];
        rstd[e] = rsqrtf(var[cc] + eps);
        gm[e] = gamma ? gamma[cc] : 1.0f;
        bt[e] = beta ? beta[cc] : 0.0f;
      }
#pragma unroll 8
      for (int64_t r = r0 + rl; r < r1; r += R) {
        float xv[4], dv[4];
        if (vec) {
          const f4 tx = ld4(x + r * C + c, true), td = ld4(dy + r * C + c, true);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            xv[e] = tx.v[e];
            dv[e] = td.v[e];
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const bool ok = c + e < C;
            xv[e] = ok ? x[r * C + c + e] : 0.0f;
            dv[e] = ok ? dy[r * C + c + e] : 0.0f;
          }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float xh = (xv[e] - mu[e]) * rstd[e];
          float dz = dv[e];
          if (relu && xh * gm[e] + bt[e] <= 0.0f) dz = 0.0f;
          sa[e] += dz;
          sb[e] += dz * xh;
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      s_a[threadIdx.x * 4 + e] = sa[e];
      s_b[threadIdx.x * 4 + e] = sb[e];
    }
    __syncthreads();
    if (g < G && rl == 0) {
      float ta[4] = {0.f, 0.f, 0.f, 0.f}, tb[4] = {0.f, 0.f, 0.f, 0.f};
      for (int j = 0; j < R; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          ta[e] += s_a[(j * gw + (threadIdx.x % gw)) * 4 + e];
          tb[e] += s_b[(j * gw + (threadIdx.x % gw)) * 4 + e];
        }
      float* p = partial + (int64_t)blockIdx.x * 2 * Cp;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        p[g * 4 + e] = ta[e];
        p[Cp + g * 4 + e] = tb[e];
      }
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(64) void bn_bwd_final_kernel(const float* __restrict__ partial, int nblk, int C, int Cp,
                                                          float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int c = blockIdx.x;
  double a = 0.0, b = 0.0;
#pragma unroll 8
  for (int k = threadIdx.x; k < nblk; k += 64) {
    a += partial[(int64_t)k * 2 * Cp + c];
    b += partial[(int64_t)k * 2 * Cp + Cp + c];
  }
  const double A = wave_sum_f64(a), B = wave_sum_f64(b);
  if (threadIdx.x == 0) {
    dbeta[c] = (float)A;
    dgamma[c] = (float)B;
  }
}

// the reduction of the backward pass from per-slice partials (sum dz, sum dz * xhat) written by the epilogue of the
// dIn convolution that produced dy (wsis_spconv_fwd_t_bn): fp64 sums in a fixed order, two levels like the forward
// statistics.  grid (G, ceil(C / 32)), 256 threads = 32 channel lanes x 8 partial lanes; G == 1 writes the result.
__device__ __forceinline__ bool bn_sum_chunk_stage(const float* __restrict__ partial, int nblk, int C,
                                                   double* __restrict__ chunk, float* dbeta, float* dgamma,
                                                   unsigned* __restrict__ ticket, int g0, int G, int cgi,
                                                   double (&red)[2][8][33], int& s_last) {
  const int cl = threadIdx.x & 31, pl = threadIdx.x >> 5;
  const int c = cgi * 32 + cl;
  const int per = (nblk + G - 1) / G;
  const int lo = g0 * per;
  const int hi = lo + per < nblk ? lo + per : nblk;
  double a = 0.0, b = 0.0;
  if (c < C) {
#pragma unroll 4
    for (int k = lo + pl; k < hi; k += 8) {
      a += partial[(int64_t)k * 2 * C + c];
      b += partial[(int64_t)k * 2 * C + C + c];
    }
  }
  red[0][pl][cl] = a;
  red[1][pl][cl] = b;
  __syncthreads();
  if (pl == 0 && c < C) {
    double A = 0.0, B = 0.0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {      // fixed order
      A += red[0][j][cl];
      B += red[1][j][cl];
    }
    if (G == 1) {
      dbeta[c] = (float)A;
      dgamma[c] = (float)B;
    } else {
      st_sc1(chunk + (int64_t)g0 * 2 * C + c, A);
      st_sc1(chunk + (int64_t)g0 * 2 * C + C + c, B);
    }
  }
  if (G == 1) return true;
  if (!ticket) return false;
  // the last workgroup to arrive adds the chunks in chunk order (see bn_chunk_centred_stage)
  wait_stores_left();
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned t = atomicAdd(ticket + cgi, 1u);
    s_last = t == (unsigned)G - 1;
    if (s_last) ticket[cgi] = 0u;
  }
  __syncthreads();
  if (!s_last) return false;
  double a2 = 0.0, b2 = 0.0;
  if (c < C)
    for (int g = pl; g < G; g += 8) {
      const double* o = chunk + (int64_t)g * 2 * C;
      a2 += ld_sc1(o + c);
      b2 += ld_sc1(o + C + c);
    }
  __syncthreads();
  red[0][pl][cl] = a2;
  red[1][pl][cl] = b2;
  __syncthreads();
  if (pl == 0 && c < C) {
    double A = 0.0, B = 0.0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      A += red[0][j][cl];
      B += red[1][j][cl];
    }
    dbeta[c] = (float)A;
    dgamma[c] = (float)B;
  }
  return true;
}

__global__ __launch_bounds__(256) void bn_sum_chunk_kernel(const float* __restrict__ partial, int nblk, int C,
                                                           double* __restrict__ chunk, float* __restrict__ dbeta,
                                                           float* __restrict__ dgamma, unsigned* __restrict__ ticket) {
  __shared__ double red[2][8][33];
  __shared__ int s_last;
  bn_sum_chunk_stage(partial, nblk, C, chunk, dbeta, dgamma, ticket, (int)blockIdx.x, (int)gridDim.x, (int)blockIdx.y, red,
                     s_last);
}

__global__ __launch_bounds__(256) void bn_sum_final_kernel(const double* __restrict__ chunk, int G, int C,
                                                           float* __restrict__ dbeta, float* __restrict__ dgamma) {
  const int lane = threadIdx.x & 63;            // one wavefront per channel: lane g holds chunk g (G <= 64)
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= C) return;
  const bool in = lane < G;
  const double A = wave_sum_f64(in ? chunk[(int64_t)lane * 2 * C + c] : 0.0);
  const double B = wave_sum_f64(in ? chunk[(int64_t)lane * 2 * C + C + c] : 0.0);
  if (lane == 0) {
    dbeta[c] = (float)A;
    dgamma[c] = (float)B;
  }
}

// backward pass 2: dx = gamma*rstd*(dz - dbeta/M - xhat*dgamma/M)  (training);  gamma*rstd*dz (eval)
template <bool COHERENT>     // COHERENT: dgamma / dbeta come from other workgroups of this launch (see bn_apply_body)
__device__ __forceinline__ void bn_bwd_apply_body(const float* __restrict__ x, const float* __restrict__ dy,
                                                  const float* __restrict__ mean, const float* __restrict__ var,
                                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                                  const float* dgamma, const float* dbeta,
                                                  const float* __restrict__ addend, float eps, int relu, int training,
                                                  float* __restrict__ dx, int64_t M, int C) {
  const int64_t total = M * C;
  const float inv_m = 1.0f / (float)M;
  const bool vec = (C & 3) == 0;
  const int W = vec ? 4 : 1;
  const int64_t totalw = total / W;
  const unsigned Cw = (unsigned)(C / W);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  // per-channel coefficients stay in registers while the thread's channel group does not change (it never does
  // when the grid stride is a multiple of the channel groups; the launcher rounds the grid to make it so)
  const bool fixed_c = (stride % Cw) == 0;
  float mu[4], rstd[4], gm[4], bt[4], k1[4], k2[4];
  auto coef = [&](int c0) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (e < W) {
        const int c = c0 + e;
        mu[e] = mean[c];
        rstd[e] = rsqrtf(var[c] + eps);
        gm[e] = gamma ? gamma[c] : 1.0f;
        bt[e] = beta ? beta[c] : 0.0f;
        k1[e] = training ? bn_ld_stat<COHERENT>(dbeta + c) * inv_m : 0.0f;
        k2[e] = training ? bn_ld_stat<COHERENT>(dgamma + c) * inv_m : 0.0f;
      }
    }
  };
  if (t < totalw) coef((int)(t % Cw) * W);
  auto one = [&](const float (&xv)[4], const float (&dv)[4], const float (&av)[4], float (&ov)[4]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (e < W) {
        const float xh = (xv[e] - mu[e]) * rstd[e];
        float dz = dv[e];
        if (relu && xh * gm[e] + bt[e] <= 0.0f) dz = 0.0f;
        float r = dz;
        if (training) r = dz - k1[e] - xh * k2[e];
        ov[e] = gm[e] * rstd[e] * r;
        if (addend) ov[e] += av[e];
      }
    }
  };
  if (vec && fixed_c) {   // two float4 triples per trip in flight
    for (; t + stride < totalw; t += 2 * stride) {
      const int64_t u = t + stride;
      const f4 tx0 = ld4(x + t * 4, true), td0 = ld4(dy + t * 4, true);
      const f4 tx1 = ld4(x + u * 4, true), td1 = ld4(dy + u * 4, true);
      f4 ta0, ta1;
#pragma unroll
      for (int e = 0; e < 4; ++e) ta0.v[e] = ta1.v[e] = 0.f;
      if (addend) {
        ta0 = ld4(addend + t * 4, true);
        ta1 = ld4(addend + u * 4, true);
      }
      float o0[4], o1[4];
      one(tx0.v, td0.v, ta0.v, o0);
      one(tx1.v, td1.v, ta1.v, o1);
      reinterpret_cast<float4*>(dx)[t] = make_float4(o0[0], o0[1], o0[2], o0[3]);
      reinterpret_cast<float4*>(dx)[u] = make_float4(o1[0], o1[1], o1[2], o1[3]);
    }
  }
  for (; t < totalw; t += stride) {
    if (!fixed_c) coef((int)(t % Cw) * W);
    float xv[4], dv[4], av[4] = {0.f, 0.f, 0.f, 0.f}, ov[4];
    if (vec) {
      const f4 tx = ld4(x + t * 4, true), td = ld4(dy + t * 4, true);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        xv[e] = tx.v[e];
        dv[e] = td.v[e];
      }
      if (addend) {
        const f4 ta = ld4(addend + t * 4, true);
#pragma unroll
        for (int e = 0; e < 4; ++e) av[e] = ta.v[e];
      }
    } else {
      xv[0] = x[t];
      dv[0] = dy[t];
      if (addend) av[0] = addend[t];
    }
    one(xv, dv, av, ov);
    if (vec)
      reinterpret_cast<float4*>(dx)[t] = make_float4(ov[0], ov[1], ov[2], ov[3]);
    else
      dx[t] = ov[0];
  }
}

__global__ void bn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                    const float* __restrict__ mean, const float* __restrict__ var,
                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                    const float* __restrict__ dgamma, const float* __restrict__ dbeta,
                                    const float* __restrict__ addend, float eps, int relu, int training,
                                    float* __restrict__ dx, int64_t M, int C) {
  bn_bwd_apply_body<false>(x, dy, mean, var, gamma, beta, dgamma, dbeta, addend, eps, relu, training, dx, M, C);
}

// reduction finish + apply of the backward pass in ONE launch: the form of bn_finalize_apply_kernel (chunk stage and
// tickets in the first G * ceil(C/32) workgroups, epoch flag, every workgroup waits and applies)
__global__ __launch_bounds__(256) void bn_bwd_finish_apply_kernel(
    const float* __restrict__ partial, int nblk, int C, double* __restrict__ chunk, int G, float* dbeta, float* dgamma,
    SyncSlot* __restrict__ sync, const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ mean,
    const float* __restrict__ var, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ addend, float eps, int relu, float* __restrict__ dx, int64_t M) {
  __shared__ double red[2][8][33];
  __shared__ int s_last;
  const int CG = (C + 31) / 32;
  if ((int)blockIdx.x < G * CG) {
    const bool fin = bn_sum_chunk_stage(partial, nblk, C, chunk, dbeta, dgamma, sync->ticket, (int)blockIdx.x % G, G,
                                        (int)blockIdx.x / G, red, s_last);
    if (fin) {
      __threadfence();
      __syncthreads();
      if (threadIdx.x == 0) bn_publish(sync, CG);
    }
  }
  bn_wait_published(sync);
  bn_bwd_apply_body<true>(x, dy, mean, var, gamma, beta, dgamma, dbeta, addend, eps, relu, 1, dx, M, C);
}

// ---- levels of fewer than 4,096 rows (n_part < 128: ONE chunk): statistics finish + apply, and backward reduction finish +
// apply, as ONE launch WITHOUT any hand-off between workgroups.  Grid (row blocks, 32-channel groups); every workgroup
// runs the (single-chunk) finish of ITS channel group itself -- <= 127 partial rows of 32 channels, the arithmetic and
// order of bn_chunk_centred_stage / bn_sum_chunk_stage with G == 1 -- keeps the coefficients in LDS and applies them to
// its rows; row block 0 also writes mean / var / running statistics (dgamma / dbeta).  The finish launch it replaces is
// 4-5 us + a kernel boundary for a few hundred floats of work.  Same values as the two launches, bit for bit.
constexpr int BN_SF_ROWS = 256;      // rows per workgroup (32 rows x 8 float4 lanes per pass)

__global__ __launch_bounds__(256) void bn_small_finish_apply_kernel(
    const float* __restrict__ partial, int nblk, int C, int64_t M, float* __restrict__ mean, float* __restrict__ var,
    float* __restrict__ running_mean, float* __restrict__ running_var, float momentum, const float* __restrict__ x,
    const float* __restrict__ gamma, const float* __restrict__ beta, float eps, int relu, float* __restrict__ y, int G) {
  __shared__ double red[3][8][33];
  __shared__ float s_mu[32], s_sc[32], s_bt[32];
  const int cgi = blockIdx.y;
  const int cl = threadIdx.x & 31, pl = threadIdx.x >> 5;
  const int c = cgi * 32 + cl;
  // G chunks of `per` partial rows (round 6: G <= 8, the arithmetic of the two-level reduction: the chunk sums of
  // bn_chunk_centred_stage -- lane pl walks rows lo + pl, + 8, ..., the eight lane sums added in lane order -- then the
  // chunk sums added in chunk order, which is what the last-arriving workgroup of that stage does with one chunk per lane)
  const int per = (nblk + G - 1) / G;
  double S = 0.0, Q = 0.0, W = 0.0;
  for (int g = 0; g < G; ++g) {
    const int lo = g * per;
    const int hi = lo + per < nblk ? lo + per : nblk;
    double s = 0.0, q = 0.0, w = 0.0;
    if (c < C) {
#pragma unroll 4
      for (int b = lo + pl; b < hi; b += 8) {
        const float sf = partial[(int64_t)b * 2 * C + c];
        const float qf = partial[(int64_t)b * 2 * C + C + c];
        const int64_t left = M - (int64_t)b * 32;
        const double si = sf;
        s += si;
        q += qf;
        w += si * si * (left < 32 ? 1.0 / (double)left : 0.03125);
      }
    }
    if (g) __syncthreads();            // (the previous chunk's lane sums have been read)
    red[0][pl][cl] = s;
    red[1][pl][cl] = q;
    red[2][pl][cl] = w;
    __syncthreads();
    if (pl == 0 && c < C) {
      double Sg = 0.0, Qg = 0.0, Wg = 0.0;
#pragma unroll
      for (int j = 0; j < 8; ++j) {      // fixed order
        Sg += red[0][j][cl];
        Qg += red[1][j][cl];
        Wg += red[2][j][cl];
      }
      if (G == 1) {
        S = Sg; Q = Qg; W = Wg;
      } else {                           // (0.0 + chunk 0 + chunk 1 + ...: the order of the stage's second level)
        S += Sg; Q += Qg; W += Wg;
      }
    }
  }
  if (pl == 0 && c < C) {
    const double n = (double)M;        // (bn_finish_centred)
    const double mu = S / n;
    double v = (Q + (W - n * mu * mu)) / n;
    if (v < 0.0) v = 0.0;
    const float muf = (float)mu, vf = (float)v;
    if (blockIdx.x == 0) {
      mean[c] = muf;
      var[c] = vf;
      if (running_mean) {
        const double unb = n > 1 ? v * n / (n - 1) : v;
        running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mu);
        running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unb);
      }
    }
    s_mu[cl] = muf;
    s_sc[cl] = (gamma ? gamma[c] : 1.0f) * rsqrtf(vf + eps);
    s_bt[cl] = beta ? beta[c] : 0.0f;
  }
  __syncthreads();
  const int q4 = threadIdx.x & 7, rl = threadIdx.x >> 3;
  const int c4 = cgi * 32 + q4 * 4;
  if (c4 >= C) return;
  float mu4[4], sc4[4], bt4[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    mu4[e] = s_mu[q4 * 4 + e];
    sc4[e] = s_sc[q4 * 4 + e];
    bt4[e] = s_bt[q4 * 4 + e];
  }
  const int64_t r0 = (int64_t)blockIdx.x * BN_SF_ROWS;
#pragma unroll 4
  for (int it = 0; it < BN_SF_ROWS / 32; ++it) {
    const int64_t r = r0 + it * 32 + rl;
    if (r < M) {
      const float4 vx = *reinterpret_cast<const float4*>(x + r * C + c4);
      float o[4] = {vx.x, vx.y, vx.z, vx.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float z = __builtin_fmaf(o[e] - mu4[e], sc4[e], bt4[e]);      // (bn_apply_body)
        if (relu) z = fmaxf(z, 0.0f);
        o[e] = z;
      }
      *reinterpret_cast<float4*>(y + r * C + c4) = make_float4(o[0], o[1], o[2], o[3]);
    }
  }
}

__global__ __launch_bounds__(256) void bn_small_bwd_finish_apply_kernel(
    const float* __restrict__ partial, int nblk, int C, int64_t M, const float* __restrict__ x,
    const float* __restrict__ dy, const float* __restrict__ mean, const float* __restrict__ var,
    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ addend, float eps, int relu,
    float* __restrict__ dx, float* __restrict__ dgamma, float* __restrict__ dbeta, int G) {
  __shared__ double red[2][8][33];
  __shared__ float s_k1[32], s_k2[32];
  const int cgi = blockIdx.y;
  const int cl = threadIdx.x & 31, pl = threadIdx.x >> 5;
  const int c = cgi * 32 + cl;
  const int per = (nblk + G - 1) / G;      // (G chunks: the two levels of bn_sum_chunk_stage, see the forward kernel)
  double A = 0.0, B = 0.0;
  for (int g = 0; g < G; ++g) {
    const int lo = g * per;
    const int hi = lo + per < nblk ? lo + per : nblk;
    double a = 0.0, b = 0.0;
    if (c < C) {
#pragma unroll 4
      for (int k = lo + pl; k < hi; k += 8) {
        a += partial[(int64_t)k * 2 * C + c];
        b += partial[(int64_t)k * 2 * C + C + c];
      }
    }
    if (g) __syncthreads();
    red[0][pl][cl] = a;
    red[1][pl][cl] = b;
    __syncthreads();
    if (pl == 0 && c < C) {
      double Ag = 0.0, Bg = 0.0;
#pragma unroll
      for (int j = 0; j < 8; ++j) {      // fixed order
        Ag += red[0][j][cl];
        Bg += red[1][j][cl];
      }
      if (G == 1) {
        A = Ag; B = Bg;
      } else {
        A += Ag; B += Bg;
      }
    }
  }
  if (pl == 0 && c < C) {
    const float db = (float)A, dg = (float)B;
    if (blockIdx.x == 0) {
      dbeta[c] = db;
      dgamma[c] = dg;
    }
    const float inv_m = 1.0f / (float)M;      // (bn_bwd_apply_body)
    s_k1[cl] = db * inv_m;
    s_k2[cl] = dg * inv_m;
  }
  __syncthreads();
  const int q4 = threadIdx.x & 7, rl = threadIdx.x >> 3;
  const int c4 = cgi * 32 + q4 * 4;
  if (c4 >= C) return;
  float mu[4], rstd[4], gm[4], bt[4], k1[4], k2[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int cc = c4 + e;
    mu[e] = mean[cc];
    rstd[e] = rsqrtf(var[cc] + eps);
    gm[e] = gamma ? gamma[cc] : 1.0f;
    bt[e] = beta ? beta[cc] : 0.0f;
    k1[e] = s_k1[q4 * 4 + e];
    k2[e] = s_k2[q4 * 4 + e];
  }
  const int64_t r0 = (int64_t)blockIdx.x * BN_SF_ROWS;
#pragma unroll 2
  for (int it = 0; it < BN_SF_ROWS / 32; ++it) {
    const int64_t r = r0 + it * 32 + rl;
    if (r < M) {
      const float4 vx = *reinterpret_cast<const float4*>(x + r * C + c4);
      const float4 vd = *reinterpret_cast<const float4*>(dy + r * C + c4);
      float4 va = make_float4(0.f, 0.f, 0.f, 0.f);
      if (addend) va = *reinterpret_cast<const float4*>(addend + r * C + c4);
      const float xv[4] = {vx.x, vx.y, vx.z, vx.w}, dv[4] = {vd.x, vd.y, vd.z, vd.w}, av[4] = {va.x, va.y, va.z, va.w};
      float ov[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float xh = (xv[e] - mu[e]) * rstd[e];
        float dz = dv[e];
        if (relu && xh * gm[e] + bt[e] <= 0.0f) dz = 0.0f;
        const float rr = dz - k1[e] - xh * k2[e];
        ov[e] = gm[e] * rstd[e] * rr;
        if (addend) ov[e] += av[e];
      }
      *reinterpret_cast<float4*>(dx + r * C + c4) = make_float4(ov[0], ov[1], ov[2], ov[3]);
    }
  }
}

// chunks up to which a level takes finish + apply as ONE launch in which every workgroup redoes the (chunked) finish of its
// channel group: G = 1 is round 5's form (rows < 4,096); round 6: up to WSIS_BN_SMALL_G chunks (default 4: rows < ~10,240 --
// level 2 of a scene, level 3 of four) -- a workgroup re-reads <= 320 partial rows of 32 channels (80 KB from L2) instead
// of the step paying a launch + a kernel boundary per layer and direction.  Same bits as the two launches (the chunk order
// of the ticketed second level).  Read per call.
static int bn_small_gmax() {
  const char* e = getenv("WSIS_BN_SMALL_G");
  const int g = e ? atoi(e) : 4;
  return g < 1 ? 1 : g > 8 ? 8 : g;
}
static bool bn_small_fused_on() {      // WSIS_BN_SMALL_FUSED=0 (read per call): the two launches
  const char* e = getenv("WSIS_BN_SMALL_FUSED");
  return !e || atoi(e) != 0;
}

// ---- small inputs (M <= bn_small_rows): the whole reduction in ONE launch, one workgroup per channel group of 4.
// Two launches (partial + final) of a few microseconds each are pure latency at the deep UNet levels (<= 4096 rows;
// above that the 16-byte-per-row slices of one workgroup per channel group stall on cache-line throughput).
// 1024 threads stride over the rows (<= 8 rows each, every load in flight at once: the kernel is one memory latency
// long, not M/256 of them), then a fixed reduction in fp64: xor-shuffle inside each wave, 16 wave results through
// LDS, summed in wave order -> deterministic.
int64_t bn_small_rows() {  // WSIS_BN_SMALL_ROWS: rows up to which the one-launch reduction is used
  static const int64_t v = [] {
    const char* e = getenv("WSIS_BN_SMALL_ROWS");
    return e ? (int64_t)atoll(e) : (int64_t)4096;
  }();
  return v;
}
constexpr int BN_SMALL_THREADS = 1024;

__device__ __forceinline__ void block_sum2_f64(double (&a)[4], double (&b)[4], double* sh) {
  // sh: [BN_SMALL_THREADS / 64][8] doubles
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      a[e] += __shfl_xor(a[e], off, 64);
      b[e] += __shfl_xor(b[e], off, 64);
    }
  }
  if (lane == 0) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      sh[wave * 8 + e] = a[e];
      sh[wave * 8 + 4 + e] = b[e];
    }
  }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    double sa = 0.0, sb = 0.0;
    for (int w = 0; w < BN_SMALL_THREADS / 64; ++w) {
      sa += sh[w * 8 + e];
      sb += sh[w * 8 + 4 + e];
    }
    a[e] = sa;
    b[e] = sb;
  }
}

__global__ __launch_bounds__(BN_SMALL_THREADS) void bn_stats_small_kernel(const float* __restrict__ x, int64_t M, int C,
                                                                    float* __restrict__ mean, float* __restrict__ var,
                                                                    float* __restrict__ running_mean,
                                                                    float* __restrict__ running_var, float momentum) {
  __shared__ double sh[BN_SMALL_THREADS / 64 * 8];
  const int c0 = blockIdx.x * 4;
  const bool vec = (C & 3) == 0;
  float sa[4] = {0.f, 0.f, 0.f, 0.f}, sb[4] = {0.f, 0.f, 0.f, 0.f};
  float K[4];                                 // pivot: sums of x - x[0][c] (see bn_stats_partial_kernel)
#pragma unroll
  for (int e = 0; e < 4; ++e) K[e] = (c0 + e < C) ? bn_pivot(x, M, C, c0 + e) : 0.0f;
#pragma unroll 8
  for (int64_t r = threadIdx.x; r < M; r += BN_SMALL_THREADS) {
    float v[4];
    if (vec) {
      const f4 t = ld4(x + r * C + c0, true);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = t.v[e] - K[e];
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (c0 + e < C) ? x[r * C + c0 + e] - K[e] : 0.0f;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      sa[e] += v[e];
      sb[e] += v[e] * v[e];
    }
  }
  double a[4], b[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    a[e] = sa[e];
    b[e] = sb[e];
  }
  block_sum2_f64(a, b, sh);
  if (threadIdx.x < 4 && c0 + (int)threadIdx.x < C) {
    const int e = threadIdx.x, c = c0 + e;
    const double n = (double)M;
    const double ms = a[e] / n;
    const double mu = (double)K[e] + ms;
    double v = b[e] / n - ms * ms;
    if (v < 0.0) v = 0.0;
    mean[c] = (float)mu;
    var[c] = (float)v;
    if (running_mean) {
      const double unb = n > 1 ? v * n / (n - 1) : v;
      running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mu);
      running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unb);
    }
  }
}

__global__ __launch_bounds__(BN_SMALL_THREADS) void bn_bwd_small_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ mean,
    const float* __restrict__ var, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
    int relu, int64_t M, int C, float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ double sh[BN_SMALL_THREADS / 64 * 8];
  const int c0 = blockIdx.x * 4;
  const bool vec = (C & 3) == 0;
  float mu[4], rstd[4], gm[4], bt[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int cc = min(c0 + e, C - 1);
    mu[e] = mean[cc];
    rstd[e] = rsqrtf(var[cc] + eps);
    gm[e] = gamma ? gamma[cc] : 1.0f;
    bt[e] = beta ? beta[cc] : 0.0f;
  }
  float sa[4] = {0.f, 0.f, 0.f, 0.f}, sb[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
  for (int64_t r = threadIdx.x; r < M; r += BN_SMALL_THREADS) {
    float xv[4], dv[4];
    if (vec) {
      const f4 tx = ld4(x + r * C + c0, true), td = ld4(dy + r * C + c0, true);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        xv[e] = tx.v[e];
        dv[e] = td.v[e];
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool ok = c0 + e < C;
        xv[e] = ok ? x[r * C + c0 + e] : 0.0f;
        dv[e] = ok ? dy[r * C + c0 + e] : 0.0f;
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float xh = (xv[e] - mu[e]) * rstd[e];
      float dz = dv[e];
      if (relu && xh * gm[e] + bt[e] <= 0.0f) dz = 0.0f;
      sa[e] += dz;
      sb[e] += dz * xh;
    }
  }
  double a[4], b[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    a[e] = sa[e];
    b[e] = sb[e];
  }
  block_sum2_f64(a, b, sh);
  if (threadIdx.x < 4 && c0 + (int)threadIdx.x < C) {
    dbeta[c0 + threadIdx.x] = (float)a[threadIdx.x];
    dgamma[c0 + threadIdx.x] = (float)b[threadIdx.x];
  }
}

int bn_nblk(int64_t M, int Cp) { return (int)ceil_div(M > 0 ? M : 1, (int64_t)bn_rows_per_wg(Cp)); }

}  // namespace

extern "C" {

int64_t wsis_bn_workspace_bytes(int64_t M, int32_t C) {
  if (M < 0 || C < 1) return -1;
  const int64_t Cp = (C + 3) / 4 * 4;
  return (int64_t)bn_nblk(M, (int)Cp) * 2 * Cp * (int64_t)sizeof(float) + 256;
}

int wsis_bn_stats(const float* d_x, int64_t M, int32_t C, float* d_mean, float* d_var, float* d_running_mean,
                  float* d_running_var, float momentum, void* d_ws, int64_t ws_bytes, void* stream) {
  WSIS_REQUIRE(M >= 1 && C >= 1 && d_x && d_mean && d_var && d_ws, "bad args");
  WSIS_REQUIRE(ws_bytes >= wsis_bn_workspace_bytes(M, C), "workspace too small");
  WSIS_REQUIRE((d_running_mean == nullptr) == (d_running_var == nullptr), "running stats come in pairs");
  const int Cp = (C + 3) / 4 * 4;
  const int nblk = bn_nblk(M, Cp);
  float* partial = static_cast<float*>(d_ws);
  hipStream_t st = as_stream(stream);
  WSIS_REQUIRE(Cp / 4 <= BN_THREADS, "C > 1024 is not supported");
  if (M <= bn_small_rows()) {
    hipLaunchKernelGGL(bn_stats_small_kernel, dim3(Cp / 4), dim3(BN_SMALL_THREADS), 0, st, d_x, M, C, d_mean, d_var,
                       d_running_mean, d_running_var, momentum);
    WSIS_LAUNCH_CHECK();
    return WSIS_OK;
  }
  hipLaunchKernelGGL(bn_stats_partial_kernel, dim3(nblk), dim3(BN_THREADS), 0, st, d_x, M, C, Cp, partial);
  WSIS_LAUNCH_CHECK();
  hipLaunchKernelGGL(bn_stats_final_kernel, dim3(C), dim3(64), 0, st, partial, nblk, C, Cp, M, d_x, d_mean,
                     d_var, d_running_mean, d_running_var, momentum);
  WSIS_LAUNCH_CHECK();
  return WSIS_OK;
}

// ---- column sums of x [M, C] (bias gradient of a Linear layer over M rows): two levels in ONE launch, fixed order.
// 256 threads = (C/4 float4 column lanes) x (row lanes); every workgroup sums a block of rows, the one that arrives
// last adds the block results in block order.

// CROSS: the rows read (IN) or the row written (OUT) are handed between workgroups of this launch -> sc1 accesses
__device__ __forceinline__ void colsum_block(const bool IN, const bool OUT, const float* __restrict__ x, int64_t lo,
                                             int64_t hi, int C4, float4* red, float* __restrict__ out) {
  const int RL = 256 / C4;
  const int cl = threadIdx.x % C4, rl = threadIdx.x / C4;
  float4 a = {0.f, 0.f, 0.f, 0.f};
  if (rl < RL) {
    const float4* x4 = reinterpret_cast<const float4*>(x);
#pragma unroll 8
    for (int64_t r = lo + rl; r < hi; r += RL) {
      float4 v;
      if (IN) {
        const float* p = x + (r * C4 + cl) * 4;
        v = make_float4(ld_sc1(p), ld_sc1(p + 1), ld_sc1(p + 2), ld_sc1(p + 3));
      } else {
        v = x4[r * C4 + cl];
      }
      a.x += v.x;
      a.y += v.y;
      a.z += v.z;
      a.w += v.w;
    }
    red[rl * C4 + cl] = a;
  }
  __syncthreads();
  if (threadIdx.x < C4) {
    float4 s = red[threadIdx.x];
    for (int j = 1; j < RL; ++j) {      // fixed order
      const float4 v = red[j * C4 + threadIdx.x];
      s.x += v.x;
      s.y += v.y;
      s.z += v.z;
      s.w += v.w;
    }
    if (OUT) {
      float* p = out + threadIdx.x * 4;
      st_sc1(p, s.x);
      st_sc1(p + 1, s.y);
      st_sc1(p + 2, s.z);
      st_sc1(p + 3, s.w);
    } else {
      reinterpret_cast<float4*>(out)[threadIdx.x] = s;
    }
  }
}

__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, int64_t M, int C, int64_t per,
                                                     float* __restrict__ chunk, float* __restrict__ out,
                                                     unsigned* __restrict__ ticket) {
  __shared__ float4 red[256];
  __shared__ int s_last;
  const int C4 = C >> 2;
  const int64_t lo = (int64_t)blockIdx.x * per;
  const int64_t hi = lo + per < M ? lo + per : M;
  if (gridDim.x == 1) {
    colsum_block(false, false, x, lo, hi, C4, red, out);
    return;
  }
  colsum_block(false, true, x, lo, hi, C4, red, chunk + (int64_t)blockIdx.x * C);
  wait_stores_left();
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned t = atomicAdd(ticket, 1u);
    s_last = t == gridDim.x - 1;
    if (s_last) *ticket = 0u;
  }
  __syncthreads();
  if (!s_last) return;
  __syncthreads();            // `red` of the first pass has been read by everyone
  colsum_block(true, false, chunk, 0, gridDim.x, C4, red, out);
}

// ticket row of a two-level reduction: the caller's sync slot (nullptr without one, or with WSIS_BN_TICKET=0: the finish
// runs as a second launch)
static unsigned* bn_tickets(void* d_sync) {
  static int on = -1;
  if (on < 0) {
    const char* e = tune_env("WSIS_BN_TICKET");
    on = e ? atoi(e) : 1;
  }
  if (!on || !d_sync) return nullptr;
  return static_cast<SyncSlot*>(d_sync)->ticket;
}


int64_t wsis_bn_stats_finalize_workspace_bytes(int64_t n_part, int32_t C) {
  return (int64_t)bn_fin_chunks(n_part) * 3 * C * (int64_t)sizeof(double) + 256;
}

int wsis_bn_stats_finalize(const float* d_partials, int64_t n_part, int64_t M, int32_t C, float* d_mean, float* d_var,
                           float* d_running_mean, float* d_running_var, float momentum, void* d_ws, int64_t ws_bytes,
                           void* d_sync, void* stream) {
  WSIS_REQUIRE(n_part >= 1 && M >= 1 && C >= 1 && d_partials && d_mean && d_var, "bad args");
  WSIS_REQUIRE(n_part < ((int64_t)1 << 31), "too many partials");
  WSIS_REQUIRE((d_running_mean == nullptr) == (d_running_var == nullptr), "running stats come in pairs");
  WSIS_REQUIRE(n_part == (M + 31) / 32, "one partial per 32-row slice");
  const int G = bn_fin_chunks(n_part);
  WSIS_REQUIRE(G == 1 || (d_ws && ws_bytes >= wsis_bn_stats_finalize_workspace_bytes(n_part, C)), "workspace too small");
  double* chunk = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(d_ws) + 255) & ~(uintptr_t)255);
  WSIS_REQUIRE(C <= 512, "more than 512 channels");
  unsigned* tickets = bn_tickets(d_sync);
  hipLaunchKernelGGL(bn_stats_chunk_centred_kernel, dim3(G, (C + 31) / 32), dim3(256), 0, as_stream(stream), d_partials,
                     (int)n_part, C, M, chunk, d_mean, d_var, d_running_mean, d_running_var, momentum, tickets);
  WSIS_LAUNCH_CHECK();
  if (G > 1 && !tickets) {
    hipLaunchKernelGGL(bn_stats_final_centred_kernel, dim3((C + 3) / 4), dim3(256), 0, as_stream(stream), chunk, G, C, M,
                       d_mean, d_var, d_running_mean, d_running_var, momentum);
    WSIS_LAUNCH_CHECK();
  }
  return WSIS_OK;
}

// grid of a producer-consumer launch over M x C elements: every workgroup resident (<= 2 per CU of this device),
// a multiple of the channel groups; 0 when the one-launch form does not apply
static int bn_fused_grid(int64_t M, int C, int need_chunk_wgs, int which = 0) {
#if !WSIS_EXPERIMENTAL
  return 0;      // the polled producer / consumer form (WSIS_BN_FUSED_APPLY) is in the EXPERIMENTAL build only
#endif
  // read per call (a test switches the form on): WSIS_BN_FUSED_APPLY for both directions, WSIS_BN_FUSED_FWD / _BWD per
  // direction.  Default OFF since round 3: with the flag / ticket words in caller slots (one more arrival counter per
  // workgroup) and the convolutions on one-launch plans the two-launch form measures faster -- 10.37 / 10.38 ms per C2
  // step against 10.59 / 10.63 with both directions fused, 10.48 / 10.46 and 10.50 / 10.53 with one of them
  // (alternating runs on one box) -- and no workgroup of the default path ever waits for another one inside a launch
  const char* e = getenv("WSIS_BN_FUSED_APPLY");
  const int on = e ? atoi(e) : 0;
  const char* ed = getenv(which ? "WSIS_BN_FUSED_BWD" : "WSIS_BN_FUSED_FWD");
  const int on_dir = ed ? atoi(ed) : on;
  const char* g = tune_env("WSIS_BN_FUSED_GRID");
  int gmax = g ? atoi(g) : 256;
  if (gmax < 64 || gmax > 512) gmax = 256;
  int n_cu = 0;
  {   // per call: the CU count of the CURRENT device (cheap attribute query, no cache shared between devices)
    int dev = 0, v = 0;
    n_cu = (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess) ? v : 0;
  }
  if (!on_dir || (C & 3) != 0) return 0;
  const int cw = C >> 2;
  const int64_t work = (M * C) >> 2;
  int gcap = (work > 600000 && gmax == 256) ? 512 : gmax;     // level 0: two workgroups per CU for the apply pass
  if (gcap > 2 * n_cu) gcap = 2 * n_cu;
  if (gcap < 1) return 0;
  int grid = gcap - (gcap % cw);
  const int need = grid_for(work, 256);
  if (need < grid) {
    grid = need;
    if (grid > cw) grid -= grid % cw;
  }
  return (grid >= need_chunk_wgs && grid >= 1) ? grid : 0;
}

int wsis_bn_stats_finalize_apply(const float* d_partials, int64_t n_part, int64_t M, int32_t C, float* d_mean,
                                 float* d_var, float* d_running_mean, float* d_running_var, float momentum,
                                 const float* d_x, const float* d_gamma, const float* d_beta, float eps, int32_t relu,
                                 float* d_y, void* d_ws, int64_t ws_bytes, void* d_sync, void* stream) {
  WSIS_REQUIRE(n_part >= 1 && M >= 1 && C >= 1 && d_partials && d_mean && d_var && d_x && d_y, "bad args");
  WSIS_REQUIRE(n_part == (M + 31) / 32 && n_part < ((int64_t)1 << 31), "one partial per 32-row slice");
  WSIS_REQUIRE(C <= 512, "more than 512 channels");
  hipStream_t st = as_stream(stream);
  const int G = bn_fin_chunks(n_part);
  const int CG = (C + 31) / 32;
  WSIS_REQUIRE(G == 1 || (d_ws && ws_bytes >= wsis_bn_stats_finalize_workspace_bytes(n_part, C)), "workspace too small");
  double* chunk = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(d_ws) + 255) & ~(uintptr_t)255);
  unsigned* tickets = bn_tickets(d_sync);
  // the one-launch form needs: a sync slot (tickets + flag), every workgroup resident, vector rows
  const int grid = tickets ? bn_fused_grid(M, C, G * CG) : 0;
  const bool fused = grid > 0;
  if (G <= bn_small_gmax() && (G == 1 || tickets) && (C & 3) == 0 && bn_small_fused_on() &&
      (reinterpret_cast<uintptr_t>(d_x) & 15) == 0 &&
      (reinterpret_cast<uintptr_t>(d_y) & 15) == 0) {      // few chunks: finish + apply without any hand-off
    hipLaunchKernelGGL(bn_small_finish_apply_kernel, dim3((unsigned)ceil_div(M, BN_SF_ROWS), (unsigned)CG), dim3(256), 0,
                       st, d_partials, (int)n_part, (int)C, M, d_mean, d_var, d_running_mean, d_running_var, momentum, d_x,
                       d_gamma, d_beta, eps, (int)relu, d_y, G);
    WSIS_LAUNCH_CHECK();
    return WSIS_OK;
  }
  if (!fused) {
    const int rc = wsis_bn_stats_finalize(d_partials, n_part, M, C, d_mean, d_var, d_running_mean, d_running_var, momentum,
                                          d_ws, ws_bytes, d_sync, stream);
    if (rc != WSIS_OK) return rc;
    return wsis_bn_apply(d_x, d_mean, d_var, d_gamma, d_beta, eps, relu, d_y, M, C, stream);
  }
  hipLaunchKernelGGL(bn_finalize_apply_kernel, dim3(grid), dim3(256), 0, st, d_partials, (int)n_part, (int)C, M, chunk, G,
                     d_mean, d_var, d_running_mean, d_running_var, momentum, static_cast<SyncSlot*>(d_sync), d_x, d_gamma,
                     d_beta, eps, (int)relu, d_y);
  WSIS_LAUNCH_CHECK();
  return WSIS_OK;
}

// grid of an apply pass over `work` vector items in `cw` channel groups: eight items per thread (four trips of two) where
// that still leaves 512 workgroups, else one workgroup per 256 items up to 512 -- rounded down to a multiple of the
// channel groups (a thread then keeps its channel group for its whole walk).  The round-4 grid (one item per thread up
// to 2,048 workgroups) left a level-0 thread 2.3 items -- a third trip that a third of the threads take -- and a level-1
// thread ONE: nothing in flight behind it.  tools/bn_bench.py, apply / backward apply in us: 153,685 x 32: 9.0 / 14.7 ->
// 7.4 / 12.0; 26,819 x 64: 9.5 / 16.4 -> 4.7 / 9.1; 26,819 x 128: 11.3 / 18.4 -> 6.5 / 10.2; small levels unchanged.
static int bn_apply_grid(int64_t work, int cw, int pt_unused) {
  (void)pt_unused;
  const int pt = tune_int("WSIS_BN_APPLY_PT", 8);
  int64_t g;
  if (pt < 0) {      // the round-4 grid
    g = grid_for(work, 256);
  } else {
    g = ceil_div(work, (int64_t)256 * pt);
    if (g < 512) {
      g = ceil_div(work, 256);
      if (g > 512) g = 512;
    }
    if (g > 16384) g = 16384;
  }
  if (g < 1) g = 1;
  if (g > cw) g -= g % cw;
  return (int)g;
}

int wsis_bn_apply(const float* d_x, const float* d_mean, const float* d_var, const float* d_gamma,
                  const float* d_beta, float eps, int32_t relu, float* d_y, int64_t M, int32_t C, void* stream) {
  WSIS_REQUIRE(M >= 0 && C >= 1, "bad sizes");
  if (M == 0) return WSIS_OK;
  WSIS_REQUIRE(d_x && d_mean && d_var && d_y, "null pointer");
  const int64_t work = (C & 3) == 0 ? (M * C) >> 2 : M * C;
  // grid rounded to a multiple of the channel groups: a thread then keeps one channel group for its whole walk
  const int cw = (C & 3) == 0 ? C >> 2 : C;
  const int grid = bn_apply_grid(work, cw, 2);
  hipLaunchKernelGGL(bn_apply_kernel, dim3(grid), dim3(256), 0, as_stream(stream), d_x, d_mean, d_var,
                     d_gamma, d_beta, eps, relu, d_y, M, C);
  WSIS_LAUNCH_CHECK();
  return WSIS_OK;
}

int wsis_bn_bwd_from_partials(const float* d_partials, int64_t n_part, const float* d_x, const float* d_dy,
                              const float* d_mean, const float* d_var, const float* d_gamma, const float* d_beta,
                              float eps, int32_t relu, float* d_dx, float* d_dgamma, float* d_dbeta,
                              const float* d_addend, int64_t M, int32_t C, void* d_ws, int64_t ws_bytes, void* d_sync,
                              void* stream) {
  WSIS_REQUIRE(M >= 1 && C >= 1 && d_partials && d_x && d_dy && d_mean && d_var && d_dgamma && d_dbeta, "bad args");
  WSIS_REQUIRE(n_part == (M + 31) / 32 && n_part < ((int64_t)1 << 31), "one partial per 32-row slice");
  const int G = bn_fin_chunks(n_part);
  WSIS_REQUIRE(G == 1 || (d_ws && ws_bytes >= wsis_bn_stats_finalize_workspace_bytes(n_part, C)), "workspace too small");
  double* chunk = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(d_ws) + 255) & ~(uintptr_t)255);
  hipStream_t st = as_stream(stream);
  WSIS_REQUIRE(C <= 512, "more than 512 channels");
  unsigned* tickets = bn_tickets(d_sync);
  if (d_dx && G <= bn_small_gmax() && (G == 1 || tickets) && (C & 3) == 0 && bn_small_fused_on() &&
      (reinterpret_cast<uintptr_t>(d_x) & 15) == 0 &&
      (reinterpret_cast<uintptr_t>(d_dy) & 15) == 0 && (reinterpret_cast<uintptr_t>(d_dx) & 15) == 0 &&
      (reinterpret_cast<uintptr_t>(d_addend) & 15) == 0) {
    hipLaunchKernelGGL(bn_small_bwd_finish_apply_kernel, dim3((unsigned)ceil_div(M, BN_SF_ROWS), (unsigned)((C + 31) / 32)),
                       dim3(256), 0, st, d_partials, (int)n_part, (int)C, M, d_x, d_dy, d_mean, d_var, d_gamma, d_beta,
                       d_addend, eps, (int)relu, d_dx, d_dgamma, d_dbeta, G);
    WSIS_LAUNCH_CHECK();
    return WSIS_OK;
  }
  if (d_dx) {     // reduction finish + apply in one launch (the form of wsis_bn_stats_finalize_apply)
    const int CG = (C + 31) / 32;
    const int fgrid = tickets ? bn_fused_grid(M, C, G * CG, 1) : 0;
    if (fgrid > 0) {
      hipLaunchKernelGGL(bn_bwd_finish_apply_kernel, dim3(fgrid), dim3(256), 0, st, d_partials, (int)n_part, (int)C, chunk, G,
                         d_dbeta, d_dgamma, static_cast<SyncSlot*>(d_sync), d_x, d_dy, d_mean,
                         d_var, d_gamma, d_beta, d_addend, eps, (int)relu, d_dx, M);
      WSIS_LAUNCH_CHECK();
      return WSIS_OK;
    }
  }
  hipLaunchKernelGGL(bn_sum_chunk_kernel, dim3(G, (C + 31) / 32), dim3(256), 0, st, d_partials, (int)n_part, C, chunk,
                     d_dbeta, d_dgamma, tickets);
  WSIS_LAUNCH_CHECK();
  if (G > 1 && !tickets) {
    hipLaunchKernelGGL(bn_sum_final_kernel, dim3((C + 3) / 4), dim3(256), 0, st, chunk, G, C, d_dbeta, d_dgamma);
    WSIS_LAUNCH_CHECK();
  }
  if (d_dx) {
    const int64_t work = (C & 3) == 0 ? (M * C) >> 2 : M * C;
    const int cw = (C & 3) == 0 ? C >> 2 : C;
    const int grid = bn_apply_grid(work, cw, 2);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid), dim3(256), 0, st, d_x, d_dy, d_mean, d_var, d_gamma, d_beta,
                       d_dgamma, d_dbeta, d_addend, eps, relu, 1, d_dx, M, C);
    WSIS_LAUNCH_CHECK();
  }
  return WSIS_OK;
}

int wsis_bn_bwd_apply(const float* d_x, const float* d_dy, const float* d_mean, const float* d_var,
                      const float* d_gamma, const float* d_beta, const float* d_sum_dz_xhat, const float* d_sum_dz,
                      float eps, int32_t relu, float* d_dx, const float* d_addend, int64_t M, int32_t C, void* stream) {
  WSIS_REQUIRE(M >= 1 && C >= 1 && d_x && d_dy && d_mean && d_var && d_sum_dz_xhat && d_sum_dz && d_dx, "bad args");
  const int64_t work = (C & 3) == 0 ? (M * C) >> 2 : M * C;
  const int cw = (C & 3) == 0 ? C >> 2 : C;
  const int grid = bn_apply_grid(work, cw, 2);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid), dim3(256), 0, as_stream(stream), d_x, d_dy, d_mean, d_var, d_gamma,
                     d_beta, d_sum_dz_xhat, d_sum_dz, d_addend, eps, relu, 1, d_dx, M, C);
  WSIS_LAUNCH_CHECK();
  return WSIS_OK;
}

int wsis_bn_bwd(const float* d_x, const float* d_dy, const float* d_mean, const float* d_var,
                const float* d_gamma, const float* d_beta, float eps, int32_t relu, int32_t training,
                float* d_dx, float* d_dgamma, float* d_dbeta, const float* d_addend, int64_t M, int32_t C, void* d_ws,
                int64_t ws_bytes, void* stream) {
  WSIS_REQUIRE(M >= 1 && C >= 1 && d_x && d_dy && d_mean && d_var && d_dgamma && d_dbeta && d_ws, "bad args");
  WSIS_REQUIRE(ws_bytes >= wsis_bn_workspace_bytes(M, C), "workspace too small");
  const int Cp = (C + 3) / 4 * 4;
  const int nblk = bn_nblk(M, Cp);
  float* partial = static_cast<float*>(d_ws);
  hipStream_t st = as_stream(stream);
  WSIS_REQUIRE(Cp / 4 <= BN_THREADS, "C > 1024 is not supported");
  if (M <= bn_small_rows()) {
    hipLaunchKernelGGL(bn_bwd_small_kernel, dim3(Cp / 4), dim3(BN_SMALL_THREADS), 0, st, d_x, d_dy, d_mean, d_var, d_gamma,
                       d_beta, eps, relu, M, C, d_dgamma, d_dbeta);
    WSIS_LAUNCH_CHECK();
  } else {
    hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(nblk), dim3(BN_THREADS), 0, st, d_x, d_dy, d_mean, d_var, d_gamma,
                       d_beta, eps, relu, M, C, Cp, partial);
    WSIS_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_bwd_final_kernel, dim3(C), dim3(64), 0, st, partial, nblk, C, Cp, d_dgamma, d_dbeta);
    WSIS_LAUNCH_CHECK();
  }
  if (d_dx) {
    // grid rounded to a multiple of the channel groups: a thread then keeps one channel group for its whole walk
    const int64_t work = (C & 3) == 0 ? (M * C) >> 2 : M * C;
    const int cw = (C & 3) == 0 ? C >> 2 : C;
    const int grid = bn_apply_grid(work, cw, 2);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid), dim3(256), 0, st, d_x, d_dy, d_mean, d_var, d_gamma, d_beta,
                       d_dgamma, d_dbeta, d_addend, eps, relu, training, d_dx, M, C);
    WSIS_LAUNCH_CHECK();
  }
  return WSIS_OK;
}

int64_t wsis_sync_bytes(void) { return (int64_t)kSyncSlotsTotal * (int64_t)sizeof(SyncSlot); }

int64_t wsis_colsum_workspace_bytes(int64_t M, int32_t C) {
  if (M < 0 || C < 4) return -1;
  // <= 128 chunks: every chunk ends in an agent-scope ticket add on ONE word (~50 ns each, serialised across the XCDs)
  // and the last workgroup walks the chunk rows -- 781 chunks of 256 rows made the [199790, 20] bias gradient 58 us
  int64_t per = (M + 127) / 128;
  if (per < 256) per = 256;
  const int64_t G = (M + per - 1) / per;
  return (G > 1 ? G : 1) * (int64_t)C * (int64_t)sizeof(float) + 256;
}

int wsis_colsum(const float* d_x, int64_t M, int32_t C, float* d_out, void* d_ws, int64_t ws_bytes, void* d_sync,
                void* stream) {
  WSIS_REQUIRE(M >= 0 && C >= 4 && C % 4 == 0 && C <= 1024 && d_out, "colsum: C must be a multiple of 4, <= 1024");
  hipStream_t st = as_stream(stream);
  if (M == 0) {
    WSIS_HIP_CHECK(hipMemsetAsync(d_out, 0, sizeof(float) * (size_t)C, st));
    return WSIS_OK;
  }
  WSIS_REQUIRE(d_x && (reinterpret_cast<uintptr_t>(d_x) & 15) == 0 && (reinterpret_cast<uintptr_t>(d_out) & 15) == 0,
               "colsum: 16-byte alignment");
  int64_t per = (M + 127) / 128;      // as wsis_colsum_workspace_bytes
  if (per < 256) per = 256;
  const int64_t G = (M + per - 1) / per;
  float* chunk = nullptr;
  unsigned* ticket = nullptr;
  if (G > 1) {
    WSIS_REQUIRE(d_ws && ws_bytes >= wsis_colsum_workspace_bytes(M, C), "workspace too small");
    chunk = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(d_ws) + 255) & ~(uintptr_t)255);
    WSIS_REQUIRE(d_sync, "colsum over more than one chunk needs a sync slot");
    ticket = static_cast<SyncSlot*>(d_sync)->ticket;
  }
  hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)G), dim3(256), 0, st, d_x, M, (int)C, per, chunk, d_out, ticket);
  WSIS_LAUNCH_CHECK();
  return WSIS_OK;
}

}  // extern "C"
