// Inter-superpoint affinity.
//
//  a16  edge attention  modules/model/backbone_3D_WSIS.py:218-249
//       logit_e = (q[u].k[v]) * scale * pos_e ; a = softmax over the out-edges of u ;
//       res[u] = sum_e a_e v[v_e].  One wavefront owns one source superpoint (CSR over u), so the
//       segment softmax and the weighted sum need no atomics and have a fixed order.
//  a17  dense S x S fp64 affinity matrix, masked row-normalised transition matrix, fp64 matrix
//       product on the f64 MFMA (v_mfma_f64_16x16x4_f64), column max / first argmax.
//       train_scannetv2.py:562-570, modules/datasets/scannetv2_dataset.py:679-721
#include "common.h"

using namespace wsis;

typedef double f64x4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// dot of two D-vectors spread over the 64 lanes (fixed order: lane-strided partials + butterfly)
__device__ __forceinline__ float wave_dot(const float* __restrict__ a, const float* __restrict__ b, int D,
                                          int lane) {
  float p = 0.0f;
  for (int d = lane; d < D; d += 64) p += a[d] * b[d];
  return wave_sum(p);
}

constexpr int AF_MAXDEG = 32;     // edges of a node the batched paths keep in LDS

__global__ __launch_bounds__(256) void edge_affinity_fwd_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
    const float* __restrict__ pos, const int64_t* __restrict__ ev, const int32_t* __restrict__ perm_u,
    const int32_t* __restrict__ off_u, float scale, float* __restrict__ aff, float* __restrict__ res,
    int64_t Su, int D) {
  __shared__ float s_lg[4][AF_MAXDEG];
  __shared__ int s_ve[4][AF_MAXDEG];
  __shared__ int s_e[4][AF_MAXDEG];
  const int lane = threadIdx.x & 63;
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  for (int64_t u = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); u < Su; u += nwaves) {
    const int beg = off_u[u], end = off_u[u + 1];
    const float* qu = q + u * D;
    if (D == 64 && end - beg <= AF_MAXDEG) {
      // the model's shape (64 features, a handful of edges per superpoint): the three passes below walk the edges one
      // by one, each with three dependent loads (edge id -> target -> row) and a wave-wide dot product -- 38 us for
      // 20 k edges.  Here the rows of eight edges are in flight together, the logits are kept (LDS, per wave) and the
      // arithmetic is the same in the same order: identical results.
      const int w = threadIdx.x >> 6, n = end - beg;
      float* lgs = s_lg[w];
      int* ves = s_ve[w];
      int* es = s_e[w];
      const float qd = qu[lane];
      for (int j0 = 0; j0 < n; j0 += 8) {
        int32_t e[8];
        int64_t ve[8];
        float kv[8], pe[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) e[i] = perm_u[beg + (j0 + i < n ? j0 + i : n - 1)];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          ve[i] = ev[e[i]];
          pe[i] = pos[e[i]];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) kv[i] = k[ve[i] * 64 + lane];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float lg = wave_sum(qd * kv[i]) * scale * pe[i];
          if (lane == 0 && j0 + i < n) {
            lgs[j0 + i] = lg;
            ves[j0 + i] = (int)ve[i];
            es[j0 + i] = e[i];
          }
        }
      }
      float mx = -INFINITY;
      for (int j = 0; j < n; ++j) mx = fmaxf(mx, lgs[j]);
      float tot = 0.0f;
      for (int j = 0; j < n; ++j) tot += expf(lgs[j] - mx);
      float r = 0.0f;
      for (int j0 = 0; j0 < n; j0 += 8) {
        float vv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) vv[i] = v[(int64_t)ves[j0 + i < n ? j0 + i : n - 1] * 64 + lane];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          if (j0 + i < n) {
            const float a = expf(lgs[j0 + i] - mx) / tot;
            if (lane == 0) aff[es[j0 + i]] = a;
            r += a * vv[i];
          }
        }
      }
      res[u * 64 + lane] = r;
      continue;
    }
    // pass 1: max logit
    float mx = -INFINITY;
    for (int j = beg; j < end; ++j) {
      const int32_t e = perm_u[j];
      const float lg = wave_dot(qu, k + ev[e] * D, D, lane) * scale * pos[e];
      mx = fmaxf(mx, lg);
    }
    // pass 2: sum of exp
    float tot = 0.0f;
    for (int j = beg; j < end; ++j) {
      const int32_t e = perm_u[j];
      const float lg = wave_dot(qu, k + ev[e] * D, D, lane) * scale * pos[e];
      tot += expf(lg - mx);
    }
    // pass 3: normalised affinity and weighted value sum
    for (int d0 = 0; d0 < D; d0 += 64) {
      const int d = d0 + lane;
      float r = 0.0f;
      for (int j = beg; j < end; ++j) {
        const int32_t e = perm_u[j];
        const int64_t ve = ev[e];
        const float lg = wave_dot(qu, k + ve * D, D, lane) * scale * pos[e];
        const float a = expf(lg - mx) / tot;
        if (d0 == 0 && lane == 0) aff[e] = a;
        if (d < D) r += a * v[ve * D + d];
      }
      if (d < D) res[u * D + d] = r;
    }
  }
}

// backward, pass over sources u: dq, dpos, and per-edge ds (stored in tmp[0:E])
__global__ __launch_bounds__(256) void edge_affinity_bwd_u_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
    const float* __restrict__ pos, const float* __restrict__ aff, const int64_t* __restrict__ ev,
    const int32_t* __restrict__ perm_u, const int32_t* __restrict__ off_u, float scale,
    const float* __restrict__ daff, const float* __restrict__ dres, float* __restrict__ dq,
    float* __restrict__ dpos, float* __restrict__ ds_out, int64_t Su, int D) {
  __shared__ float s_da[4][AF_MAXDEG];
  __shared__ float s_s[4][AF_MAXDEG];
  __shared__ int s_ve[4][AF_MAXDEG];
  __shared__ int s_e[4][AF_MAXDEG];
  const int lane = threadIdx.x & 63;
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  for (int64_t u = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); u < Su; u += nwaves) {
    const int beg = off_u[u], end = off_u[u + 1];
    const float* qu = q + u * D;
    const float* dru = dres + u * D;
    if (D == 64 && end - beg <= AF_MAXDEG) {      // batched form of the loops below: same arithmetic, same order
      const int w = threadIdx.x >> 6, n = end - beg;
      float* das = s_da[w];
      float* ss = s_s[w];
      int* ves = s_ve[w];
      int* es = s_e[w];
      const float qd = qu[lane], drd = dru[lane];
      for (int j0 = 0; j0 < n; j0 += 8) {
        int32_t e[8];
        int64_t ve[8];
        float kv[8], vv[8], dae[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) e[i] = perm_u[beg + (j0 + i < n ? j0 + i : n - 1)];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          ve[i] = ev[e[i]];
          dae[i] = daff ? daff[e[i]] : 0.0f;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          vv[i] = v[ve[i] * 64 + lane];
          kv[i] = k[ve[i] * 64 + lane];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          float da = wave_sum(drd * vv[i]);
          if (daff) da += dae[i];
          const float sd = wave_sum(qd * kv[i]) * scale;
          if (lane == 0 && j0 + i < n) {
            das[j0 + i] = da;
            ss[j0 + i] = sd;
            ves[j0 + i] = (int)ve[i];
            es[j0 + i] = e[i];
          }
        }
      }
      float dotsum = 0.0f;
      for (int j = 0; j < n; ++j) dotsum += aff[es[j]] * das[j];
      float gq = 0.0f;
      for (int j0 = 0; j0 < n; j0 += 8) {
        float kv[8], ae[8], pe[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int jj = j0 + i < n ? j0 + i : n - 1;
          kv[i] = k[(int64_t)ves[jj] * 64 + lane];
          ae[i] = aff[es[jj]];
          pe[i] = pos[es[jj]];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          if (j0 + i < n) {
            const float dlogit = ae[i] * (das[j0 + i] - dotsum);
            const float dsv = dlogit * pe[i] * scale;
            if (lane == 0) {
              dpos[es[j0 + i]] = dlogit * ss[j0 + i];
              ds_out[es[j0 + i]] = dsv;
            }
            gq += dsv * kv[i];
          }
        }
      }
      dq[u * 64 + lane] = gq;
      continue;
    }
    // sum_e a_e * da_e
    float dotsum = 0.0f;
    for (int j = beg; j < end; ++j) {
      const int32_t e = perm_u[j];
      float da = wave_dot(dru, v + ev[e] * D, D, lane);
      if (daff) da += daff[e];
      dotsum += aff[e] * da;
    }
    for (int d0 = 0; d0 < D; d0 += 64) {
      const int d = d0 + lane;
      float gq = 0.0f;
      for (int j = beg; j < end; ++j) {
        const int32_t e = perm_u[j];
        const int64_t ve = ev[e];
        float da = wave_dot(dru, v + ve * D, D, lane);
        if (daff) da += daff[e];
        const float a = aff[e];
        const float dlogit = a * (da - dotsum);
        const float s = wave_dot(qu, k + ve * D, D, lane) * scale;
        const float dsv = dlogit * pos[e] * scale;
        if (d0 == 0 && lane == 0) {
          dpos[e] = dlogit * s;
          ds_out[e] = dsv;
        }
        if (d < D) gq += dsv * k[ve * D + d];
      }
      if (d < D) dq[u * D + d] = gq;
    }
  }
}

// backward, pass over targets v: dk[v] = sum ds_e q[u_e], dv[v] = sum a_e dres[u_e]
__global__ __launch_bounds__(256) void edge_affinity_bwd_v_kernel(
    const float* __restrict__ q, const float* __restrict__ aff, const float* __restrict__ ds,
    const int64_t* __restrict__ eu, const int32_t* __restrict__ perm_v, const int32_t* __restrict__ off_v,
    const float* __restrict__ dres, float* __restrict__ dk, float* __restrict__ dv, int64_t S, int64_t Su,
    int D) {
  const int lane = threadIdx.x & 63;
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  for (int64_t t = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); t < S; t += nwaves) {
    const int beg = off_v[t], end = off_v[t + 1];
    for (int d0 = 0; d0 < D; d0 += 64) {
      const int d = d0 + lane;
      if (d >= D) continue;
      float gk = 0.0f, gv = 0.0f;
      for (int j = beg; j < end; ++j) {
        const int32_t e = perm_v[j];
        const int64_t ue = eu[e];
        gk += ds[e] * q[ue * D + d];
        gv += aff[e] * dres[ue * D + d];
      }
      dk[t * D + d] = gk;
      dv[t * D + d] = gv;
    }
  }
}

__global__ void dense_build_kernel(const int64_t* __restrict__ eu, const int64_t* __restrict__ ev,
                                   const float* __restrict__ aff, int64_t E, double* __restrict__ A,
                                   int64_t S) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < E;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t u = eu[e], w = ev[e];
    if (u >= 0 && u < S && w >= 0 && w < S) A[u * S + w] = (double)aff[e];
  }
}

// one workgroup per row: W = A*adj*sem, d = rowsum (0 -> 1), T0 = W/d
__global__ __launch_bounds__(256) void transition_kernel(const double* __restrict__ A,
                                                         const uint8_t* __restrict__ adj,
                                                         const int32_t* __restrict__ pred,
                                                         const float* __restrict__ conf,
                                                         const int32_t* __restrict__ label, int cls, float thr,
                                                         double* __restrict__ T0, int64_t S) {
  __shared__ double red[256];
  const int64_t r = blockIdx.x;
  const bool mr = pred[r] == cls && conf[r] > thr;
  const bool lr = label[r] == cls;
  double part = 0.0;
  for (int64_t j = threadIdx.x; j < S; j += blockDim.x) {
    const bool mj = pred[j] == cls && conf[j] > thr;
    const bool sem = (mr && mj) || (j == r && lr);
    const double w = sem ? A[r * S + j] * (double)adj[r * S + j] : 0.0;
    T0[r * S + j] = w;
    part += w;
  }
  red[threadIdx.x] = part;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  double dsum = red[0];
  if (dsum == 0.0) dsum = 1.0;
  for (int64_t j = threadIdx.x; j < S; j += blockDim.x) T0[r * S + j] = T0[r * S + j] / dsum;
}

// ---- fp64 GEMM on the f64 matrix cores ----------------------------------------------------
// block tile 64x64, BK = 16, 4 waves each owning a 32x32 quadrant (2x2 MFMA 16x16x4 tiles).
// v_mfma_f64_16x16x4_f64: A[i=lane&15][k=lane>>4], B[k=lane>>4][j=lane&15],
// C/D: col = lane&15, row = (lane>>4) + 4*reg.
constexpr int GB = 64;
constexpr int GK = 16;
__global__ __launch_bounds__(256) void dgemm_kernel(const double* __restrict__ A, const double* __restrict__ B,
                                                    double* __restrict__ C, int64_t M, int64_t N, int64_t Kd) {
  __shared__ double As[GB][GK + 1];
  __shared__ double Bs[GK][GB + 1];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wr = (wave >> 1) * 32, wc = (wave & 1) * 32;
  const int64_t row0 = (int64_t)blockIdx.y * GB, col0 = (int64_t)blockIdx.x * GB;
  f64x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.0;
  for (int64_t k0 = 0; k0 < Kd; k0 += GK) {
    for (int f = tid; f < GB * GK; f += 256) {
      const int r = f / GK, c = f % GK;
      const int64_t gr = row0 + r, gc = k0 + c;
      As[r][c] = (gr < M && gc < Kd) ? A[gr * Kd + gc] : 0.0;
    }
    for (int f = tid; f < GK * GB; f += 256) {
      const int r = f / GB, c = f % GB;
      const int64_t gr = k0 + r, gc = col0 + c;
      Bs[r][c] = (gr < Kd && gc < N) ? B[gr * N + gc] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < GK; kk += 4) {
      const int kq = kk + (lane >> 4);
      double a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = As[wr + i * 16 + (lane & 15)][kq];
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = Bs[kq][wc + j * 16 + (lane & 15)];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int64_t gr = row0 + wr + i * 16 + (lane >> 4) + 4 * e;
        const int64_t gc = col0 + wc + j * 16 + (lane & 15);
        if (gr < M && gc < N) C[gr * N + gc] = acc[i][j][e];
      }
}

__global__ void colmax_kernel(const double* __restrict__ T, const int32_t* __restrict__ label, int cls,
                              double* __restrict__ scores, int32_t* __restrict__ arg, int64_t S) {
  for (int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; j < S;
       j += (int64_t)gridDim.x * blockDim.x) {
    // np.max / np.argmax over axis 0 of a matrix whose non-class rows are zero: first max wins
    double best = 0.0;
    int32_t bi = -1;
    for (int64_t r = 0; r < S; ++r) {
      const double val = (label[r] == cls) ? T[r * S + j] : 0.0;
      if (bi < 0 || val > best) {
        best = val;
        bi = (int32_t)r;
      }
    }
    scores[j] = best;
    arg[j] = bi < 0 ? 0 : bi;
  }
}

// ---- a17, sparse form.  T0_c = rownorm(A * adj * sem_c) has the sparsity of the edge list (~9 entries per row) and only
// the rows of the superpoints LABELLED c of T0_c^(n+1) are ever looked at (scannetv2_dataset.py:707-714), so the dense
// S^3 product per class and iteration (24 GFLOP at S = 2,289) is a chain of (row vector) x (sparse matrix) products:
//   W0 = A * adj as CSR (class independent; one pass over the dense inputs per scene)
//   D[c][k] = sum_j W0[k][j] sem_c(k, j)      (0 -> 1)
//   x_0 = e_r^T T0_c ; x_{i+1}[j] = sum_k x_i[k] W0[k][j] sem_c(k, j) / D[c][k]   (k ascending: the order of a dot product)
// one wavefront per labelled superpoint r (class c = label[r]) keeps x in LDS; all classes run in ONE launch.
__global__ __launch_bounds__(256) void prop_nz_count_kernel(const double* __restrict__ A, const uint8_t* __restrict__ adj,
                                                            int64_t S, int32_t* __restrict__ rowcnt) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= S) return;
  int n = 0;
  for (int64_t j0 = 0; j0 < S; j0 += 64) {
    const int64_t j = j0 + lane;
    const bool nz = j < S && A[r * S + j] * (double)adj[r * S + j] != 0.0;
    n += __builtin_popcountll(__ballot(nz));
  }
  if (lane == 0) rowcnt[r] = n;
}

// exclusive scan of n ints by one workgroup (n ~ 10^3 - 10^4); out[n] = total
__global__ __launch_bounds__(1024) void prop_scan_kernel(const int32_t* __restrict__ in, int32_t* __restrict__ out, int n) {
  __shared__ int part[1024];
  const int per = (n + 1023) / 1024;
  const int lo = threadIdx.x * per, hi = lo + per < n ? lo + per : n;
  int s = 0;
  for (int i = lo; i < hi; ++i) s += in[i];
  part[threadIdx.x] = s;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const int v = (int)threadIdx.x >= d ? part[threadIdx.x - d] : 0;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  int base = threadIdx.x ? part[threadIdx.x - 1] : 0;
  for (int i = lo; i < hi; ++i) {
    out[i] = base;
    base += in[i];
  }
  if (threadIdx.x == 1023) out[n] = part[1023];
}

__global__ __launch_bounds__(256) void prop_nz_fill_kernel(const double* __restrict__ A, const uint8_t* __restrict__ adj,
                                                           int64_t S, const int32_t* __restrict__ rowptr,
                                                           int32_t* __restrict__ col, double* __restrict__ val,
                                                           int64_t cap) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= S) return;
  int base = rowptr[r];
  for (int64_t j0 = 0; j0 < S; j0 += 64) {          // ascending columns
    const int64_t j = j0 + lane;
    const double w = j < S ? A[r * S + j] * (double)adj[r * S + j] : 0.0;
    const unsigned long long m = __ballot(w != 0.0);
    if (w != 0.0) {
      const int pos = base + __builtin_popcountll(m & ((1ull << lane) - 1ull));
      if (pos < cap) {      // (the caller sized col / val from the non-zeros of A: never exceeded)
        col[pos] = (int32_t)j;
        val[pos] = w;
      }
    }
    base += __builtin_popcountll(m);
  }
}

__device__ __forceinline__ bool prop_sem(const int32_t* pred, const float* conf, const int32_t* label, int cls, float thr,
                                         int k, int j) {
  const bool mk = pred[k] == cls && conf[k] > thr;
  const bool mj = pred[j] == cls && conf[j] > thr;
  return (mk && mj) || (j == k && label[k] == cls);
}

// D[ci][k], ci = index of a PRESENT class; one wavefront per (k, ci); entries added in column order
__global__ __launch_bounds__(256) void prop_rowsum_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                          const double* __restrict__ val, const int32_t* __restrict__ pred,
                                                          const float* __restrict__ conf, const int32_t* __restrict__ label,
                                                          const int32_t* __restrict__ cls_of, float thr, int64_t S,
                                                          double* __restrict__ D) {
  const int lane = threadIdx.x & 63;
  const int64_t k = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (k >= S) return;
  const int cls = cls_of[blockIdx.y];
  double d = 0.0;
  const int lo = rowptr[k], hi = rowptr[k + 1];
  for (int e0 = lo; e0 < hi; e0 += 64) {
    const int e = e0 + lane;
    double w = 0.0;
    if (e < hi && prop_sem(pred, conf, label, cls, thr, (int)k, col[e])) w = val[e];
    // lanes in order: a fixed (sequential) sum over the row's entries
    for (int l = 0; l < 64 && e0 + l < hi; ++l) d += __shfl(w, l, 64);
  }
  if (lane == 0) D[(int64_t)blockIdx.y * S + k] = d == 0.0 ? 1.0 : d;
}

// one wavefront per superpoint r with label[r] = a present class: row r of T0_c^(iterations + 1) -> X[r][:]
__global__ __launch_bounds__(64) void prop_rows_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                       const double* __restrict__ val, const int32_t* __restrict__ pred,
                                                       const float* __restrict__ conf, const int32_t* __restrict__ label,
                                                       const int32_t* __restrict__ ci_of_cls, int class_num, float thr,
                                                       int iterations, int64_t S, const double* __restrict__ D,
                                                       double* __restrict__ X) {
  extern __shared__ __attribute__((aligned(16))) unsigned char prop_lds[];
  const int lane = threadIdx.x;
  const int r = blockIdx.x;
  const int cls = label[r];
  if (cls < 0 || cls >= class_num) return;
  const int ci = ci_of_cls[cls];
  if (ci < 0) return;
  const double* Dc = D + (int64_t)ci * S;
  double* cur = reinterpret_cast<double*>(prop_lds);
  double* nxt = cur + S;
  int32_t* nzl = reinterpret_cast<int32_t*>(nxt + S);       // ascending list of the non-zero positions of cur
  for (int64_t j = lane; j < S; j += 64) cur[j] = 0.0;
  {
    const int lo = rowptr[r], hi = rowptr[r + 1];
    const double d = Dc[r];
    for (int e = lo + lane; e < hi; e += 64) {
      const int j = col[e];
      if (prop_sem(pred, conf, label, cls, thr, r, j)) cur[j] = val[e] / d;
    }
  }
  __syncthreads();
  for (int it = 0; it < iterations; ++it) {
    int n = 0;
    for (int64_t j0 = 0; j0 < S; j0 += 64) {
      const int64_t j = j0 + lane;
      const bool nz = j < S && cur[j] != 0.0;
      const unsigned long long m = __ballot(nz);
      if (nz) nzl[n + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = (int32_t)j;
      n += __builtin_popcountll(m);
      if (j < S) nxt[j] = 0.0;
    }
    __syncthreads();      // (one wavefront per workgroup: orders the LDS passes for the compiler, costs nothing)
    for (int i = 0; i < n; ++i) {
      const int k = nzl[i];
      const double xk = cur[k];
      const double d = Dc[k];
      const int lo = rowptr[k], hi = rowptr[k + 1];
      for (int e = lo + lane; e < hi; e += 64) {      // the columns of one CSR row are distinct: no collisions
        const int j = col[e];
        if (prop_sem(pred, conf, label, cls, thr, k, j)) nxt[j] += xk * (val[e] / d);
      }
      __syncthreads();
    }
    double* t = cur;
    cur = nxt;
    nxt = t;
  }
  double* out = X + (int64_t)r * S;
  for (int64_t j = lane; j < S; j += 64) out[j] = cur[j];
}

// colmax_kernel for every present class at once (blockIdx.y = class index); rows of other classes are never read
__global__ void prop_colmax_kernel(const double* __restrict__ X, const int32_t* __restrict__ label,
                                   const int32_t* __restrict__ cls_of, double* __restrict__ scores,
                                   int32_t* __restrict__ arg, int64_t S) {
  const int cls = cls_of[blockIdx.y];
  const int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (j >= S) return;
  double best = 0.0;
  int32_t bi = -1;
  for (int64_t r = 0; r < S; ++r) {
    const double v = (label[r] == cls) ? X[r * S + j] : 0.0;
    if (bi < 0 || v > best) {
      best = v;
      bi = (int32_t)r;
    }
  }
  scores[(int64_t)blockIdx.y * S + j] = best;
  arg[(int64_t)blockIdx.y * S + j] = bi < 0 ? 0 : bi;
}

int waves_grid(int64_t segments) {
  int64_t g = ceil_div(segments, 4);
  if (g < 1) g = 1;
  if (g > 256 * 16) g = 256 * 16;
  return (int)g;
}

// ------------------------------------------------------------------------------------------------------------------
// Position encoding of the edge affinity (backbone_3D_WSIS.py:54-58, 222-224): pos_e = fc_position(centre[u_e] - centre[v_e])
// with fc_position = Linear(3,16) -> ReLU -> Linear(16,1).  The reference runs two gathers, a subtraction and three
// module launches forward and a dozen small launches backward (two of them [E,16] x [16,3] weight gradients) over
// E ~ 20 k rows; here one thread per edge does the whole chain, and the backward leaves the 81 parameter-gradient sums
// of a workgroup (thread order, through an LDS tile) as one partial row, summed by a second tiny launch.
constexpr int PE_H = 16;                           // hidden width
constexpr int PE_G = PE_H * 3 + PE_H + PE_H + 1;   // dW1 [16,3], db1 [16], dW2 [16], db2

__device__ __forceinline__ void pe_hidden(const float* __restrict__ centre, int64_t u, int64_t v, const float* __restrict__ W1,
                                          const float* __restrict__ b1, float (&d)[3], float (&hid)[PE_H]) {
#pragma unroll
  for (int k = 0; k < 3; ++k) d[k] = centre[u * 3 + k] - centre[v * 3 + k];
#pragma unroll
  for (int j = 0; j < PE_H; ++j) {
    float a = b1[j];
#pragma unroll
    for (int k = 0; k < 3; ++k) a += d[k] * W1[j * 3 + k];
    hid[j] = a;
  }
}

__global__ __launch_bounds__(256) void pos_enc_fwd_kernel(const float* __restrict__ centre, const int64_t* __restrict__ eu,
                                                          const int64_t* __restrict__ ev, const float* __restrict__ W1,
                                                          const float* __restrict__ b1, const float* __restrict__ W2,
                                                          const float* __restrict__ b2, float* __restrict__ pos, int64_t E) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= E) return;
  float d[3], hid[PE_H];
  pe_hidden(centre, eu[e], ev[e], W1, b1, d, hid);
  float p = b2[0];
#pragma unroll
  for (int j = 0; j < PE_H; ++j) p += fmaxf(hid[j], 0.0f) * W2[j];
  pos[e] = p;
}

constexpr int PE_TP = 256 + 4;          // row pitch of the workgroup's transposition tile (floats)

__global__ __launch_bounds__(256) void pos_enc_bwd_kernel(const float* __restrict__ centre, const int64_t* __restrict__ eu,
                                                          const int64_t* __restrict__ ev, const float* __restrict__ W1,
                                                          const float* __restrict__ b1, const float* __restrict__ W2,
                                                          const float* __restrict__ dpos, float* __restrict__ partial,
                                                          int64_t E) {
  // the 81 sums of a workgroup through an LDS tile [81][256]: every thread stores its 81 terms in its column, thread i
  // adds row i in thread order (16-byte reads).  As 81 wave butterflies of six cross-lane exchanges each the reduction
  // was 486 ds_bpermute per wave: 13 of the kernel's 20 us.
  extern __shared__ __attribute__((aligned(16))) float pe_tile[];
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  float g[PE_G];
#pragma unroll
  for (int i = 0; i < PE_G; ++i) g[i] = 0.0f;
  if (e < E) {
    float d[3], hid[PE_H];
    pe_hidden(centre, eu[e], ev[e], W1, b1, d, hid);
    const float gp = dpos[e];
#pragma unroll
    for (int j = 0; j < PE_H; ++j) {
      const float dh = hid[j] > 0.0f ? gp * W2[j] : 0.0f;
#pragma unroll
      for (int k = 0; k < 3; ++k) g[j * 3 + k] = dh * d[k];
      g[PE_H * 3 + j] = dh;
      g[PE_H * 4 + j] = gp * fmaxf(hid[j], 0.0f);
    }
    g[PE_G - 1] = gp;
  }
#pragma unroll
  for (int i = 0; i < PE_G; ++i) pe_tile[i * PE_TP + threadIdx.x] = g[i];
  __syncthreads();
  if (threadIdx.x < PE_G) {
    const float4* row = reinterpret_cast<const float4*>(pe_tile + threadIdx.x * PE_TP);
    float sum = 0.0f;
#pragma unroll 8
    for (int q = 0; q < 64; ++q) {
      const float4 v = row[q];
      sum += v.x;
      sum += v.y;
      sum += v.z;
      sum += v.w;
    }
    partial[(int64_t)blockIdx.x * PE_G + threadIdx.x] = sum;
  }
}

// workgroup partials added in workgroup order: eight lanes per sum split the rows, added in lane order
__global__ __launch_bounds__(PE_G * 8) void pos_enc_bwd_final_kernel(const float* __restrict__ partial, int n, float* __restrict__ dW1,
                                                                      float* __restrict__ db1, float* __restrict__ dW2,
                                                                      float* __restrict__ db2) {
  __shared__ float red[8][PE_G];
  const int i = threadIdx.x % PE_G, l = threadIdx.x / PE_G;
  float s = 0.0f;
  for (int r = l; r < n; r += 8) s += partial[(int64_t)r * PE_G + i];
  red[l][i] = s;
  __syncthreads();
  if (l != 0) return;
#pragma unroll
  for (int q = 1; q < 8; ++q) s += red[q][i];
  if (i < PE_H * 3) {
    if (dW1) dW1[i] = s;
  } else if (i < PE_H * 4) {
    if (db1) db1[i - PE_H * 3] = s;
  } else if (i < PE_H * 5) {
    if (dW2) dW2[i - PE_H * 4] = s;
  } else if (db2) {
    db2[0] = s;
  }
}

}  // namespace

extern "C" {

int wsis_edge_affinity_fwd(const float* d_q, const float* d_k, const float* d_v, const float* d_pos,
                           const int64_t* d_eu, const int64_t* d_ev, const int32_t* d_perm_u,
                           const int32_t* d_off_u, float scale, float* d_aff, float* d_res, int64_t E,
                           int64_t Su, int32_t D, void* stream) {
  WSIS_REQUIRE(E >= 0 && Su >= 0 && D >= 1, "bad sizes");
  if (Su == 0) return WSIS_OK;
  WSIS_REQUIRE(d_q && d_k && d_v && d_off_u && d_res, "null pointer");
  WSIS_REQUIRE(E == 0 || (d_pos && d_ev && d_perm_u && d_aff), "null pointer");
  (void)d_eu;
  hipLaunchKernelGGL(edge_affinity_fwd_kernel, dim3(waves_grid(Su)), dim3(256), 0, as_stream(stream), d_q,
                     d_k, d_v, d_pos, d_ev, d_perm_u, d_off_u, scale, d_aff, d_res, Su, D);
  WSIS_LAUNCH_CHECK();
  return WSIS_OK;
}

int wsis_edge_affinity_bwd(const float* d_q, const float* d_k, const float* d_v, const float* d_pos,
                           const float* d_aff, const int64_t* d_eu, const int64_t* d_ev,
                           const int32_t* d_perm_u, const int32_t* d_off_u, const int32_t* d_perm_v,
                           const int32_t* d_off_v, float scale, const float* d_daff,
                           const float* d_dres, float* d_dq, float* d_dk, float* d_dv, float* d_dpos,
                           float* d_tmp, int64_t E, int64_t S, int64_t Su, int32_t D, void* stream) {
  WSIS_REQUIRE(E >= 0 && S >= 0 && Su >= 0 && Su <= S && D >= 1, "bad sizes");
  if (S == 0) return WSIS_OK;
  WSIS_REQUIRE(d_q && d_k && d_v && d_dres && d_dq && d_dk && d_dv && d_off_u && d_off_v, "null pointer");
  WSIS_REQUIRE(E == 0 || (d_pos && d_aff && d_eu && d_ev && d_perm_u && d_perm_v && d_dpos && d_tmp),
               "null pointer");
  hipStream_t st = as_stream(stream);
  WSIS_HIP_CHECK(hipMemsetAsync(d_dq, 0, sizeof(float) * (size_t)S * D, st));
  if (Su > 0) {
    hipLaunchKernelGGL(edge_affinity_bwd_u_kernel, dim3(waves_grid(Su)), dim3(256), 0, st, d_q, d_k, d_v,
                       d_pos, d_aff, d_ev, d_perm_u, d_off_u, scale, d_daff, d_dres, d_dq, d_dpos, d_tmp, Su,
                       D);
    WSIS_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(edge_affinity_bwd_v_kernel, dim3(waves_grid(S)), dim3(256), 0, st, d_q, d_aff, d_tmp,
                     d_eu, d_perm_v, d_off_v, d_dres, d_dk, d_dv, S, Su, D);
  WSIS_LAUNCH_CHECK();
  return WSIS_OK;
}

int wsis_affinity_dense_build(const int64_t* d_eu, const int64_t* d_ev, const float* d_aff, int64_t E,
                              double* d_A, int64_t S, void* stream) {
  WSIS_REQUIRE(E >= 0 && S >= 0, "bad sizes");
  if (S == 0) return WSIS_OK;
  WSIS_REQUIRE(d_A, "null pointer");
  hipStream_t st = as_stream(stream);
  WSIS_HIP_CHECK(hipMemsetAsync(d_A, 0, sizeof(double) * (size_t)S * S, st));
  if (E == 0) return WSIS_OK;
  WSIS_REQUIRE(d_eu && d_ev && d_aff, "null pointer");
  hipLaunchKernelGGL(dense_build_kernel, dim3(grid_for(E, 256)), dim3(256), 0, st, d_eu, d_ev, d_aff, E, d_A,
                     S);
  WSIS_LAUNCH_CHECK();
  return WSIS_OK;
}

int wsis_affinity_transition(const double* d_A, const uint8_t* d_adj, const int32_t* d_pred,
                             const float* d_conf, const int32_t* d_label, int32_t cls, float thr,
                             double* d_T0, int64_t S, void* stream) {
  WSIS_REQUIRE(S >= 0, "bad size");
  if (S == 0) return WSIS_OK;
  WSIS_REQUIRE(d_A && d_adj && d_pred && d_conf && d_label && d_T0, "null pointer");
  hipLaunchKernelGGL(transition_kernel, dim3((unsigned)S), dim3(256), 0, as_stream(stream), d_A, d_adj,
                     d_pred, d_conf, d_label, cls, thr, d_T0, S);
  WSIS_LAUNCH_CHECK();
  return WSIS_OK;
}

int wsis_dgemm(const double* d_A, const double* d_B, double* d_C, int64_t M, int64_t N, int64_t Kd,
               void* stream) {
  WSIS_REQUIRE(M >= 0 && N >= 0 && Kd >= 0, "bad sizes");
  if (M == 0 || N == 0) return WSIS_OK;
  WSIS_REQUIRE(d_A && d_B && d_C, "null pointer");
  const dim3 grid((unsigned)ceil_div(N, GB), (unsigned)ceil_div(M, GB));
  hipLaunchKernelGGL(dgemm_kernel, grid, dim3(256), 0, as_stream(stream), d_A, d_B, d_C, M, N, Kd);
  WSIS_LAUNCH_CHECK();
  return WSIS_OK;
}

int wsis_affinity_colmax(const double* d_T, const int32_t* d_label, int32_t cls, double* d_scores,
                         int32_t* d_arg, int64_t S, void* stream) {
  WSIS_REQUIRE(S >= 0, "bad size");
  if (S == 0) return WSIS_OK;
  WSIS_REQUIRE(d_T && d_label && d_scores && d_arg, "null pointer");
  hipLaunchKernelGGL(colmax_kernel, dim3(grid_for(S, 64)), dim3(64), 0, as_stream(stream), d_T, d_label, cls,
                     d_scores, d_arg, S);
  WSIS_LAUNCH_CHECK();
  return WSIS_OK;
}

int64_t wsis_affinity_propagate_sparse_workspace_bytes(int64_t S, int32_t n_present) {
  if (S < 0 || n_present < 0) return -1;
  return (int64_t)(2 * (S + 2)) * 4 + (int64_t)n_present * S * 8 + S * S * 8 + 1024;
}

int wsis_affinity_propagate_sparse(const double* d_A, const uint8_t* d_adj, const int32_t* d_pred, const float* d_conf,
                                   const int32_t* d_label, const int32_t* d_cls_of, const int32_t* d_ci_of_cls,
                                   int32_t n_present, int32_t class_num, float thr, int32_t iterations, int64_t S,
                                   int32_t* d_col, double* d_val, int64_t nnz_cap, double* d_scores, int32_t* d_arg,
                                   void* d_ws, int64_t ws_bytes, void* stream) {
  WSIS_REQUIRE(S >= 0 && n_present >= 0 && iterations >= 0 && class_num >= 0 && nnz_cap >= 0, "bad sizes");
  if (S == 0 || n_present == 0) return WSIS_OK;
  WSIS_REQUIRE(d_A && d_adj && d_pred && d_conf && d_label && d_cls_of && d_ci_of_cls && d_scores && d_arg && d_ws &&
               (nnz_cap == 0 || (d_col && d_val)), "null pointer");
  WSIS_REQUIRE(ws_bytes >= wsis_affinity_propagate_sparse_workspace_bytes(S, n_present), "workspace too small");
  const size_t lds = (size_t)S * 20 + 64;
  WSIS_REQUIRE(lds <= 160 * 1024, "sparse propagation keeps two rows of S doubles in LDS: S <= 8188 (use the dense path)");
  hipStream_t st = as_stream(stream);
  char* w = static_cast<char*>(d_ws);
  int32_t* rowcnt = reinterpret_cast<int32_t*>(w);
  int32_t* rowptr = rowcnt + (S + 2);
  double* D = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(rowptr + (S + 2)) + 255) & ~(uintptr_t)255);
  double* X = D + (int64_t)n_present * S;
  const unsigned rows4 = (unsigned)ceil_div(S, 4);
  hipLaunchKernelGGL(prop_nz_count_kernel, dim3(rows4), dim3(256), 0, st, d_A, d_adj, S, rowcnt);
  WSIS_LAUNCH_CHECK();
  hipLaunchKernelGGL(prop_scan_kernel, dim3(1), dim3(1024), 0, st, rowcnt, rowptr, (int)S);
  WSIS_LAUNCH_CHECK();
  hipLaunchKernelGGL(prop_nz_fill_kernel, dim3(rows4), dim3(256), 0, st, d_A, d_adj, S, rowptr, d_col, d_val, nnz_cap);
  WSIS_LAUNCH_CHECK();
  hipLaunchKernelGGL(prop_rowsum_kernel, dim3(rows4, (unsigned)n_present), dim3(256), 0, st, rowptr, d_col, d_val, d_pred,
                     d_conf, d_label, d_cls_of, thr, S, D);
  WSIS_LAUNCH_CHECK();
  static size_t attr = 0;
  if (attr < lds) {
    WSIS_HIP_CHECK(hipFuncSetAttribute((const void*)prop_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr = lds;
  }
  hipLaunchKernelGGL(prop_rows_kernel, dim3((unsigned)S), dim3(64), lds, st, rowptr, d_col, d_val, d_pred, d_conf, d_label,
                     d_ci_of_cls, (int)class_num, thr, (int)iterations, S, D, X);
  WSIS_LAUNCH_CHECK();
  hipLaunchKernelGGL(prop_colmax_kernel, dim3((unsigned)ceil_div(S, 64), (unsigned)n_present), dim3(64), 0, st, X, d_label,
                     d_cls_of, d_scores, d_arg, S);
  WSIS_LAUNCH_CHECK();
  return WSIS_OK;
}

int64_t wsis_pos_enc_workspace_bytes(int64_t E) {
  if (E < 0) return -1;
  return ceil_div(E > 0 ? E : 1, 256) * PE_G * (int64_t)sizeof(float) + 256;
}

int wsis_pos_enc_fwd(const float* d_centre, const int64_t* d_eu, const int64_t* d_ev, const float* d_W1, const float* d_b1,
                     const float* d_W2, const float* d_b2, float* d_pos, int64_t E, void* stream) {
  WSIS_REQUIRE(E >= 0, "bad sizes");
  if (E == 0) return WSIS_OK;
  WSIS_REQUIRE(d_centre && d_eu && d_ev && d_W1 && d_b1 && d_W2 && d_b2 && d_pos, "null pointer");
  hipLaunchKernelGGL(pos_enc_fwd_kernel, dim3((unsigned)ceil_div(E, 256)), dim3(256), 0, as_stream(stream), d_centre, d_eu, d_ev,
                     d_W1, d_b1, d_W2, d_b2, d_pos, E);
  WSIS_LAUNCH_CHECK();
  return WSIS_OK;
}

int wsis_pos_enc_bwd(const float* d_centre, const int64_t* d_eu, const int64_t* d_ev, const float* d_W1, const float* d_b1,
                     const float* d_W2, const float* d_dpos, float* d_dW1, float* d_db1, float* d_dW2, float* d_db2, int64_t E,
                     void* d_ws, int64_t ws_bytes, void* stream) {
  WSIS_REQUIRE(E >= 1, "bad sizes");
  WSIS_REQUIRE(d_centre && d_eu && d_ev && d_W1 && d_b1 && d_W2 && d_dpos && d_ws, "null pointer");
  WSIS_REQUIRE(ws_bytes >= wsis_pos_enc_workspace_bytes(E), "workspace too small");
  const int n = (int)ceil_div(E, 256);
  float* partial = static_cast<float*>(d_ws);
  const size_t ldsb = (size_t)PE_G * PE_TP * sizeof(float);       // 84 KB
  static bool attr_set = false;       // (one process per GPU: include/wsis_hip.h)
  if (!attr_set) {
    WSIS_HIP_CHECK(hipFuncSetAttribute((const void*)pos_enc_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
    attr_set = true;
  }
  hipLaunchKernelGGL(pos_enc_bwd_kernel, dim3((unsigned)n), dim3(256), ldsb, as_stream(stream), d_centre, d_eu, d_ev, d_W1, d_b1,
                     d_W2, d_dpos, partial, E);
  WSIS_LAUNCH_CHECK();
  hipLaunchKernelGGL(pos_enc_bwd_final_kernel, dim3(1), dim3(PE_G * 8), 0, as_stream(stream), partial, n, d_dW1, d_db1, d_dW2,
                     d_db2);
  WSIS_LAUNCH_CHECK();
  return WSIS_OK;
}

}  // extern "C"
