// Sparse convolution on the gather-table rulebook.
//
//   out[r,:] = sum_k X[nbr[k][r],:] @ W[k]          (wsis_spconv_fwd; also every dIn pass)
//   dW[k]    = sum_r X[nbr[k][r],:]^T (x) dY[r,:]   (wsis_spconv_dw)
//
// Output-stationary implicit GEMM.  A workgroup owns a tile of 128 output rows *in tile order* (rows
// sorted by their set of active kernel offsets, wsis_mask_order), reads the tile's slice of the packed
// gather table once, derives which offsets the tile (and each 32-row wave slice) uses, and walks only
// those.  Per (offset, 32-channel chunk) step the gathered input rows and the weight chunk are fetched
// into registers while the previous step's MFMAs run (issue-early / write-late staging), written to LDS,
// and consumed by the exact-fp32 matrix instruction v_mfma_f32_32x32x2_f32.  No per-offset gather
// buffer in HBM and no scatter-add atomics (upstream spconv does both); the reduction order is fixed,
// so results are run-to-run deterministic and independent of the tile order.
#include "common.h"

using namespace wsis;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int TM = 128;       // output rows per workgroup
constexpr int CK = 32;        // input channels staged per step
constexpr int A_STRIDE = 36;  // words; 16-B aligned rows, conflict-free ds_read_b128
constexpr int KG = 32;        // kernel offsets handled per group (bit masks are 32 bit)

// ------------------------------------------------------------------------------------------
// forward / dIn kernel
// ------------------------------------------------------------------------------------------
template <int NB>
__global__ __launch_bounds__(256) void spconv_fwd_kernel(
    const float* __restrict__ X, const int32_t* __restrict__ nbrS, const int32_t* __restrict__ order,
    const float* __restrict__ W, const float* __restrict__ bias, const float* __restrict__ residual,
    float* __restrict__ out, float* __restrict__ partial, int64_t M_in, int64_t M_out, int K, int Cin, int Cout,
    int k_per) {
  __shared__ __attribute__((aligned(16))) float As[TM * A_STRIDE];
  __shared__ __attribute__((aligned(16))) float Bs[CK * NB * 32];
  __shared__ int32_t nbT[KG * TM];   // [offset in group][tile row] -> input row or -1
  __shared__ int32_t rowId[TM];
  __shared__ uint32_t grpMask[4];    // per 32-row slice: bit k set iff some row of the slice uses offset k

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int half = lane >> 5;
  const int l31 = lane & 31;
  const int64_t tile0 = (int64_t)blockIdx.x * TM;
  const int col0 = blockIdx.y * (NB * 32);
  const int ncols = min(NB * 32, Cout - col0);
  const bool vec4 = (Cin & 3) == 0;
  const bool wvec4 = (Cout & 3) == 0;

  int32_t my_row = -1;
  if (tid < TM) {
    const int64_t t = tile0 + tid;
    if (t < M_out) my_row = order ? order[t] : (int32_t)t;
    rowId[tid] = my_row;
  }

  f32x16 acc[NB];
#pragma unroll
  for (int cb = 0; cb < NB; ++cb)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[cb][i] = 0.0f;

  // kernel-offset split: blockIdx.z owns offsets [k_begin, k_end); with gridDim.z > 1 the raw sums go to a
  // partial slab that spconv_reduce_kernel adds in a fixed order (small levels have too few tiles to fill
  // 256 CUs otherwise).
  const int k_begin = blockIdx.z * k_per;
  const int k_end = min(K, k_begin + k_per);

  for (int kg0 = k_begin; kg0 < k_end; kg0 += KG) {
    const int kcount = min(KG, k_end - kg0);
    __syncthreads();  // previous group's nbT / grpMask readers are done
    if (tid < 4) grpMask[tid] = 0u;
    __syncthreads();
    // ---- tile slice of the packed gather table: nbrS[k][tile0 + t], coalesced 512-B pieces
    if (tid < TM) {
      uint32_t bits_lo = 0u, bits_hi = 0u;  // which offsets are active in my 32-row slice (lo/hi half of the wave)
      for (int kk = 0; kk < kcount; ++kk) {
        int32_t v = -1;
        if (my_row >= 0) v = nbrS ? nbrS[(int64_t)(kg0 + kk) * M_out + tile0 + tid] : my_row;
        nbT[kk * TM + tid] = v;
        const unsigned long long b = __ballot(v >= 0);
        if ((uint32_t)b) bits_lo |= 1u << kk;
        if ((uint32_t)(b >> 32)) bits_hi |= 1u << kk;
      }
      if (lane == 0) {
        grpMask[wave * 2] = bits_lo;
        grpMask[wave * 2 + 1] = bits_hi;
      }
    }
    __syncthreads();
    const uint32_t my_mask = grpMask[wave];
    uint32_t tile_mask = grpMask[0] | grpMask[1] | grpMask[2] | grpMask[3];
    if (tile_mask == 0u) continue;

    // ---- pipelined walk over (active offset, channel chunk) steps
    f32x4 ra[4];
    f32x4 rb[NB];
    auto fetch = [&](int kk, int ci0) {
      const int cin_here = min(CK, Cin - ci0);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int f = tid + 256 * j;
        const int row = f >> 3;
        const int c4 = (f & 7) * 4;
        const int32_t g = nbT[kk * TM + row];
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (g >= 0) {
          const float* src = X + (int64_t)g * Cin + ci0 + c4;
          if (vec4) {
            if (c4 < cin_here) v = *reinterpret_cast<const f32x4*>(src);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (c4 + e < cin_here) v[e] = src[e];
          }
        }
        ra[j] = v;
      }
      const float* Wk = W + (int64_t)(kg0 + kk) * Cin * Cout;
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int f = tid + 256 * j;
        const int kr = f / (NB * 8);
        const int c4 = (f - kr * (NB * 8)) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (kr < cin_here) {
          const float* src = Wk + (int64_t)(ci0 + kr) * Cout + col0 + c4;
          if (wvec4 && c4 + 3 < ncols) {
            v = *reinterpret_cast<const f32x4*>(src);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (c4 + e < ncols) v[e] = src[e];
          }
        }
        rb[j] = v;
      }
    };

    int kk = __builtin_ctz(tile_mask);
    tile_mask &= tile_mask - 1;
    int ci0 = 0;
    fetch(kk, ci0);
    for (;;) {
      // write the fetched step to LDS
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int f = tid + 256 * j;
        *reinterpret_cast<f32x4*>(&As[(f >> 3) * A_STRIDE + (f & 7) * 4]) = ra[j];
      }
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int f = tid + 256 * j;
        const int kr = f / (NB * 8);
        *reinterpret_cast<f32x4*>(&Bs[kr * (NB * 32) + (f - kr * (NB * 8)) * 4]) = rb[j];
      }
      __syncthreads();
      // next step (issue its loads now, they land while the MFMAs below run)
      const int cur_kk = kk;
      const int cin_here = min(CK, Cin - ci0);
      int nkk = kk, nci = ci0 + CK;
      bool more = true;
      if (nci >= Cin) {
        nci = 0;
        if (tile_mask) {
          nkk = __builtin_ctz(tile_mask);
          tile_mask &= tile_mask - 1;
        } else {
          more = false;
        }
      }
      if (more) fetch(nkk, nci);
      if ((my_mask >> cur_kk) & 1u) {
        // MFMA k index (step s, half h) <-> staged channel h*16 + s, so a lane reads 16 contiguous
        // floats of its A row with four ds_read_b128.
        const float* arow = &As[(wave * 32 + l31) * A_STRIDE + half * 16];
        float a[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 t = *reinterpret_cast<const f32x4*>(arow + 4 * q);
          a[4 * q + 0] = t[0];
          a[4 * q + 1] = t[1];
          a[4 * q + 2] = t[2];
          a[4 * q + 3] = t[3];
        }
        const int nsteps = min(16, cin_here);  // channels >= cin_here are zero in A and B
        const float* bcol = &Bs[(half * 16) * (NB * 32) + l31];
#pragma unroll
        for (int s = 0; s < 16; ++s) {
          if (s < nsteps) {
#pragma unroll
            for (int cb = 0; cb < NB; ++cb) {
              const float b = bcol[s * (NB * 32) + cb * 32];
              acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b, acc[cb], 0, 0, 0);
            }
          }
        }
      }
      if (!more) break;
      __syncthreads();
      kk = nkk;
      ci0 = nci;
    }
  }

  // ---- epilogue: C/D map of the 32x32 tile: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*half
  if (gridDim.z > 1) {
    float* dst = partial + (int64_t)blockIdx.z * M_out * Cout;
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) {
      const int c = cb * 32 + l31;
      if (c >= ncols) continue;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int rr = (reg & 3) + 8 * (reg >> 2) + 4 * half;
        const int32_t r = rowId[wave * 32 + rr];
        if (r >= 0) dst[(int64_t)r * Cout + col0 + c] = acc[cb][reg];
      }
    }
    return;
  }
#pragma unroll
  for (int cb = 0; cb < NB; ++cb) {
    const int c = cb * 32 + l31;
    if (c >= ncols) continue;
    const float bv = bias ? bias[col0 + c] : 0.0f;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int rr = (reg & 3) + 8 * (reg >> 2) + 4 * half;
      const int32_t r = rowId[wave * 32 + rr];
      if (r < 0) continue;
      const int64_t o = (int64_t)r * Cout + col0 + c;
      float v = acc[cb][reg] + bv;
      if (residual) v += residual[o];
      out[o] = v;
    }
  }
}

__global__ void spconv_reduce_kernel(const float* __restrict__ partial, const float* __restrict__ bias,
                                     const float* __restrict__ residual, float* __restrict__ out, int64_t M_out,
                                     int Cout, int ksplit) {
  const int64_t total = M_out * Cout;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    float s = 0.0f;
    for (int z = 0; z < ksplit; ++z) s += partial[(int64_t)z * total + t];
    if (bias) s += bias[t % Cout];
    if (residual) s += residual[t];
    out[t] = s;
  }
}

int fwd_ksplit(int64_t M_out, int K, int Cout) {
  const int nblk = (Cout + 31) / 32;
  const int NB = nblk <= 5 ? nblk : 4;
  const int64_t wgs = ceil_div(M_out, TM) * ceil_div(Cout, NB * 32);
  if (wgs >= 512 || K == 1) return 1;
  int64_t ks = ceil_div(1024, wgs);
  if (ks > K) ks = K;
  return (int)ks;
}

__global__ void weight_transpose_kernel(const float* __restrict__ W, float* __restrict__ WT, int K,
                                        int Cin, int Cout, int flip) {
  const int64_t total = (int64_t)K * Cin * Cout;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    // t indexes WT [K, Cout, Cin]
    const int ci = (int)(t % Cin);
    const int64_t u = t / Cin;
    const int co = (int)(u % Cout);
    const int kt = (int)(u / Cout);
    const int ks = flip ? (K - 1 - kt) : kt;
    WT[t] = W[((int64_t)ks * Cin + ci) * Cout + co];
  }
}

// nbrS[k][t] = nbr[k][order[t]]  (gather table with its columns in tile order)
__global__ void rulebook_pack_kernel(const int32_t* __restrict__ nbr, const int32_t* __restrict__ order,
                                     int32_t* __restrict__ nbrS, int64_t M, int K) {
  const int64_t total = M * K;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t k = t / M;
    const int64_t c = t - k * M;
    nbrS[t] = nbr[k * M + order[c]];
  }
}

// ------------------------------------------------------------------------------------------
// dW kernel: grid (chunk, k, ci-block).  Each wave streams the rows of its chunk in tile order; the MFMA
// A operand is X^T (lane = input channel, the two k-slots = two consecutive tile rows), B is dY -- both
// are coalesced 128-B row segments loaded straight from global memory, no LDS transposition.  Row pairs
// whose offset k is inactive are skipped (tile order groups equal offset sets, so most steps are either
// fully active or fully skipped).
// ------------------------------------------------------------------------------------------
template <int NBO>
__global__ __launch_bounds__(256) void spconv_dw_kernel(
    const float* __restrict__ X, const int32_t* __restrict__ nbrS, const int32_t* __restrict__ order,
    const float* __restrict__ dY, float* __restrict__ partial, int64_t M_in, int64_t M_out, int K, int Cin,
    int Cout, int rows_per_chunk, int n_cib, int co0) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // [4][NBO][1024]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int half = lane >> 5;
  const int l31 = lane & 31;
  const int chunk = blockIdx.x;
  const int k = blockIdx.y;
  const int cib = blockIdx.z;
  const int ci = cib * 32 + l31;
  const bool ci_ok = ci < Cin;

  const int64_t row_begin = (int64_t)chunk * rows_per_chunk;
  const int64_t row_end = min(M_out, row_begin + rows_per_chunk);
  // each wave takes a contiguous quarter (multiple of 2 rows)
  const int64_t span = row_end - row_begin;
  int64_t per_wave = ((span + 3) / 4 + 1) & ~(int64_t)1;
  const int64_t w_begin = row_begin + wave * per_wave;
  const int64_t w_end = min(row_end, w_begin + per_wave);

  f32x16 acc[NBO];
#pragma unroll
  for (int cb = 0; cb < NBO; ++cb)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[cb][i] = 0.0f;

  const int32_t* nbk = nbrS ? nbrS + (int64_t)k * M_out : nullptr;
  bool cob_ok[NBO];
#pragma unroll
  for (int cb = 0; cb < NBO; ++cb) cob_ok[cb] = (co0 + cb * 32 + l31) < Cout;

  constexpr int U = 4;  // steps (pairs of rows) in flight
  for (int64_t r0 = w_begin; r0 < w_end; r0 += 2 * U) {
    int32_t g[U];
    int64_t yr[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t t = r0 + 2 * u + half;
      g[u] = -1;
      yr[u] = 0;
      if (t < w_end) {
        yr[u] = order ? order[t] : t;
        g[u] = nbk ? nbk[t] : (int32_t)yr[u];
      }
    }
    float a[U];
    float b[U][NBO];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      a[u] = 0.0f;
      if (g[u] >= 0 && ci_ok) a[u] = X[(int64_t)g[u] * Cin + ci];
#pragma unroll
      for (int cb = 0; cb < NBO; ++cb) {
        b[u][cb] = 0.0f;
        if (g[u] >= 0 && cob_ok[cb]) b[u][cb] = dY[yr[u] * Cout + co0 + cb * 32 + l31];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (__ballot(g[u] >= 0) == 0ull) continue;  // both rows of the step have no pair
#pragma unroll
      for (int cb = 0; cb < NBO; ++cb)
        acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u][cb], acc[cb], 0, 0, 0);
    }
  }

  // cross-wave reduction in a fixed order, then one plain store of the partial slab
#pragma unroll
  for (int cb = 0; cb < NBO; ++cb)
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int row = (reg & 3) + 8 * (reg >> 2) + 4 * half;  // ci within block
      red[(wave * NBO + cb) * 1024 + row * 32 + l31] = acc[cb][reg];
    }
  __syncthreads();
  const int ci_pad = n_cib * 32;
  float* dst = partial + (((int64_t)chunk * K + k) * ci_pad + cib * 32) * (int64_t)Cout;
  for (int f = tid; f < NBO * 1024; f += 256) {
    const int cb = f >> 10;
    const int row = (f >> 5) & 31;
    const int c = f & 31;
    const int co = co0 + cb * 32 + c;
    if (co >= Cout) continue;
    const float v = ((red[(0 * NBO + cb) * 1024 + row * 32 + c] + red[(1 * NBO + cb) * 1024 + row * 32 + c]) +
                     red[(2 * NBO + cb) * 1024 + row * 32 + c]) +
                    red[(3 * NBO + cb) * 1024 + row * 32 + c];
    dst[(int64_t)row * Cout + co] = v;
  }
}

__global__ void dw_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dW, int n_chunks,
                                 int K, int Cin, int Cout, int ci_pad) {
  const int64_t total = (int64_t)K * Cin * Cout;
  const int64_t slab = (int64_t)K * ci_pad * Cout;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int co = (int)(t % Cout);
    const int64_t u = t / Cout;
    const int ci = (int)(u % Cin);
    const int k = (int)(u / Cin);
    const int64_t off = ((int64_t)k * ci_pad + ci) * Cout + co;
    float s = 0.0f;
    for (int c = 0; c < n_chunks; ++c) s += partial[c * slab + off];
    dW[t] = s;
  }
}

int dw_rows_per_chunk(int64_t M_out) {
  int64_t rpc = ceil_div(M_out, 64);
  if (rpc < 512) rpc = 512;
  rpc = (rpc + 7) & ~(int64_t)7;
  return (int)rpc;
}

}  // namespace

extern "C" {

int wsis_rulebook_pack(const int32_t* d_nbr, const int32_t* d_order, int32_t* d_nbr_packed, int64_t M,
                       int32_t K, void* stream) {
  WSIS_REQUIRE(M >= 0 && K >= 1, "bad sizes");
  if (M == 0) return WSIS_OK;
  WSIS_REQUIRE(d_nbr && d_order && d_nbr_packed, "null pointer");
  hipLaunchKernelGGL(rulebook_pack_kernel, dim3(grid_for(M * K, 256)), dim3(256), 0, as_stream(stream), d_nbr,
                     d_order, d_nbr_packed, M, K);
  WSIS_LAUNCH_CHECK();
  return WSIS_OK;
}

int64_t wsis_spconv_fwd_workspace_bytes(int64_t M_out, int32_t K, int32_t Cin, int32_t Cout) {
  if (M_out < 0 || K < 1 || Cin < 1 || Cout < 1) return -1;
  const int ks = fwd_ksplit(M_out, K, Cout);
  return ks <= 1 ? 256 : (int64_t)ks * M_out * Cout * (int64_t)sizeof(float) + 256;
}

int wsis_spconv_fwd(const float* d_X, const int32_t* d_nbr, const int32_t* d_order, const float* d_W,
                    const float* d_bias, const float* d_residual, float* d_out, int64_t M_in,
                    int64_t M_out, int32_t K, int32_t Cin, int32_t Cout, void* d_ws, int64_t ws_bytes,
                    void* stream) {
  WSIS_REQUIRE(M_in >= 0 && M_out >= 0 && K >= 1 && Cin >= 1 && Cout >= 1, "bad sizes");
  if (M_out == 0) return WSIS_OK;
  WSIS_REQUIRE(d_X && d_W && d_out, "null pointer");
  WSIS_REQUIRE(d_nbr || (K == 1 && M_in == M_out), "nbr may be null only for the dense 1x1 case");
  WSIS_REQUIRE(M_out < (int64_t)1 << 31 && M_in < (int64_t)1 << 31, "row count exceeds int32");
  const int nblk = (Cout + 31) / 32;
  const int NB = nblk <= 5 ? nblk : 4;
  const int ksplit = fwd_ksplit(M_out, K, Cout);
  const int k_per = (K + ksplit - 1) / ksplit;
  const int kz = (K + k_per - 1) / k_per;   // blocks along z that own at least one offset
  float* partial = nullptr;
  if (kz > 1) {
    WSIS_REQUIRE(d_ws && ws_bytes >= (int64_t)kz * M_out * Cout * (int64_t)sizeof(float), "workspace too small");
    partial = static_cast<float*>(d_ws);
  }
  const dim3 grid((unsigned)ceil_div(M_out, TM), (unsigned)ceil_div(Cout, NB * 32), (unsigned)kz);
  hipStream_t st = as_stream(stream);
#define WSIS_FWD_CASE(n)                                                                          \
  case n:                                                                                         \
    hipLaunchKernelGGL(spconv_fwd_kernel<n>, grid, dim3(256), 0, st, d_X, d_nbr, d_order, d_W,    \
                       d_bias, d_residual, d_out, partial, M_in, M_out, K, Cin, Cout, k_per);     \
    break;
  switch (NB) {
    WSIS_FWD_CASE(1)
    WSIS_FWD_CASE(2)
    WSIS_FWD_CASE(3)
    WSIS_FWD_CASE(4)
    WSIS_FWD_CASE(5)
    default:
      return fail(WSIS_ERR_ARG, "spconv_fwd: unsupported NB");
  }
#undef WSIS_FWD_CASE
  WSIS_LAUNCH_CHECK();
  if (kz > 1) {
    hipLaunchKernelGGL(spconv_reduce_kernel, dim3(grid_for(M_out * Cout, 256)), dim3(256), 0, st, partial, d_bias,
                       d_residual, d_out, M_out, Cout, kz);
    WSIS_LAUNCH_CHECK();
  }
  return WSIS_OK;
}

int wsis_weight_transpose(const float* d_W, float* d_WT, int32_t K, int32_t Cin, int32_t Cout,
                          int32_t flip, void* stream) {
  WSIS_REQUIRE(K >= 1 && Cin >= 1 && Cout >= 1 && d_W && d_WT, "bad args");
  const int64_t total = (int64_t)K * Cin * Cout;
  hipLaunchKernelGGL(weight_transpose_kernel, dim3(grid_for(total, 256)), dim3(256), 0, as_stream(stream),
                     d_W, d_WT, K, Cin, Cout, flip);
  WSIS_LAUNCH_CHECK();
  return WSIS_OK;
}

int64_t wsis_spconv_dw_workspace_bytes(int64_t M_out, int32_t K, int32_t Cin, int32_t Cout) {
  if (M_out < 0 || K < 1 || Cin < 1 || Cout < 1) return -1;
  const int rpc = dw_rows_per_chunk(M_out);
  const int64_t n_chunks = M_out == 0 ? 1 : ceil_div(M_out, rpc);
  const int64_t ci_pad = ceil_div(Cin, 32) * 32;
  return n_chunks * K * ci_pad * Cout * (int64_t)sizeof(float) + 256;
}

int wsis_spconv_dw(const float* d_X, const int32_t* d_nbr, const int32_t* d_order, const float* d_dY,
                   float* d_dW, int64_t M_in, int64_t M_out, int32_t K, int32_t Cin, int32_t Cout, void* d_ws,
                   int64_t ws_bytes, void* stream) {
  WSIS_REQUIRE(M_in >= 0 && M_out >= 0 && K >= 1 && Cin >= 1 && Cout >= 1 && d_dW, "bad args");
  hipStream_t st = as_stream(stream);
  if (M_out == 0) {
    WSIS_HIP_CHECK(hipMemsetAsync(d_dW, 0, sizeof(float) * (size_t)K * Cin * Cout, st));
    return WSIS_OK;
  }
  WSIS_REQUIRE(d_X && d_dY && d_ws, "null pointer");
  WSIS_REQUIRE(d_nbr || (K == 1 && M_in == M_out), "nbr may be null only for the dense 1x1 case");
  WSIS_REQUIRE(ws_bytes >= wsis_spconv_dw_workspace_bytes(M_out, K, Cin, Cout), "workspace too small");
  const int rpc = dw_rows_per_chunk(M_out);
  const int n_chunks = (int)ceil_div(M_out, rpc);
  const int n_cib = (int)ceil_div(Cin, 32);
  const int ci_pad = n_cib * 32;
  float* partial = static_cast<float*>(d_ws);
  const int nblk = (Cout + 31) / 32;
  const dim3 grid((unsigned)n_chunks, (unsigned)K, (unsigned)n_cib);
  // output-channel blocks are processed in groups of <= 5 per launch (register budget)
  for (int cb0 = 0; cb0 < nblk; cb0 += 5) {
    const int nbo = min(5, nblk - cb0);
    const int co0 = cb0 * 32;
    const size_t lds = (size_t)4 * nbo * 1024 * sizeof(float);
#define WSIS_DW_CASE(n)                                                                                  \
  case n:                                                                                                \
    WSIS_HIP_CHECK(hipFuncSetAttribute((const void*)spconv_dw_kernel<n>,                                 \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));           \
    hipLaunchKernelGGL(spconv_dw_kernel<n>, grid, dim3(256), lds, st, d_X, d_nbr, d_order, d_dY, partial, \
                       M_in, M_out, K, Cin, Cout, rpc, n_cib, co0);                                      \
    break;
    switch (nbo) {
      WSIS_DW_CASE(1)
      WSIS_DW_CASE(2)
      WSIS_DW_CASE(3)
      WSIS_DW_CASE(4)
      WSIS_DW_CASE(5)
    }
#undef WSIS_DW_CASE
    WSIS_LAUNCH_CHECK();
  }
  const int64_t total = (int64_t)K * Cin * Cout;
  hipLaunchKernelGGL(dw_reduce_kernel, dim3(grid_for(total, 256)), dim3(256), 0, st, partial, d_dW, n_chunks,
                     K, Cin, Cout, ci_pad);
  WSIS_LAUNCH_CHECK();
  return WSIS_OK;
}

}  // extern "C"
