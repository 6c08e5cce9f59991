// Edge-conditioned message passing of the superpoint GNN:
//   inp[s,:] = mean over edges e with src_e = s of  x[dst_e,:] @ W_e ,   W_e = weights[e] in R^{C x C}
// modules/model/spg_modules.py:97-121 (PyG NNConv, flow=target_to_source, aggr='mean') evaluated 7 times per
// forward with the same per-edge filters (spg_modules.py:168-183).  The reference runs it as index_select +
// bmm + scatter-mean (and three more launches in backward); here it is one kernel forward and one backward,
// each reading the [E,C,C] filter tensor exactly once.  One wavefront per source node (forward) / per target
// node (backward), fixed summation order => deterministic.  C <= 32 (the model uses 32).
#include "common.h"

using namespace wsis;

namespace {

constexpr int EC = 32;

// forward: wave per source node s
__global__ __launch_bounds__(256) void ecc_msg_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const int64_t* __restrict__ dst,
                                                          const int32_t* __restrict__ perm_src,
                                                          const int32_t* __restrict__ off_src, float* __restrict__ out,
                                                          int64_t S, int C) {
  const int lane = threadIdx.x & 63;
  const int o = lane & 31, h = lane >> 5;
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  for (int64_t s = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); s < S; s += nwaves) {
    const int beg = off_src[s], end = off_src[s + 1];
    float acc = 0.0f;
    for (int j = beg; j < end; ++j) {
      const int32_t e = perm_src[j];
      const float* xd = x + dst[e] * C;
      const float* we = w + (int64_t)e * C * C;
      float p = 0.0f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int ii = h * 16 + i;
        const bool ok = ii < C && o < C;
        const float xv = xd[ok ? ii : 0];
        const float wv = we[ok ? ii * C + o : 0];
        p += ok ? xv * wv : 0.0f;
      }
      p += __shfl_xor(p, 32, 64);
      acc += p;
    }
    const int cnt = end - beg;
    if (h == 0 && o < C) out[s * C + o] = acc / (float)(cnt > 0 ? cnt : 1);
  }
}

// backward: wave per target node d.  dm_e = dout[src_e]/cnt(src_e);  dW_e = x[d] (x) dm_e;  dx[d] = sum_e W_e dm_e
__global__ __launch_bounds__(256) void ecc_msg_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ dout,
                                                          const int64_t* __restrict__ src,
                                                          const int32_t* __restrict__ perm_dst,
                                                          const int32_t* __restrict__ off_dst,
                                                          const int32_t* __restrict__ off_src, float* __restrict__ dx,
                                                          float* __restrict__ dw, int64_t S, int C) {
  const int lane = threadIdx.x & 63;
  const int i = lane & 31, h = lane >> 5;   // lane owns input channel i and half h of the output channels
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  for (int64_t d = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); d < S; d += nwaves) {
    const int beg = off_dst[d], end = off_dst[d + 1];
    const float xi = (i < C) ? x[d * C + i] : 0.0f;
    float acc = 0.0f;
    for (int j = beg; j < end; ++j) {
      const int32_t e = perm_dst[j];
      const int64_t s = src[e];
      const int cnt = off_src[s + 1] - off_src[s];
      const float inv = 1.0f / (float)(cnt > 0 ? cnt : 1);
      const float* dm = dout + s * C;
      const float* we = w + (int64_t)e * C * C;
      float* dwe = dw + (int64_t)e * C * C;
      float p = 0.0f;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int oo = h * 16 + q;
        const bool ok = i < C && oo < C;
        const float g = dm[ok ? oo : 0] * inv;
        const float wv = we[ok ? i * C + oo : 0];
        if (ok) dwe[i * C + oo] = xi * g;
        p += ok ? wv * g : 0.0f;
      }
      p += __shfl_xor(p, 32, 64);
      acc += p;
    }
    if (h == 0 && i < C) dx[d * C + i] = acc;
  }
}

// backward, C == 32 (the model's width): wave per target node, lane l owns filter row i = l>>1 and the 16 output
// channels of half l&1 as four float4 -- every filter / filter-gradient matrix moves as one contiguous 4 KB
// transaction set.  Edge metadata (edge id, source, 1/cnt) of up to 64 incoming edges is fetched by 64 lanes at
// once and handed out with readlane, so the per-edge loop has no dependent index loads.
__global__ __launch_bounds__(256) void ecc_msg_bwd32_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ dout,
                                                            const int64_t* __restrict__ src,
                                                            const int32_t* __restrict__ perm_dst,
                                                            const int32_t* __restrict__ off_dst,
                                                            const int32_t* __restrict__ off_src,
                                                            float* __restrict__ dx, float* __restrict__ dw, int64_t S) {
  constexpr int C = 32;
  const int lane = threadIdx.x & 63;
  const int i = lane >> 1, h = lane & 1;
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  for (int64_t d = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); d < S; d += nwaves) {
    const int beg = off_dst[d], end = off_dst[d + 1];
    const float xi = x[d * C + i];
    float acc = 0.0f;
    for (int base = beg; base < end; base += 64) {
      const int n = min(64, end - base);
      int32_t e_l = 0, s_l = 0;
      float inv_l = 0.0f;
      if (lane < n) {
        e_l = perm_dst[base + lane];
        s_l = (int32_t)src[e_l];
        const int cnt = off_src[s_l + 1] - off_src[s_l];
        inv_l = 1.0f / (float)(cnt > 0 ? cnt : 1);
      }
      for (int j = 0; j < n; ++j) {
        const int32_t e = __builtin_amdgcn_readlane(e_l, j);
        const int32_t sn = __builtin_amdgcn_readlane(s_l, j);
        const float inv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int32_t, inv_l), j));
        const float4* dm = reinterpret_cast<const float4*>(dout + (int64_t)sn * C + h * 16);
        const float4* we = reinterpret_cast<const float4*>(w + (int64_t)e * C * C + i * C + h * 16);
        float4* dwe = reinterpret_cast<float4*>(dw + (int64_t)e * C * C + i * C + h * 16);
        float p = 0.0f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float4 g = dm[q];
          const float4 wv = we[q];
          g.x *= inv; g.y *= inv; g.z *= inv; g.w *= inv;
          dwe[q] = make_float4(xi * g.x, xi * g.y, xi * g.z, xi * g.w);
          p += wv.x * g.x;
          p += wv.y * g.y;
          p += wv.z * g.z;
          p += wv.w * g.w;
        }
        const float other = __shfl_xor(p, 1, 64);
        acc += (h == 0) ? (p + other) : (other + p);   // both halves form (low half + high half)
      }
    }
    if (h == 0) dx[d * C + i] = acc;
  }
}

int waves_grid(int64_t n) {
  int64_t g = ceil_div(n, 4);
  if (g < 1) g = 1;
  if (g > 256 * 16) g = 256 * 16;
  return (int)g;
}


// ------------------------------------------------------------------------------------------------------------------
// Edge-conditioned messages WITHOUT the per-edge filter tensor.  The filter of edge e is an affine map
// of its fnet hidden state h_e in R^64: W_e[a,b] = sum_c Wl[a*32+b, c] h_e[c] + bl[a*32+b] (graphnet.py:19-36, last
// Linear), so   m_e = x_t @ W_e = sum_c h_e[c] U_t[c,:] + U_t[64,:]   with   U_t[c,b] = sum_a x_t[a] Wl[a*32+b, c]
// (row 64: the bias term) -- a per-NODE tensor U [S, 65*32] from one small GEMM per GRU step instead of a per-EDGE
// [E, 1024] tensor read 7 times each way.  One wavefront per target node t: U_t stays in registers while the wave
// walks the in-edges of t (h_e rows through a per-wave LDS slot, broadcast reads); the mean over the out-edges of
// the source follows as the segmented mean the rest of the path already uses.
constexpr int EH = 64;           // fnet hidden width
constexpr int EU = (EH + 1) * EC;   // floats of U per node
constexpr int EBATCH = 8;        // in-edges staged per trip

// m[e,b] = sum_c h[e,c] U[t,c,b] + U[t,64,b]   for the in-edges e of t (CSR over targets)
__global__ __launch_bounds__(256) void ecc_contract_fwd_kernel(const float* __restrict__ h, const float* __restrict__ U,
                                                               const int32_t* __restrict__ perm_dst,
                                                               const int32_t* __restrict__ off_dst,
                                                               float* __restrict__ m, int64_t S) {
  __shared__ __attribute__((aligned(16))) float hs[4][EBATCH][EH];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = lane & 31, half = lane >> 5;
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  for (int64_t t = (int64_t)blockIdx.x * 4 + wave; t < S; t += nwaves) {
    const int beg = off_dst[t], end = off_dst[t + 1];
    if (beg == end) continue;
    const float* Ut = U + t * EU;
    float u[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) u[j] = Ut[(half * 32 + j) * EC + b];
    const float ub = Ut[EH * EC + b];
    for (int j0 = beg; j0 < end; j0 += EBATCH) {
      const int n = min(EBATCH, end - j0);
      int32_t eid[EBATCH];
#pragma unroll
      for (int i = 0; i < EBATCH; ++i) eid[i] = perm_dst[i < n ? j0 + i : j0];
      float hv[EBATCH];
#pragma unroll
      for (int i = 0; i < EBATCH; ++i) hv[i] = h[(int64_t)eid[i] * EH + lane];
#pragma unroll
      for (int i = 0; i < EBATCH; ++i) hs[wave][i][lane] = hv[i];
#pragma unroll
      for (int i = 0; i < EBATCH; ++i) {
        if (i < n) {       // wave-uniform
          float p = 0.0f;
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const float4 v = *reinterpret_cast<const float4*>(&hs[wave][i][half * 32 + q * 4]);
            p += v.x * u[q * 4 + 0];
            p += v.y * u[q * 4 + 1];
            p += v.z * u[q * 4 + 2];
            p += v.w * u[q * 4 + 3];
          }
          p += __shfl_xor(p, 32, 64);
          if (half == 0) m[(int64_t)eid[i] * EC + b] = p + ub;
        }
      }
    }
  }
}

// dU[t,c,b] = sum_{e in in(t)} haug[e,c] dm[e,b] (haug[e,64] = 1);  dh[e,c] = sum_b dm[e,b] U[t,c,b]
// MEAN: dm is not materialised -- dm[e,:] = d_inp[src_e,:] / out-degree(src_e), the backward of the segmented mean that
// follows the messages (the expression of segment_bwd_kernel: the same values)
template <bool MEAN>
__global__ __launch_bounds__(256) void ecc_contract_bwd_kernel(const float* __restrict__ h, const float* __restrict__ U,
                                                               const float* __restrict__ dm,
                                                               const int64_t* __restrict__ src_index,
                                                               const int32_t* __restrict__ off_src,
                                                               const int32_t* __restrict__ perm_dst,
                                                               const int32_t* __restrict__ off_dst,
                                                               float* __restrict__ dU, float* __restrict__ dh, int64_t S,
                                                               int accumulate) {
  __shared__ __attribute__((aligned(16))) float hs[4][EBATCH][EH];
  __shared__ __attribute__((aligned(16))) float ds[4][EBATCH][EC];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = lane & 31, half = lane >> 5;
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  for (int64_t t = (int64_t)blockIdx.x * 4 + wave; t < S; t += nwaves) {
    const int beg = off_dst[t], end = off_dst[t + 1];
    float* dUt = dU + t * EU;
    if (beg == end) {      // no in-edge: the node's U is unused
      for (int i = lane; i < EU; i += 64) dUt[i] = 0.0f;
      continue;
    }
    const float* Ut = U + t * EU;
    // lane c (0..63) keeps row c of U_t for the dh pass: 32 contiguous floats
    float ur[32];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float4 v = *reinterpret_cast<const float4*>(Ut + lane * EC + q * 4);
      ur[q * 4 + 0] = v.x; ur[q * 4 + 1] = v.y; ur[q * 4 + 2] = v.z; ur[q * 4 + 3] = v.w;
    }
    float du[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) du[j] = 0.0f;
    float dub = 0.0f;
    for (int j0 = beg; j0 < end; j0 += EBATCH) {
      const int n = min(EBATCH, end - j0);
      int32_t eid[EBATCH];
#pragma unroll
      for (int i = 0; i < EBATCH; ++i) eid[i] = perm_dst[i < n ? j0 + i : j0];
      float hv[EBATCH], dv[EBATCH], ov[EBATCH];
      // (the edge rows of dh that this trip accumulates into are read here, with the other loads of the trip: as
      // `*o + p` inside the per-edge loop each was a global round trip of its own behind the previous edge's store --
      // ~9 dependent round trips per node; an edge row belongs to exactly one wave of the launch)
#pragma unroll
      for (int i = 0; i < EBATCH; ++i) ov[i] = accumulate ? dh[(int64_t)eid[i] * EH + lane] : 0.0f;
#pragma unroll
      for (int i = 0; i < EBATCH; ++i) {
        hv[i] = h[(int64_t)eid[i] * EH + lane];
        if (MEAN) {
          const int64_t sn = src_index[eid[i]];
          const int cnt = off_src[sn + 1] - off_src[sn];
          dv[i] = dm[sn * EC + b] / (float)(cnt > 0 ? cnt : 1);
        } else {
          dv[i] = dm[(int64_t)eid[i] * EC + b];
        }
      }
#pragma unroll
      for (int i = 0; i < EBATCH; ++i) {
        hs[wave][i][lane] = hv[i];
        if (half == 0) ds[wave][i][b] = dv[i];
      }
#pragma unroll
      for (int i = 0; i < EBATCH; ++i) {
        if (i < n) {       // wave-uniform; edges in CSR order: a fixed order of additions
          // dU: lane (b, half) adds h[e, half*32 + j] * dm[e, b]
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const float4 v = *reinterpret_cast<const float4*>(&hs[wave][i][half * 32 + q * 4]);
            du[q * 4 + 0] += v.x * dv[i];
            du[q * 4 + 1] += v.y * dv[i];
            du[q * 4 + 2] += v.z * dv[i];
            du[q * 4 + 3] += v.w * dv[i];
          }
          dub += dv[i];
          // dh: lane c adds dm[e, :] . U_t[c, :]
          float p = 0.0f;
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const float4 v = *reinterpret_cast<const float4*>(&ds[wave][i][q * 4]);
            p += v.x * ur[q * 4 + 0];
            p += v.y * ur[q * 4 + 1];
            p += v.z * ur[q * 4 + 2];
            p += v.w * ur[q * 4 + 3];
          }
          dh[(int64_t)eid[i] * EH + lane] = accumulate ? ov[i] + p : p;     // every edge row is written by exactly one wave
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 32; ++j) dUt[(half * 32 + j) * EC + b] = du[j];
    if (half == 0) dUt[EH * EC + b] = dub;
  }
}


// ------------------------------------------------------------------------------------------------------------------
// The dense product in front of the contraction, K = 32 (hipBLASLt: 12.9 us on 2,289 rows for 19 MB written):
//   U  [S, 2080] = hx [S,32] @ W' [32, 2080]       one wave per (32-row slice, 5 of the 65 column blocks): 9.5 us
// exact fp32 (v_mfma_f32_32x32x2_f32), fixed order.  (The backward product dU @ W'^T stays on hipBLASLt: one workgroup
// per 32-row slice with the 65 k-chunks on 8 or 16 waves measured 18-30 us against 13.)
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int UB = EU / 32;      // 65 column blocks
constexpr int UC = 5;            // column blocks per wave of the forward product (65 = 13 x 5)

__device__ __forceinline__ int mrow(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

__global__ __launch_bounds__(256) void ecc_u_fwd_kernel(const float* __restrict__ hx, const float* __restrict__ W,
                                                        float* __restrict__ U, int64_t S) {
  // wave = one 32-row slice x UC consecutive column blocks: all weight loads of the wave in flight together
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r31 = lane & 31, half = lane >> 5;
  const int64_t s = (int64_t)blockIdx.x * 4 + wave, row = s * 32 + r31;
  if (s * 32 >= S) return;
  float a[16];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < S) v = *reinterpret_cast<const float4*>(hx + row * EC + half * 16 + q * 4);
    a[q * 4 + 0] = v.x; a[q * 4 + 1] = v.y; a[q * 4 + 2] = v.z; a[q * 4 + 3] = v.w;
  }
  const int cb0 = blockIdx.y * UC;
  float b[UC][16];
#pragma unroll
  for (int u = 0; u < UC; ++u)
#pragma unroll
    for (int i = 0; i < 16; ++i) b[u][i] = W[(half * 16 + i) * EU + (cb0 + u) * 32 + r31];
#pragma unroll
  for (int u = 0; u < UC; ++u) {
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[u][i], acc, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int64_t g = s * 32 + mrow(i, half);
      if (g < S) U[g * EU + (cb0 + u) * 32 + r31] = acc[i];
    }
  }
}

}  // namespace

extern "C" {

int wsis_ecc_message_fwd(const float* d_x, const float* d_w, const int64_t* d_dst, const int32_t* d_perm_src,
                         const int32_t* d_off_src, float* d_out, int64_t S, int64_t E, int32_t C, void* stream) {
  WSIS_REQUIRE(S >= 0 && E >= 0 && C >= 1 && C <= EC, "bad sizes (C <= 32)");
  if (S == 0) return WSIS_OK;
  WSIS_REQUIRE(d_x && d_off_src && d_out && (E == 0 || (d_w && d_dst && d_perm_src)), "null pointer");
  hipLaunchKernelGGL(ecc_msg_fwd_kernel, dim3(waves_grid(S)), dim3(256), 0, as_stream(stream), d_x, d_w, d_dst,
                     d_perm_src, d_off_src, d_out, S, C);
  WSIS_LAUNCH_CHECK();
  return WSIS_OK;
}

int wsis_ecc_message_bwd(const float* d_x, const float* d_w, const float* d_dout, const int64_t* d_src,
                         const int32_t* d_perm_dst, const int32_t* d_off_dst, const int32_t* d_off_src,
                         float* d_dx, float* d_dw, int64_t S, int64_t E, int32_t C, void* stream) {
  WSIS_REQUIRE(S >= 0 && E >= 0 && C >= 1 && C <= EC, "bad sizes (C <= 32)");
  if (S == 0) return WSIS_OK;
  WSIS_REQUIRE(d_x && d_dout && d_off_dst && d_off_src && d_dx && (E == 0 || (d_w && d_src && d_perm_dst && d_dw)),
               "null pointer");
  const bool fast = C == 32 && ((reinterpret_cast<uintptr_t>(d_w) | reinterpret_cast<uintptr_t>(d_dw) |
                                  reinterpret_cast<uintptr_t>(d_dout)) & 15) == 0;
  if (fast)
    hipLaunchKernelGGL(ecc_msg_bwd32_kernel, dim3(waves_grid(S)), dim3(256), 0, as_stream(stream), d_x, d_w, d_dout,
                       d_src, d_perm_dst, d_off_dst, d_off_src, d_dx, d_dw, S);
  else
    hipLaunchKernelGGL(ecc_msg_bwd_kernel, dim3(waves_grid(S)), dim3(256), 0, as_stream(stream), d_x, d_w, d_dout,
                       d_src, d_perm_dst, d_off_dst, d_off_src, d_dx, d_dw, S, C);
  WSIS_LAUNCH_CHECK();
  return WSIS_OK;
}


int wsis_ecc_contract_fwd(const float* d_h, const float* d_U, const int32_t* d_perm_dst, const int32_t* d_off_dst,
                          float* d_m, int64_t S, int64_t E, void* stream) {
  WSIS_REQUIRE(S >= 0 && E >= 0, "bad sizes");
  if (S == 0 || E == 0) return WSIS_OK;
  WSIS_REQUIRE(d_h && d_U && d_perm_dst && d_off_dst && d_m, "null pointer");
  WSIS_REQUIRE(((reinterpret_cast<uintptr_t>(d_U) | reinterpret_cast<uintptr_t>(d_h)) & 15) == 0, "16-byte alignment");
  hipLaunchKernelGGL(ecc_contract_fwd_kernel, dim3(waves_grid(S)), dim3(256), 0, as_stream(stream), d_h, d_U, d_perm_dst,
                     d_off_dst, d_m, S);
  WSIS_LAUNCH_CHECK();
  return WSIS_OK;
}

static int ecc_contract_bwd_impl(const float* d_h, const float* d_U, const float* d_dm, const int64_t* d_src_index,
                                 const int32_t* d_off_src, const int32_t* d_perm_dst, const int32_t* d_off_dst, float* d_dU,
                                 float* d_dh, int64_t S, int64_t E, int accumulate, void* stream) {
  WSIS_REQUIRE(S >= 0 && E >= 0, "bad sizes");
  if (S == 0) return WSIS_OK;
  WSIS_REQUIRE(d_U && d_off_dst && d_dU && (E == 0 || (d_h && d_dm && d_perm_dst && d_dh)), "null pointer");
  WSIS_REQUIRE(((reinterpret_cast<uintptr_t>(d_U) | reinterpret_cast<uintptr_t>(d_h)) & 15) == 0, "16-byte alignment");
  if (d_src_index)
    hipLaunchKernelGGL(ecc_contract_bwd_kernel<true>, dim3(waves_grid(S)), dim3(256), 0, as_stream(stream), d_h, d_U, d_dm,
                       d_src_index, d_off_src, d_perm_dst, d_off_dst, d_dU, d_dh, S, accumulate);
  else
    hipLaunchKernelGGL(ecc_contract_bwd_kernel<false>, dim3(waves_grid(S)), dim3(256), 0, as_stream(stream), d_h, d_U, d_dm,
                       d_src_index, d_off_src, d_perm_dst, d_off_dst, d_dU, d_dh, S, accumulate);
  WSIS_LAUNCH_CHECK();
  return WSIS_OK;
}

int wsis_ecc_contract_bwd(const float* d_h, const float* d_U, const float* d_dm, const int32_t* d_perm_dst,
                          const int32_t* d_off_dst, float* d_dU, float* d_dh, int64_t S, int64_t E, void* stream) {
  return ecc_contract_bwd_impl(d_h, d_U, d_dm, nullptr, nullptr, d_perm_dst, d_off_dst, d_dU, d_dh, S, E, 0, stream);
}

int wsis_ecc_contract_bwd_acc(const float* d_h, const float* d_U, const float* d_dm, const int32_t* d_perm_dst,
                              const int32_t* d_off_dst, float* d_dU, float* d_dh, int64_t S, int64_t E,
                              int32_t accumulate, void* stream) {
  return ecc_contract_bwd_impl(d_h, d_U, d_dm, nullptr, nullptr, d_perm_dst, d_off_dst, d_dU, d_dh, S, E, accumulate ? 1 : 0,
                               stream);
}

int wsis_ecc_contract_bwd_mean(const float* d_h, const float* d_U, const float* d_dinp, const int64_t* d_src_index,
                               const int32_t* d_off_src, const int32_t* d_perm_dst, const int32_t* d_off_dst, float* d_dU,
                               float* d_dh, int64_t S, int64_t E, int32_t accumulate, void* stream) {
  WSIS_REQUIRE(E == 0 || (d_src_index && d_off_src), "source index / offsets missing");
  if (E == 0) return ecc_contract_bwd_impl(d_h, d_U, d_dinp, nullptr, nullptr, d_perm_dst, d_off_dst, d_dU, d_dh, S, E, 0, stream);
  return ecc_contract_bwd_impl(d_h, d_U, d_dinp, d_src_index, d_off_src, d_perm_dst, d_off_dst, d_dU, d_dh, S, E,
                               accumulate ? 1 : 0, stream);
}

int wsis_ecc_u_fwd(const float* d_hx, const float* d_W, float* d_U, int64_t S, void* stream) {
  WSIS_REQUIRE(S >= 0, "bad sizes");
  if (S == 0) return WSIS_OK;
  WSIS_REQUIRE(d_hx && d_W && d_U, "null pointer");
  WSIS_REQUIRE((reinterpret_cast<uintptr_t>(d_hx) & 15) == 0, "16-byte alignment");
  hipLaunchKernelGGL(ecc_u_fwd_kernel, dim3((unsigned)ceil_div(S, 128), UB / UC), dim3(256), 0, as_stream(stream), d_hx, d_W, d_U,
                     S);
  WSIS_LAUNCH_CHECK();
  return WSIS_OK;
}

}  // extern "C"
