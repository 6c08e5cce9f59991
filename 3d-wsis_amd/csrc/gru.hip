// GRUCellEx of the superpoint GNN: GRU cell with an input gate and per-row normalisation of the
// gate pre-activations (modules/model/spg_modules.py:207-253), evaluated 7 times per forward.  The reference
// runs ~30 elementwise / small-GEMM launches per evaluation (and ~60 in backward) on [S,32]..[S,96] tensors that
// are pure launch latency; here one kernel does the forward and one the backward (+ a fixed-order reduce of the
// parameter gradients).  One wavefront per row, weights (and their transposes) resident in LDS, C = 32.
//
//   xin = sigmoid(Wig h + big) * x
//   gi = rownorm(Wih xin), gh = rownorm(Whh h)              rownorm(v) = (v - mean)/sqrt(var + 1e-5) over 3C
//   r = sigmoid(gi_r + bih_r + gh_r + bhh_r), z = sigmoid(gi_z + bih_z + gh_z + bhh_z)
//   n = tanh(gi_n + bih_n + r * (gh_n + bhh_n)),  hy = n + z * (h - n)
#include <algorithm>
#include <cstdlib>

#include "common.h"

using namespace wsis;

namespace {

constexpr int GC = 32;         // channels
constexpr int G3 = 96;         // 3 * channels
constexpr int GRU_WAVES = 4;       // backward (8 waves per workgroup: 29.9 -> 54 us on 2,289 rows)
constexpr int GRU_WF = 8;          // forward: one row per wave, the weights staged once per 8 rows (16.2 -> 12.1 us)
constexpr int GRU_P = 2 * G3 * GC + GC * GC + 2 * G3 + GC;   // parameter-gradient floats: Wih, Whh, Wig, bih, bhh, big

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ float sigm(float v) { return 1.0f / (1.0f + expf(-v)); }

struct GruLds {
  float WigT[GC][GC + 1];    // [j][c]   (rows padded by one float: the transposed stores below hit 32 different banks)
  float WihT[GC][G3 + 1];    // [j][o]
  float WhhT[GC][G3 + 1];
  float Wih[G3][GC];     // [o][j]   (backward only)
  float Whh[G3][GC];
  float Wig[GC][GC];     // [c][j]
  float row[GRU_WF][4][G3];      // per-wave scratch rows (GRU_WF >= GRU_WAVES)
};

// the three weight matrices -> LDS (transposed, and as they are for the backward).  Every global load of the workgroup is
// issued before the first LDS store: one memory latency for the whole staging (a load -> store loop ran 16 dependent
// round trips, 13 of the backward kernel's 28 us on 2,289 rows)
template <int NT>
__device__ __forceinline__ void gru_load_weights(GruLds& L, const float* Wig, const float* Wih, const float* Whh,
                                                 bool need_plain) {
  constexpr int N_IH = (G3 * GC / 4 + NT - 1) / NT;      // float4 pieces per thread of Wih / Whh
  constexpr int N_IG = (GC * GC / 4 + NT - 1) / NT;
  float4 a[N_IH], b[N_IH], g[N_IG];
#pragma unroll
  for (int u = 0; u < N_IH; ++u) {
    const int f4 = threadIdx.x + u * NT;
    const bool ok = f4 < G3 * GC / 4;
    a[u] = ok ? reinterpret_cast<const float4*>(Wih)[f4] : make_float4(0.f, 0.f, 0.f, 0.f);
    b[u] = ok ? reinterpret_cast<const float4*>(Whh)[f4] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int u = 0; u < N_IG; ++u) {
    const int f4 = threadIdx.x + u * NT;
    g[u] = f4 < GC * GC / 4 ? reinterpret_cast<const float4*>(Wig)[f4] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int u = 0; u < N_IH; ++u) {
    const int f4 = threadIdx.x + u * NT;
    if (f4 < G3 * GC / 4) {
      const int o = (f4 * 4) / GC, j = (f4 * 4) % GC;
      const float av[4] = {a[u].x, a[u].y, a[u].z, a[u].w}, bv[4] = {b[u].x, b[u].y, b[u].z, b[u].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        L.WihT[j + e][o] = av[e];
        L.WhhT[j + e][o] = bv[e];
      }
      if (need_plain) {
        *reinterpret_cast<float4*>(&L.Wih[o][j]) = a[u];
        *reinterpret_cast<float4*>(&L.Whh[o][j]) = b[u];
      }
    }
  }
#pragma unroll
  for (int u = 0; u < N_IG; ++u) {
    const int f4 = threadIdx.x + u * NT;
    if (f4 < GC * GC / 4) {
      const int c = (f4 * 4) / GC, j = (f4 * 4) % GC;
      const float gv[4] = {g[u].x, g[u].y, g[u].z, g[u].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) L.WigT[j + e][c] = gv[e];
      if (need_plain) *reinterpret_cast<float4*>(&L.Wig[c][j]) = g[u];
    }
  }
}

// forward of one row by one wave; returns the intermediates needed by the backward in registers
struct GruRow {
  float x, h;          // lanes 0..31 (duplicated in 32..63)
  float gin;           // input gate, lane c
  float xin;           // lane c
  float gi1, gi2;      // normalised gi[o=lane], gi[o=lane+64] (second valid for lane<32)
  float gh1, gh2;
  float sig_i, sig_h;  // 1/sqrt(var+eps) of the two row norms
  float r, z, n, hy;   // lanes 0..31
};

__device__ __forceinline__ GruRow gru_row_fwd(GruLds& L, int wave, int lane, float xval,
                                              const float* __restrict__ h, const float* __restrict__ big,
                                              const float* __restrict__ bih, const float* __restrict__ bhh,
                                              int64_t row) {
  GruRow R;
  const int c = lane & 31;
  R.x = xval;
  R.h = h[row * GC + c];
  float* hrow = L.row[wave][0];
  float* xrow = L.row[wave][1];
  if (lane < 32) hrow[c] = R.h;
  // input gate
  float a = big[c];
#pragma unroll
  for (int j = 0; j < GC; ++j) a += L.WigT[j][c] * hrow[j];
  R.gin = sigm(a);
  R.xin = R.gin * R.x;
  if (lane < 32) xrow[c] = R.xin;
  // gi = Wih xin, gh = Whh h   (o = lane, and o = lane + 64 for lane < 32)
  float gi1 = 0.f, gi2 = 0.f, gh1 = 0.f, gh2 = 0.f;
  const int o2 = (lane < 32) ? lane + 64 : lane;   // lanes >= 32 recompute o = lane (ignored)
#pragma unroll
  for (int j = 0; j < GC; ++j) {
    const float xv = xrow[j], hv = hrow[j];
    gi1 += L.WihT[j][lane] * xv;
    gh1 += L.WhhT[j][lane] * hv;
    gi2 += L.WihT[j][o2] * xv;
    gh2 += L.WhhT[j][o2] * hv;
  }
  const bool two = lane < 32;
  // row norms over the 96 values
  const float inv96 = 1.0f / (float)G3;
  const float mi = wsum(gi1 + (two ? gi2 : 0.f)) * inv96;
  const float mh = wsum(gh1 + (two ? gh2 : 0.f)) * inv96;
  const float di1 = gi1 - mi, di2 = gi2 - mi, dh1 = gh1 - mh, dh2 = gh2 - mh;
  const float vi = wsum(di1 * di1 + (two ? di2 * di2 : 0.f)) * inv96;
  const float vh = wsum(dh1 * dh1 + (two ? dh2 * dh2 : 0.f)) * inv96;
  R.sig_i = 1.0f / sqrtf(vi + 1e-5f);
  R.sig_h = 1.0f / sqrtf(vh + 1e-5f);
  R.gi1 = di1 * R.sig_i;
  R.gi2 = di2 * R.sig_i;
  R.gh1 = dh1 * R.sig_h;
  R.gh2 = dh2 * R.sig_h;
  // gates on lanes 0..31: r from o=c (own first), z from o=32+c (lane 32+c's first), n from o=64+c (own second)
  const float gi_z = __shfl(R.gi1, 32 + c, 64), gh_z = __shfl(R.gh1, 32 + c, 64);
  R.r = sigm(R.gi1 + bih[c] + R.gh1 + bhh[c]);
  R.z = sigm(gi_z + bih[32 + c] + gh_z + bhh[32 + c]);
  R.n = tanhf(R.gi2 + bih[64 + c] + R.r * (R.gh2 + bhh[64 + c]));
  R.hy = R.n + R.z * (R.h - R.n);
  return R;
}

// mean over the messages of row `row`'s segment, lane (grp = lane >> 5, c): rows beg + grp, beg + grp + 2, ... in
// ascending order, then the two groups added, then / count -- the arithmetic and order of segment_reduce_kernel<1>
// (segment.hip) at C = 32, so the fused launch gives the values of the two it replaces
__device__ __forceinline__ float gru_seg_mean(const float* __restrict__ m, const int32_t* __restrict__ perm,
                                              const int32_t* __restrict__ off, int64_t row, int lane) {
  const int c = lane & 31, grp = lane >> 5;
  const int beg = off[row], end = off[row + 1];
  float acc = 0.0f;
  constexpr int U = 8;
  int j = beg + grp;
  for (; j + (U - 1) * 2 < end; j += U * 2) {
    int32_t p[U];
    float v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) p[u] = perm[j + u * 2];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = m[(int64_t)p[u] * GC + c];
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u];
  }
  for (; j < end; j += 2) acc += m[(int64_t)perm[j] * GC + c];
  acc += __shfl_xor(acc, 32, 64);
  const int cnt = end - beg;
  return acc / (float)(cnt > 0 ? cnt : 1);
}

// MEAN: the cell's input is the segmented mean of the edge messages m [E,32] (CSR perm / off over the rows), formed
// here and written to x_out [S,32] (the backward needs it) -- the scatter-mean launch in front of the cell folded in
template <bool MEAN>
__global__ __launch_bounds__(64 * GRU_WF) void gru_fwd_kernel(
    const float* __restrict__ x, const int32_t* __restrict__ perm, const int32_t* __restrict__ off,
    float* __restrict__ x_out, const float* __restrict__ h, const float* __restrict__ Wig,
    const float* __restrict__ big, const float* __restrict__ Wih, const float* __restrict__ Whh,
    const float* __restrict__ bih, const float* __restrict__ bhh, float* __restrict__ hy, int64_t S) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  GruLds& L = *reinterpret_cast<GruLds*>(smem);
  gru_load_weights<64 * GRU_WF>(L, Wig, Wih, Whh, false);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t nw = (int64_t)gridDim.x * GRU_WF;
  for (int64_t row = (int64_t)blockIdx.x * GRU_WF + wave; row < S; row += nw) {
    float xv;
    if (MEAN) {
      xv = gru_seg_mean(x, perm, off, row, lane);
      if (lane < 32) x_out[row * GC + lane] = xv;
    } else {
      xv = x[row * GC + (lane & 31)];
    }
    const GruRow R = gru_row_fwd(L, wave, lane, xv, h, big, bih, bhh, row);
    if (lane < 32) hy[row * GC + lane] = R.hy;
  }
}

// backward: recomputes the row forward, then the gradients; parameter gradients accumulate in registers over the
// wave's rows and go to a per-wave slab partial[wave_global][GRU_P]
__global__ __launch_bounds__(64 * GRU_WAVES) void gru_bwd_kernel(
    const float* __restrict__ x, const float* __restrict__ h, const float* __restrict__ Wig,
    const float* __restrict__ big, const float* __restrict__ Wih, const float* __restrict__ Whh,
    const float* __restrict__ bih, const float* __restrict__ bhh, const float* __restrict__ dhy,
    const float* __restrict__ dhy2, int64_t dhy2_pitch,
    float* __restrict__ dx, float* __restrict__ dh, float* __restrict__ partial, int64_t S) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  GruLds& L = *reinterpret_cast<GruLds*>(smem);
  gru_load_weights<64 * GRU_WAVES>(L, Wig, Wih, Whh, true);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 31, hi = lane >> 5;
  const bool two = lane < 32;
  const int64_t nw = (int64_t)gridDim.x * GRU_WAVES;
  // parameter-gradient accumulators: lane owns column j = c and rows o = hi + 2t
  float aWih[48], aWhh[48], aWig[16];
#pragma unroll
  for (int t = 0; t < 48; ++t) aWih[t] = aWhh[t] = 0.f;
#pragma unroll
  for (int t = 0; t < 16; ++t) aWig[t] = 0.f;
  float abih1 = 0.f, abih2 = 0.f, abhh1 = 0.f, abhh2 = 0.f, abig = 0.f;   // biases at o = lane, o = lane+64, c
  float* hrow = L.row[wave][0];
  float* xrow = L.row[wave][1];
  float* dgi = L.row[wave][2];
  float* dgh = L.row[wave][3];
  for (int64_t row = (int64_t)blockIdx.x * GRU_WAVES + wave; row < S; row += nw) {
    const GruRow R = gru_row_fwd(L, wave, lane, x[row * GC + c], h, big, bih, bhh, row);
    // upstream gradient = dhy (+ dhy2: the output gradient of this hidden state in a concatenated sequence, added here
    // instead of by a launch of its own; either may be NULL)
    float g = dhy ? dhy[row * GC + c] : 0.0f;
    if (dhy2) g = dhy ? g + dhy2[row * dhy2_pitch + c] : dhy2[row * dhy2_pitch + c];
    // gates (lanes 0..31 meaningful; upper half mirrors the same c)
    const float dn = g * (1.0f - R.z);
    const float dz = g * (R.h - R.n);
    float dhc = g * R.z;                                  // direct path hy <- h
    const float dn_pre = dn * (1.0f - R.n * R.n);
    const float ghn = R.gh2 + bhh[64 + c];                // lane<32: o = 64+c
    const float dr = dn_pre * ghn;
    const float dz_pre = dz * R.z * (1.0f - R.z);
    const float dr_pre = dr * R.r * (1.0f - R.r);
    // scatter to the o-indexed layout: first value o = lane, second o = lane + 64 (lane < 32)
    const float dz_pre_up = __shfl(dz_pre, c, 64);        // value of channel c for lanes 32..63
    const float y_i1 = two ? dr_pre : dz_pre_up;          // d gi_n[o=lane]
    const float y_h1 = y_i1;                              // d gh_n[o=lane] (same for r and z blocks)
    const float y_i2 = dn_pre;                            // d gi_n[64+c]
    const float y_h2 = dn_pre * R.r;                      // d gh_n[64+c]
    abih1 += y_i1;
    abhh1 += y_h1;
    if (two) {
      abih2 += y_i2;
      abhh2 += y_h2;
    }
    // rownorm backward: dg = (dy - mean(dy) - yhat * mean(dy*yhat)) * sig
    const float inv96 = 1.0f / (float)G3;
    const float m1i = wsum(y_i1 + (two ? y_i2 : 0.f)) * inv96;
    const float m2i = wsum(y_i1 * R.gi1 + (two ? y_i2 * R.gi2 : 0.f)) * inv96;
    const float m1h = wsum(y_h1 + (two ? y_h2 : 0.f)) * inv96;
    const float m2h = wsum(y_h1 * R.gh1 + (two ? y_h2 * R.gh2 : 0.f)) * inv96;
    const float dgi1 = (y_i1 - m1i - R.gi1 * m2i) * R.sig_i;
    const float dgi2 = (y_i2 - m1i - R.gi2 * m2i) * R.sig_i;
    const float dgh1 = (y_h1 - m1h - R.gh1 * m2h) * R.sig_h;
    const float dgh2 = (y_h2 - m1h - R.gh2 * m2h) * R.sig_h;
    dgi[lane] = dgi1;
    dgh[lane] = dgh1;
    if (two) {
      dgi[64 + lane] = dgi2;
      dgh[64 + lane] = dgh2;
    }
    // dxin[j] = sum_o Wih[o][j] dgi[o],  dh[j] += sum_o Whh[o][j] dgh[o]    (lane j = c; halves split o)
    float dxin = 0.f, dhh = 0.f;
#pragma unroll     // fully: aWih / aWhh must be indexed by constants to stay in registers (a partial unroll put the 96
                   // accumulators into scratch memory: 400 bytes per thread, a load + store per accumulate)
    for (int t = 0; t < 48; ++t) {
      const int o = hi + 2 * t;
      const float gi_o = dgi[o], gh_o = dgh[o];
      dxin += L.Wih[o][c] * gi_o;
      dhh += L.Whh[o][c] * gh_o;
      aWih[t] += gi_o * xrow[c];     // dWih[o][j=c]
      aWhh[t] += gh_o * hrow[c];
    }
    dxin += __shfl_xor(dxin, 32, 64);
    dhh += __shfl_xor(dhh, 32, 64);
    // input gate: xin = gin * x
    const float dxc = dxin * R.gin;
    const float dgin = dxin * R.x;
    const float dgpre = dgin * R.gin * (1.0f - R.gin);
    abig += dgpre;
    float* dgp = dgi;                 // reuse the scratch row for dgpre (all reads of dgi are done)
    if (lane < 32) dgp[c] = dgpre;
    float dhg = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int cc = hi + 2 * t;      // rows of Wig handled by this half
      const float gp = dgp[cc];
      dhg += L.Wig[cc][c] * gp;       // dh[j=c] += Wig[cc][j] * dgpre[cc]
      aWig[t] += gp * hrow[c];        // dWig[cc][j=c]
    }
    dhg += __shfl_xor(dhg, 32, 64);
    dhc += dhh + dhg;
    if (lane < 32) {
      dx[row * GC + c] = dxc;
      dh[row * GC + c] = dhc;
    }
  }
  // per-workgroup slab [Wih 96x32][Whh 96x32][Wig 32x32][bih 96][bhh 96][big 32]: every wave leaves its accumulators in a
  // slab of its own (the weight images are no longer needed: 4 x 29.7 KB of LDS), then all threads add the four copies
  // of an element in wave order -- the order of the wave-by-wave merge this replaces (three barriers fewer) -- and the
  // sum goes out coalesced
  __syncthreads();                                   // every wave is done with the weights: reuse their LDS
  float* const mine = reinterpret_cast<float*>(smem) + wave * GRU_P;
#pragma unroll
  for (int t = 0; t < 48; ++t) {
    const int o = hi + 2 * t;
    mine[o * GC + c] = aWih[t];
    mine[G3 * GC + o * GC + c] = aWhh[t];
  }
#pragma unroll
  for (int t = 0; t < 16; ++t) mine[2 * G3 * GC + (hi + 2 * t) * GC + c] = aWig[t];
  {
    float* pb = mine + 2 * G3 * GC + GC * GC;
    pb[lane] = abih1;
    pb[G3 + lane] = abhh1;
    if (two) {
      pb[64 + lane] = abih2;
      pb[G3 + 64 + lane] = abhh2;
      pb[2 * G3 + lane] = abig;
    }
  }
  __syncthreads();
  const float* const s0 = reinterpret_cast<const float*>(smem);
  float* p = partial + (int64_t)blockIdx.x * GRU_P;
  for (int f = threadIdx.x; f < GRU_P; f += blockDim.x) {
    float v = s0[f];
#pragma unroll
    for (int w = 1; w < GRU_WAVES; ++w) v += s0[w * GRU_P + f];
    p[f] = v;
  }
}

constexpr int RED_OUT = 32;    // outputs per workgroup of the slab reduce
constexpr int RED_LANES = 8;   // slab lanes per output

__global__ __launch_bounds__(RED_OUT * RED_LANES) void gru_reduce_kernel(
    const float* __restrict__ partial, int nslab, float* __restrict__ dWih, float* __restrict__ dWhh,
    float* __restrict__ dWig, float* __restrict__ dbih, float* __restrict__ dbhh, float* __restrict__ dbig) {
  // 32 outputs x 8 slab lanes per workgroup: lane y sums slabs y, y+8, ... in order, then the 8 partial sums are
  // added in lane order -- a fixed summation tree, independent of the launch
  __shared__ float part[RED_LANES][RED_OUT];
  const int ox = threadIdx.x % RED_OUT, sy = threadIdx.x / RED_OUT;
  const int i = blockIdx.x * RED_OUT + ox;
  float s = 0.f;
  if (i < GRU_P) {
    int k = sy;
    for (; k + 15 * RED_LANES < nslab; k += 16 * RED_LANES) {      // sixteen slabs in flight, added in the same order
      float t[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) t[j] = partial[(int64_t)(k + j * RED_LANES) * GRU_P + i];
#pragma unroll
      for (int j = 0; j < 16; ++j) s += t[j];
    }
    for (; k < nslab; k += RED_LANES) s += partial[(int64_t)k * GRU_P + i];
  }
  part[sy][ox] = s;
  __syncthreads();
  if (sy != 0 || i >= GRU_P) return;
#pragma unroll
  for (int y = 1; y < RED_LANES; ++y) s += part[y][ox];
  if (i < G3 * GC)
    dWih[i] = s;
  else if (i < 2 * G3 * GC)
    dWhh[i - G3 * GC] = s;
  else if (i < 2 * G3 * GC + GC * GC)
    dWig[i - 2 * G3 * GC] = s;
  else if (i < 2 * G3 * GC + GC * GC + G3)
    dbih[i - (2 * G3 * GC + GC * GC)] = s;
  else if (i < 2 * G3 * GC + GC * GC + 2 * G3)
    dbhh[i - (2 * G3 * GC + GC * GC + G3)] = s;
  else
    dbig[i - (2 * G3 * GC + GC * GC + 2 * G3)] = s;
}

// workgroups of the backward launch: three rows per wave (191 workgroups on 2,289 rows).  What a workgroup pays once --
// staging, the wave-ordered slab merge, the 30 KB slab -- is ~11 us, a row ~5: 2 rows per wave / 256 workgroups take the
// same 25 us but leave a third more slabs to gru_reduce_kernel (19.5 -> 16.4 us); 1 row per wave / 572 workgroups 37 us
int gru_blocks(int64_t S, int waves = GRU_WAVES, int rows_per_wave = 3) {
  const char* e = getenv("WSIS_GRU_ROWS");          // (tuning knob, read per call) rows per wave of the backward launch
  if (e && waves == GRU_WAVES && atoi(e) > 0) rows_per_wave = atoi(e);
  int64_t b = ceil_div(S, waves * rows_per_wave);
  if (b < 1) b = 1;
  if (b > 256) b = 256;
  return (int)b;
}

}  // namespace

extern "C" {

int64_t wsis_gru_cell_workspace_bytes(int64_t S) {
  if (S < 0) return -1;
  return (int64_t)gru_blocks(S) * GRU_P * (int64_t)sizeof(float) + 256;
}

static int gru_fwd_impl(const float* d_x, const int32_t* d_perm, const int32_t* d_off, float* d_x_out, const float* d_h,
                        const float* d_Wig, const float* d_big, const float* d_Wih, const float* d_Whh, const float* d_bih,
                        const float* d_bhh, float* d_hy, int64_t S, void* stream) {
  const size_t lds = sizeof(GruLds);
  static bool attr_set = false;       // (one process per GPU: include/wsis_hip.h)
  if (!attr_set) {
    WSIS_HIP_CHECK(hipFuncSetAttribute((const void*)gru_fwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    WSIS_HIP_CHECK(hipFuncSetAttribute((const void*)gru_fwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  if (d_perm)
    hipLaunchKernelGGL(gru_fwd_kernel<true>, dim3(gru_blocks(S, GRU_WF, 1)), dim3(64 * GRU_WF), lds, as_stream(stream), d_x,
                       d_perm, d_off, d_x_out, d_h, d_Wig, d_big, d_Wih, d_Whh, d_bih, d_bhh, d_hy, S);
  else
    hipLaunchKernelGGL(gru_fwd_kernel<false>, dim3(gru_blocks(S, GRU_WF, 1)), dim3(64 * GRU_WF), lds, as_stream(stream), d_x,
                       d_perm, d_off, d_x_out, d_h, d_Wig, d_big, d_Wih, d_Whh, d_bih, d_bhh, d_hy, S);
  WSIS_LAUNCH_CHECK();
  return WSIS_OK;
}

int wsis_gru_cell_fwd(const float* d_x, const float* d_h, const float* d_Wig, const float* d_big,
                      const float* d_Wih, const float* d_Whh, const float* d_bih, const float* d_bhh, float* d_hy,
                      int64_t S, int32_t C, void* stream) {
  WSIS_REQUIRE(S >= 0 && C == GC, "GRUCellEx kernel supports C == 32");
  if (S == 0) return WSIS_OK;
  WSIS_REQUIRE(d_x && d_h && d_Wig && d_big && d_Wih && d_Whh && d_bih && d_bhh && d_hy, "null pointer");
  return gru_fwd_impl(d_x, nullptr, nullptr, nullptr, d_h, d_Wig, d_big, d_Wih, d_Whh, d_bih, d_bhh, d_hy, S, stream);
}

int wsis_gru_cell_fwd_mean(const float* d_m, const int32_t* d_perm, const int32_t* d_off, float* d_x_out, const float* d_h,
                           const float* d_Wig, const float* d_big, const float* d_Wih, const float* d_Whh, const float* d_bih,
                           const float* d_bhh, float* d_hy, int64_t S, int32_t C, void* stream) {
  WSIS_REQUIRE(S >= 0 && C == GC, "GRUCellEx kernel supports C == 32");
  if (S == 0) return WSIS_OK;
  WSIS_REQUIRE(d_m && d_perm && d_off && d_x_out && d_h && d_Wig && d_big && d_Wih && d_Whh && d_bih && d_bhh && d_hy,
               "null pointer");
  return gru_fwd_impl(d_m, d_perm, d_off, d_x_out, d_h, d_Wig, d_big, d_Wih, d_Whh, d_bih, d_bhh, d_hy, S, stream);
}

// one backward evaluation; slot >= 0 selects the slab region of a sequence (wsis_gru_cell_bwd_seq), n_reduce > 0
// runs the fixed-order slab reduce over that many slabs
static int gru_bwd_impl(const float* d_x, const float* d_h, const float* d_Wig, const float* d_big, const float* d_Wih,
                        const float* d_Whh, const float* d_bih, const float* d_bhh, const float* d_dhy, const float* d_dhy2,
                        int64_t dhy2_pitch, float* d_dx,
                        float* d_dh, float* d_dWig, float* d_dbig, float* d_dWih, float* d_dWhh, float* d_dbih,
                        float* d_dbhh, int64_t S, int slot, int n_reduce, void* d_ws, hipStream_t st) {
  const size_t lds = std::max(sizeof(GruLds), (size_t)GRU_WAVES * GRU_P * sizeof(float));      // (the four slabs of the merge)
  const int nb = gru_blocks(S);
  static bool attr_set = false;
  if (!attr_set) {
    WSIS_HIP_CHECK(hipFuncSetAttribute((const void*)gru_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  float* base = static_cast<float*>(d_ws);
  float* partial = base + (int64_t)slot * nb * GRU_P;
  hipLaunchKernelGGL(gru_bwd_kernel, dim3(nb), dim3(64 * GRU_WAVES), lds, st, d_x, d_h, d_Wig, d_big, d_Wih, d_Whh,
                     d_bih, d_bhh, d_dhy, d_dhy2, dhy2_pitch, d_dx, d_dh, partial, S);
  WSIS_LAUNCH_CHECK();
  if (n_reduce > 0) {
    hipLaunchKernelGGL(gru_reduce_kernel, dim3((GRU_P + RED_OUT - 1) / RED_OUT), dim3(RED_OUT * RED_LANES), 0, st, base,
                       n_reduce, d_dWih, d_dWhh, d_dWig, d_dbih, d_dbhh, d_dbig);
    WSIS_LAUNCH_CHECK();
  }
  return WSIS_OK;
}

int wsis_gru_cell_bwd(const float* d_x, const float* d_h, const float* d_Wig, const float* d_big,
                      const float* d_Wih, const float* d_Whh, const float* d_bih, const float* d_bhh,
                      const float* d_dhy, float* d_dx, float* d_dh, float* d_dWig, float* d_dbig, float* d_dWih,
                      float* d_dWhh, float* d_dbih, float* d_dbhh, int64_t S, int32_t C, void* d_ws,
                      int64_t ws_bytes, void* stream) {
  WSIS_REQUIRE(S >= 1 && C == GC, "GRUCellEx kernel supports C == 32, S >= 1");
  WSIS_REQUIRE(d_x && d_h && d_Wig && d_big && d_Wih && d_Whh && d_bih && d_bhh && d_dhy && d_dx && d_dh &&
                   d_dWig && d_dbig && d_dWih && d_dWhh && d_dbih && d_dbhh && d_ws,
               "null pointer");
  WSIS_REQUIRE(ws_bytes >= wsis_gru_cell_workspace_bytes(S), "workspace too small");
  return gru_bwd_impl(d_x, d_h, d_Wig, d_big, d_Wih, d_Whh, d_bih, d_bhh, d_dhy, nullptr, 0, d_dx, d_dh, d_dWig, d_dbig, d_dWih,
                      d_dWhh, d_dbih, d_dbhh, S, 0, gru_blocks(S), d_ws, as_stream(stream));
}

int wsis_gru_cell_bwd_seq(const float* d_x, const float* d_h, const float* d_Wig, const float* d_big,
                          const float* d_Wih, const float* d_Whh, const float* d_bih, const float* d_bhh,
                          const float* d_dhy, const float* d_dhy2, int64_t dhy2_pitch, float* d_dx, float* d_dh, float* d_dWig,
                          float* d_dbig, float* d_dWih, float* d_dWhh, float* d_dbih, float* d_dbhh, int64_t S, int32_t C,
                          int32_t slot, int32_t n_slots, int32_t finish, void* d_ws, int64_t ws_bytes, void* stream) {
  WSIS_REQUIRE(S >= 1 && C == GC, "GRUCellEx kernel supports C == 32, S >= 1");
  WSIS_REQUIRE(n_slots >= 1 && slot >= 0 && slot < n_slots, "bad slot");
  WSIS_REQUIRE(d_x && d_h && d_Wig && d_big && d_Wih && d_Whh && d_bih && d_bhh && (d_dhy || d_dhy2) && d_dx && d_dh && d_ws,
               "null pointer");
  WSIS_REQUIRE(!d_dhy2 || dhy2_pitch >= GC, "pitch of the second gradient");
  WSIS_REQUIRE(!finish || (d_dWig && d_dbig && d_dWih && d_dWhh && d_dbih && d_dbhh), "null pointer");
  WSIS_REQUIRE(ws_bytes >= (wsis_gru_cell_workspace_bytes(S) - 256) * n_slots + 256, "workspace too small");
  return gru_bwd_impl(d_x, d_h, d_Wig, d_big, d_Wih, d_Whh, d_bih, d_bhh, d_dhy, d_dhy2, dhy2_pitch, d_dx, d_dh, d_dWig, d_dbig,
                      d_dWih, d_dWhh, d_dbih, d_dbhh, S, slot, finish ? gru_blocks(S) * n_slots : 0, d_ws, as_stream(stream));
}

}  // extern "C"
