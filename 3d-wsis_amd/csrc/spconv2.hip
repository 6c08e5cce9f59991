// Sparse convolution, wave-autonomous form (the kernel of every forward and dIn pass whose channel
// counts are multiples of 32 -- all UNet layers but the 6-channel input conv).
//
//   out[r,:] = sum_k X[nbr[k][r],:] @ W[k]      with the weights given as B^T: WT[k][cout][cin]
//
// Work item = (32 output rows in tile order) x (NB blocks of 32 output channels).  ONE WAVE owns a work item (NW = 1),
// or NW waves of a workgroup split its (active offset, 32-channel chunk) steps round-robin and add their accumulators
// through LDS in wave order at the end (small pyramid levels: parallelism from the offsets instead of a second
// "partial slab + reduce" launch).  There is no workgroup barrier inside the walk and no register staging:
//   * the gathered input rows (A, 32 rows x 128 B) of a step go global -> LDS by LDS-DMA through a raw buffer
//     descriptor (per-lane 32-bit offset = the gather, missing pair = out of range = zeros) into the wave's private
//     ring (DA slots); the weight rows (B^T) go straight to registers (BD, default) or through a second ring;
//     completion is tracked with counted s_waitcnt vmcnt;
//   * the MFMA fragments of step t+1 are read from LDS (conflict-free ds_read_b128: the 16-byte pieces of a row are
//     XOR-swizzled on the SOURCE side, the LDS image stays lane-linear as the DMA requires) while the 16*NB
//     v_mfma_f32_32x32x2_f32 of step t run from registers;
//   * inactive (slice, offset) pairs cost nothing at all (the wave walks only its own slice's active offsets).
// Exact fp32 (k-ordered fma chain); the order of additions is fixed by (offset, chunk, wave), so results are
// run-to-run identical; with NW = 1 they are bit-identical to spconv_fwd_kernel.
#include <cstdlib>
#include <type_traits>

#include "common.h"

using namespace wsis;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int SL = 32;            // rows per work item
constexpr int KMAX = 32;          // kernel offsets (mask width)
constexpr int DB = 2;             // ring depth of the weight rows (steps); the gathered rows: template DA
constexpr int A_BYTES = SL * 128;
constexpr int HDR_BYTES = (KMAX + 2) * 128;   // nbT [32][32] + rowId [32] + klist [32]: ONE per workgroup (the waves of a
                                              // work item share the slice; each writes the identical header itself)

__device__ __attribute__((aligned(256))) float g_zero_row[64];   // source of masked rows (never written)

// BD: the weight fragments go global -> registers directly (no LDS ring for B: 8 NB KB less LDS per wave, more waves
// per CU); otherwise they take the same LDS-DMA ring path as the gathered rows
template <int NB, int DA, bool BD>
struct Layout {
  static constexpr int B_BYTES = NB * 32 * 128;
  static constexpr int WAVE_BYTES = DA * A_BYTES + (BD ? 0 : DB * B_BYTES);
};

// gathered rows come through a raw buffer descriptor: 32-bit byte offsets (the table in LDS holds row * pitch, so an
// issue is one add) and offsets past the end read as zero, so a missing pair (0x80000000) needs no select
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ void bdma16(rsrc_t r, uint32_t voff, uint32_t soff, void* lds_dst) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_dst, 16, (int)voff, (int)soff,
                                           0, 0);
}
constexpr uint32_t NO_ROW = 0x80000000u;

// optional reduction of the epilogue for a dIn pass whose output dy feeds the backward of a BatchNorm(+ReLU): instead
// of the (sum, centred sum of squares) statistics the slice partials are (sum dz, sum dz * xhat) with
// xhat = (x - mean) rstd, dz = dy masked by the ReLU -- the reduction pass of wsis_bn_bwd without re-reading dy and x
struct BnEpi {
  const float* x;            // the BatchNorm's input [M_out, Cout]; nullptr = plain statistics
  const float* mean;
  const float* var;
  const float* gamma;        // may be nullptr (1)
  const float* beta;         // may be nullptr (0)
  float eps;
  int relu;
};
struct BnCoef {
  float mu, rstd, gm, bt;
};
// BatchNorm(+ReLU) of the convolution's INPUT, applied while the gathered rows are read as MFMA fragments (FB kernels):
// the activation relu(bn(x)) of sparse_unet3d.py:128-137 is never written to memory.  mean / var: batch statistics
// (training) or the running statistics (evaluation).
struct BnIn {
  const float* mean;         // nullptr: off
  const float* var;
  const float* gamma;        // may be nullptr (1)
  const float* beta;         // may be nullptr (0)
  float eps;
  int relu;
};
// Finish of the output statistics inside the launch (training): the (slice, block) workgroups of a chunk of slices
// draw tickets; the last one adds the chunk's partials in the order of bn_chunk_centred_stage (bit-identical to
// wsis_bn_stats_finalize), the last chunk adds the chunk rows and writes mean / var / running statistics of up to two
// BatchNorm layers that normalise this tensor (a skip connection feeds a second one).
struct StatFin {
  double* chunk;             // [G][Cout / 32][3][32] fp64 chunk rows; nullptr: off
  unsigned* tickets;         // [G * Cout / 32] chunk tickets + [Cout / 32] final tickets, zero between launches
  int G, per;                // chunks, partial rows per chunk
  float* mean[2];
  float* var[2];
  float* rmean[2];
  float* rvar[2];
  float momentum[2];
  int n_targets;
};
__device__ __forceinline__ BnCoef bn_coef(const BnEpi& e, int c) {
  BnCoef k;
  k.mu = e.mean[c];
  k.rstd = rsqrtf(e.var[c] + e.eps);
  k.gm = e.gamma ? e.gamma[c] : 1.0f;
  k.bt = e.beta ? e.beta[c] : 0.0f;
  return k;
}
// (dz, dz * xhat) of one element, the arithmetic of bn_bwd_partial_kernel
__device__ __forceinline__ void bn_terms(const BnCoef& k, int relu, float dy, float xv, float& dz, float& dzx) {
  const float xh = (xv - k.mu) * k.rstd;
  dz = (relu && xh * k.gm + k.bt <= 0.0f) ? 0.0f : dy;
  dzx = dz * xh;
}

__device__ __forceinline__ void dma16(const void* src, void* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// swizzle of the 16-byte pieces of a 128-byte row: piece p of row r lives in slot p ^ swz(r); a 16-lane group of a
// ds_read_b128 (rows 0-3,12-15,20-27 / 4-11,16-19,28-31 of one half) then covers 16 distinct 16-byte bank groups
__device__ __forceinline__ int swz(int row) { return (row >> 1) & 7; }

// In-launch finish of the BatchNorm statistics of the tensor this launch writes (StatFin), called by ONE wave of the
// workgroup that has just stored the partials of slice `slice`, channels col0 .. col0 + 31 (block cbk of CB).
// Level 1: the last workgroup of a chunk of slices to arrive adds the chunk's partial rows; level 2: the last chunk adds
// the chunk rows.  Both in exactly the order of bn_chunk_centred_stage (thread (channel, partial lane pl of 8) walks
// rows lo + pl, + 8, ...; the eight lane sums are added in lane order), so mean / var / running statistics come out
// bit-identical to wsis_bn_stats_finalize -- whichever workgroup happens to be last.  `red`: 6 KB of LDS scratch.
__device__ __forceinline__ void stat_finish(const StatFin& fin, const float* __restrict__ stats, int64_t M_out, int Cout,
                                            int64_t slice, int col0, int CB, int cbk, double* red) {
  const int lane = threadIdx.x & 63;
  const int cl = lane & 31, ph = lane >> 5;
  const int c = col0 + cl;
  const int64_t n_part = (M_out + 31) >> 5;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wave's partial stores have left
  const int g0 = (int)(slice / fin.per);
  int last = 0;
  if (lane == 0) {
    const int64_t lo = (int64_t)g0 * fin.per;
    const int64_t cnt = (lo + fin.per < n_part ? lo + fin.per : n_part) - lo;
    unsigned* tk = fin.tickets + g0 * CB + cbk;
    const unsigned t = atomicAdd(tk, 1u);
    last = t == (unsigned)cnt - 1u;
    if (last) __hip_atomic_store(tk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
  }
  last = __builtin_amdgcn_readfirstlane(last);
  if (!last) return;
  double S = 0.0, Q = 0.0, W = 0.0;
  {
    const int64_t lo = (int64_t)g0 * fin.per;
    const int64_t hi = lo + fin.per < n_part ? lo + fin.per : n_part;
    double s[4] = {0.0, 0.0, 0.0, 0.0}, q[4] = {0.0, 0.0, 0.0, 0.0}, w[4] = {0.0, 0.0, 0.0, 0.0};
    for (int64_t b0 = lo; b0 < hi; b0 += 8) {
      float sf[4], qf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {            // partial lanes pl = ph * 4 + i; all eight loads in flight
        const int64_t b = b0 + ph * 4 + i;
        const int64_t bb = b < hi ? b : lo;
        sf[i] = ld_sc1(stats + (bb * 2 + 0) * Cout + c);
        qf[i] = ld_sc1(stats + (bb * 2 + 1) * Cout + c);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int64_t b = b0 + ph * 4 + i;
        if (b < hi) {
          const int64_t left = M_out - b * 32;
          const double si = sf[i];
          s[i] += si;
          q[i] += qf[i];
          w[i] += si * si * (left < 32 ? 1.0 / (double)left : 0.03125);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      red[(0 * 8 + ph * 4 + i) * 32 + cl] = s[i];
      red[(1 * 8 + ph * 4 + i) * 32 + cl] = q[i];
      red[(2 * 8 + ph * 4 + i) * 32 + cl] = w[i];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < 8; ++j) {              // fixed order
      S += red[(0 * 8 + j) * 32 + cl];
      Q += red[(1 * 8 + j) * 32 + cl];
      W += red[(2 * 8 + j) * 32 + cl];
    }
  }
  const int G = fin.G;
  if (G > 1) {
    double* o = fin.chunk + ((int64_t)g0 * CB + cbk) * 96;
    if (ph == 0) {
      st_sc1(o + cl, S);
      st_sc1(o + 32 + cl, Q);
      st_sc1(o + 64 + cl, W);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int last2 = 0;
    if (lane == 0) {
      unsigned* tk = fin.tickets + G * CB + cbk;
      const unsigned t = atomicAdd(tk, 1u);
      last2 = t == (unsigned)G - 1u;
      if (last2) __hip_atomic_store(tk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    last2 = __builtin_amdgcn_readfirstlane(last2);
    if (!last2) return;
    double s2[4] = {0.0, 0.0, 0.0, 0.0}, q2[4] = {0.0, 0.0, 0.0, 0.0}, w2[4] = {0.0, 0.0, 0.0, 0.0};
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      for (int g = ph * 4 + i; g < G; g += 8) {
        const double* oc = fin.chunk + ((int64_t)g * CB + cbk) * 96;
        s2[i] += ld_sc1(oc + cl);
        q2[i] += ld_sc1(oc + 32 + cl);
        w2[i] += ld_sc1(oc + 64 + cl);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      red[(0 * 8 + ph * 4 + i) * 32 + cl] = s2[i];
      red[(1 * 8 + ph * 4 + i) * 32 + cl] = q2[i];
      red[(2 * 8 + ph * 4 + i) * 32 + cl] = w2[i];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    S = Q = W = 0.0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      S += red[(0 * 8 + j) * 32 + cl];
      Q += red[(1 * 8 + j) * 32 + cl];
      W += red[(2 * 8 + j) * 32 + cl];
    }
  }
  if (ph == 0 && c < Cout) {
    for (int tg = 0; tg < fin.n_targets; ++tg)
      bn_finish_centred(S, Q, W, M_out, c, fin.mean[tg], fin.var[tg], fin.rmean[tg], fin.rvar[tg], fin.momentum[tg]);
  }
}

template <int NB, int NW, int DA, bool BD, bool DIAG = false, bool FB = false>
__global__ __launch_bounds__(64 * NW) void spconv_fwd2_kernel(
    const float* __restrict__ X, const int32_t* __restrict__ nbrS, const int32_t* __restrict__ order,
    const float* __restrict__ WT, const float* __restrict__ bias, const float* __restrict__ residual,
    float* __restrict__ out, float* __restrict__ partial, int64_t M_out, int K, int Cin, int Cout, int flip_deal,
    uint32_t x_bytes, float* __restrict__ stats, BnEpi epi, BnIn bin, StatFin fin,
    unsigned long long* __restrict__ dbg = nullptr) {
  static_assert(!FB || (BD && NB == 1), "the fused input BatchNorm is built for the weights-to-registers form");
  // flip_deal: bit 0 = offset k uses weight slice K - 1 - k; bit 1 = the waves of a work item are dealt the slice's
  // ACTIVE offsets round-robin (see the ownership block below)
  const int flip = flip_deal & 1;
  const bool deal_active = (flip_deal & 2) != 0;
  // stats (optional, final pass only): per-slice BatchNorm partials of the FINISHED output rows (bias and residual
  // included), stats[(slice * 2 + {0: sum, 1: sum of squared deviations from the SLICE mean}) * Cout + channel] -- the
  // statistics pass of the BatchNorm that consumes this tensor (sparse_unet3d.py:128-137) without re-reading it.
  // Centred per slice (and combined in fp64 by wsis_bn_stats_finalize): no E[x^2] - mean^2 cancellation in fp32.
  // DIAG build only (tools/conv2_stamps.py): per-workgroup stamps, dbg[blockIdx.x * 8 + i] =
  // {realtime at entry, realtime at exit, cycles: prologue, walk, epilogue, steps, HW_ID, 0}
  unsigned long long d_t0 = 0, d_r0 = 0, d_t1 = 0, d_t2 = 0;
  if (DIAG) {
    d_r0 = __builtin_amdgcn_s_memrealtime();
    d_t0 = __builtin_amdgcn_s_memtime();
  }
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  using L = Layout<NB, DA, BD>;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r31 = lane & 31, half = lane >> 5;
  int32_t* const nbT = reinterpret_cast<int32_t*>(lds);
  int32_t* const rowId = nbT + KMAX * 32;
  int32_t* const klist = rowId + 32;
  unsigned char* const Aring = lds + HDR_BYTES + wave * L::WAVE_BYTES;
  unsigned char* const Bring = Aring + DA * A_BYTES;
  // FB: per-channel (mean, scale, shift) of the input BatchNorm behind the rings, 3 x Cin floats; every wave writes
  // the identical values itself (like the header: no workgroup barrier in the prologue)
  float* const coef = reinterpret_cast<float*>(lds + HDR_BYTES + NW * L::WAVE_BYTES);
  if (FB) {
    for (int c = lane; c < Cin; c += 64) {
      coef[c] = bin.mean[c];
      coef[Cin + c] = (bin.gamma ? bin.gamma[c] : 1.0f) * rsqrtf(bin.var[c] + bin.eps);
      coef[2 * Cin + c] = bin.beta ? bin.beta[c] : 0.0f;
    }
  }

  const int64_t t0 = (int64_t)blockIdx.x * SL;
  const int col0 = blockIdx.y * (NB * 32);
  const int nchunk = Cin >> 5;

  // ---- the slice's rows and its column of the packed gather table (every load in flight at once)
  const int64_t t = t0 + r31;
  const uint32_t a_pitch32 = (uint32_t)Cin * 4u;
  int32_t my_row = -1;
  if (t < M_out) my_row = order ? order[t] : (int32_t)t;
  uint32_t mask = 0u;
  {
    // branch-free: every lane always issues its 16 loads (clamped address), masking happens afterwards -- with
    // conditional loads hipcc waits for each one before issuing the next
    int32_t v[KMAX / 2];
    const int32_t* tab = nbrS ? nbrS : order;      // any readable address for the masked lanes
#pragma unroll
    for (int j = 0; j < KMAX / 2; ++j) {
      const int k = 2 * j + half;
      const bool ok = k < K && t < M_out && nbrS != nullptr;      // (independent of the order[] load above)
      v[j] = tab ? tab[ok ? (int64_t)k * M_out + t : 0] : 0;
    }
#pragma unroll
    for (int j = 0; j < KMAX / 2; ++j) {
      const int k = 2 * j + half;
      const bool ok = k < K && t < M_out;
      const int32_t g = ok ? (nbrS ? v[j] : my_row) : -1;
      nbT[k * 32 + r31] = g >= 0 ? (int32_t)((uint32_t)g * a_pitch32) : (int32_t)NO_ROW;
      const unsigned long long b = __ballot(g >= 0);
      if ((uint32_t)b) mask |= 1u << (2 * j);
      if ((uint32_t)(b >> 32)) mask |= 1u << (2 * j + 1);
    }
  }
  if (lane < 32) rowId[lane] = my_row;
  // offsets of this wave: kernel offset k belongs to slab z = k % ZS (levels with very few work items: partial slabs,
  // added in z order by spconv2_reduce_kernel) and, inside the workgroup, to wave (k / ZS) % NW.  The assignment
  // depends on k alone, so the order of additions of an output row -- and with it the result, bit for bit -- does
  // not depend on which other rows share its slice (tile order independent).  All scalar: the walk needs no LDS list.
  uint32_t mymask = 0u;
  {
    const int zs = gridDim.z, z = blockIdx.z;
    const int P = zs * NW, r = z + zs * wave;         // k belongs to this wave iff k mod (zs NW) == z + zs wave
    if (deal_active && NW > 1 && zs == 1) {
      // the j-th ACTIVE offset of the slice goes to wave j % NW: every wave of the work item gets the same number of
      // steps (+- one offset) whatever the slice's geometry -- with ownership by offset INDEX the waves of a
      // workgroup differ by up to 3x and the work item lasts as long as its busiest wave.  The price: which wave adds
      // which offset now depends on the slice's active set, so an output row's order of additions depends on the rows
      // it shares a slice with (still fixed for a given input: run-to-run identical, not tile-order independent).
      uint32_t m = mask;
      int j = 0;
      while (m) {
        const int k = __builtin_ctz(m);
        m &= m - 1u;
        if (j == wave) mymask |= 1u << k;
        j = j + 1 == NW ? 0 : j + 1;
      }
    } else if ((P & (P - 1)) == 0) {
      // the launch plans only produce powers of two: the owner test is a periodic bit pattern (the 27-iteration walk
      // with two runtime divisions per offset was ~1,600 scalar instructions of every work item's prologue)
      uint32_t pat = P == 1 ? 0xffffffffu : P == 2 ? 0x55555555u : P == 4 ? 0x11111111u : P == 8 ? 0x01010101u
                   : P == 16 ? 0x00010001u : 0x00000001u;
      pat = r < 32 ? pat << r : 0u;
      mymask = mask & pat;
    } else {
      uint32_t m = mask;
      while (m) {
        const int k = __builtin_ctz(m);
        m &= m - 1u;
        if (k % zs == z && (k / zs) % NW == wave) mymask |= 1u << k;
      }
    }
  }
  mymask = __builtin_amdgcn_readfirstlane(mymask);
  const int T = __builtin_popcount(mymask) * nchunk;               // steps of this wave: (offset, chunk), chunk inner

  f32x16 acc[NB];
#pragma unroll
  for (int cb = 0; cb < NB; ++cb)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[cb][i] = 0.0f;

  // per-lane constants of the DMA pieces: instruction i of a 32-row image covers rows i*8 + (lane >> 3)
  const int d_row = lane >> 3, d_piece = lane & 7;
  const char* const Xb = reinterpret_cast<const char*>(X);
  const char* const Wb = reinterpret_cast<const char*>(WT);
  const char* const zrow = reinterpret_cast<const char*>(g_zero_row) + d_piece * 16;
  const int64_t a_pitch = (int64_t)Cin * 4;

  // step generators (wave-uniform scalars): next (offset, chunk) of the gathered-row stream / of the weight stream
  struct Gen {
    uint32_t rem;
    int k, c;
    bool valid;
  };
  auto gen_init = [&](Gen& g) {
    g.rem = mymask;
    g.c = 0;
    g.valid = g.rem != 0u;
    g.k = g.valid ? __builtin_ctz(g.rem) : 0;
    g.rem &= g.rem - 1u;
  };
  auto gen_next = [&](Gen& g) {
    if (++g.c == nchunk) {
      g.c = 0;
      g.valid = g.valid && g.rem != 0u;
      g.k = g.rem ? __builtin_ctz(g.rem) : g.k;
      g.rem &= g.rem - 1u;
    }
  };
  // DMA of a step, branch-free: a finished stream re-reads the zero row (the piece count per iteration stays fixed,
  // which keeps the counted vmcnt waits valid in the tail)
  auto loadNb = [&](const Gen& g, int32_t (&nb)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) nb[i] = nbT[g.k * 32 + i * 8 + d_row];
  };
  const rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X), (short)0, (int)x_bytes, 0x00020000);
  uint32_t a_po[4];                      // swizzled 16-byte piece of this lane in instruction i
#pragma unroll
  for (int i = 0; i < 4; ++i) a_po[i] = (uint32_t)((d_piece ^ swz(i * 8 + d_row)) << 4);
  // DIAG experiments (tools/conv2_stamps.py, never in a product launch): flip_deal bit 4 = the weight registers are
  // loaded for the first step only, bit 5 = every gathered row is row 0 (what the two operand streams cost)
  const bool dbg_noB = DIAG && (flip_deal & 16), dbg_row0 = DIAG && (flip_deal & 32);
  auto issueA1 = [&](const Gen& g, const int32_t (&nb)[4], int slot, int i) {
    const uint32_t off = g.valid ? (dbg_row0 ? (uint32_t)(i * 8 + d_row) * a_pitch32 : (uint32_t)nb[i]) + a_po[i] : NO_ROW;
    bdma16(rsX, off, (uint32_t)g.c * 128u, Aring + slot * A_BYTES + i * 1024);
  };
  auto issueB1 = [&](const Gen& g, int slot, int i) {
    const int kk = flip ? K - 1 - g.k : g.k;
    const int n = i * 8 + d_row;                                   // output column inside the block group
    const char* src = g.valid ? Wb + (((int64_t)kk * Cout + col0 + n) * Cin + g.c * 32) * 4 + ((d_piece ^ swz(n & 31)) << 4)
                              : zrow;
    dma16(src, Bring + slot * L::B_BYTES + i * 1024);
  };
  // fragments: lane (row / column r31, half) takes channels half*16 .. +15 of its row: MFMA k index (step s, half)
  // <-> channel half*16 + s, the mapping of spconv_fwd_kernel (same order of additions)
  auto readfrag = [&](int aslot, int bslot, f32x4 (&a)[4], f32x4 (&b)[NB][4]) {
    const unsigned char* arow = Aring + aslot * A_BYTES + r31 * 128;
    const int sw = swz(r31);
#pragma unroll
    for (int q = 0; q < 4; ++q) a[q] = *reinterpret_cast<const f32x4*>(arow + (((half * 4 + q) ^ sw) << 4));
    if (BD) return;
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) {
      const unsigned char* brow = Bring + bslot * L::B_BYTES + (cb * 32 + r31) * 128;
#pragma unroll
      for (int q = 0; q < 4; ++q) b[cb][q] = *reinterpret_cast<const f32x4*>(brow + (((half * 4 + q) ^ sw) << 4));
    }
  };
  // BD: the lane's 16 weights of output column cb*32 + r31 (channels half*16 .. +15 of the step's chunk) straight from
  // WT [K][Cout][Cin]: 64 contiguous bytes; a finished stream reads the zero row
  auto loadB = [&](const Gen& g, f32x4 (&b)[NB][4]) {
    const int kk = flip ? K - 1 - g.k : g.k;
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) {
      const char* src = g.valid ? Wb + (((int64_t)kk * Cout + col0 + cb * 32 + r31) * Cin + g.c * 32 + half * 16) * 4
                                : reinterpret_cast<const char*>(g_zero_row);
#pragma unroll
      for (int q = 0; q < 4; ++q) b[cb][q] = *reinterpret_cast<const f32x4*>(src + q * 16);
    }
  };
  // MFMAs [s0, s1) of a step (k-ordered chain per output block)
  auto mfma = [&](const f32x4 (&a)[4], const f32x4 (&b)[NB][4], int s0, int s1) {
#pragma unroll
    for (int s = s0; s < s1; ++s)
#pragma unroll
      for (int cb = 0; cb < NB; ++cb)
        acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s >> 2][s & 3], b[cb][s >> 2][s & 3], acc[cb], 0, 0, 0);
  };

  if (DIAG) d_t1 = __builtin_amdgcn_s_memtime();
  if (BD && T > 0) {
    // ---- weights straight to registers, gathered rows through the DA-deep LDS ring.  Every vector-memory operation
    // of the walk is counted by hand: the weight loads are inline asm (beside LDS-DMA pieces in flight hipcc waits
    // vmcnt(0) for any load it can see, which would drain the ring every step), nothing is issued for steps that do
    // not exist (no dummy pieces in the tail).  Iteration t issues B(t+1) then A(t+DA) and needs A(t+1), B(t) at its
    // top: the pieces allowed to be in flight there are A(t+2) .. A(t+DA-1).
    Gen gA, gB;
    gen_init(gA);
    gen_init(gB);
    int32_t nb[4];
    int aS = 0, arS = 0;                     // ring slot of the next A issue / of the next fragment read
    const uint32_t b_voff = (uint32_t)(r31 * Cin + half * 16) * 4u;
    auto loadB = [&](const Gen& g, f32x4 (&b)[NB][4]) {
      const int kk = flip ? K - 1 - g.k : g.k;
      const uint64_t bp = reinterpret_cast<uint64_t>(Wb) + (uint64_t)((((int64_t)kk * Cout + col0) * Cin + g.c * 32) * 4);
      const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)bp);
      const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(bp >> 32));
      const char* base = reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
#pragma unroll
      for (int cb = 0; cb < NB; ++cb) {
        const uint32_t voff = b_voff + (uint32_t)(cb * 32 * Cin * 4);
        // s_nop 4: the SGPR base comes fresh from v_readfirstlane (5 wait states before a VMEM reads it; hipcc pads
        // nothing around an asm statement)
        asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=v"(b[cb][0]) : "v"(voff), "s"(base) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:16" : "=v"(b[cb][1]) : "v"(voff), "s"(base) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:32" : "=v"(b[cb][2]) : "v"(voff), "s"(base) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:48" : "=v"(b[cb][3]) : "v"(voff), "s"(base) : "memory");
      }
    };
    // the compiler must not touch the registers of an asm load before the counted wait that covers it
    auto tie = [&](f32x4 (&b)[NB][4]) {
#pragma unroll
      for (int cb = 0; cb < NB; ++cb)
        asm volatile("" : "+v"(b[cb][0]), "+v"(b[cb][1]), "+v"(b[cb][2]), "+v"(b[cb][3])::"memory");
    };
    auto issueA_all = [&]() {
      loadNb(gA, nb);
#pragma unroll
      for (int i = 0; i < 4; ++i) issueA1(gA, nb, aS, i);
      gen_next(gA);
      aS = aS + 1 == DA ? 0 : aS + 1;
    };
    // FB: relu(bn(x)) of the fragment just read -- lane (row r31, half) holds channels half*16 .. +15 of chunk g.c of
    // its row under offset g.k; a missing pair (zeros from the out-of-range DMA) must stay zero
    Gen gR;
    gen_init(gR);
    auto bnhalf = [&](f32x4 (&a)[4], const Gen& g, int h) {       // pieces q = 2h, 2h + 1
      const bool ok = nbT[g.k * 32 + r31] != (int32_t)NO_ROW;
      const float* cm = coef + g.c * 32 + half * 16;
#pragma unroll
      for (int q = 2 * h; q < 2 * h + 2; ++q) {
        const f32x4 mu = *reinterpret_cast<const f32x4*>(cm + q * 4);
        const f32x4 sc = *reinterpret_cast<const f32x4*>(cm + Cin + q * 4);
        const f32x4 bt = *reinterpret_cast<const f32x4*>(cm + 2 * Cin + q * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float z = __builtin_fmaf(a[q][e] - mu[e], sc[e], bt[e]);      // the arithmetic of bn_apply_kernel
          z = bin.relu ? fmaxf(z, 0.0f) : z;
          a[q][e] = ok ? z : 0.0f;
        }
      }
    };
    f32x4 a0[4], a1[4], b0[NB][4], b1[NB][4];
    issueA_all();                                   // A0
    loadB(gB, b0);                                  // B0
    gen_next(gB);
    if (T > 1) issueA_all();                        // A1
    if (DA >= 3 && T > 2) issueA_all();             // A2
    {   // A0, B0 landed; A1 [A2] may fly
      const int young = (T > DA ? DA : T) - 1;
      if (young >= 2)
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (young == 1)
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    tie(b0);
    readfrag(0, 0, a0, b0);
    if (FB) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (the coefficient stores of this wave)
      bnhalf(a0, gR, 0);
      bnhalf(a0, gR, 1);
      gen_next(gR);
    }
    arS = 1 == DA ? 0 : 1;
    int t = 0;
    auto iter = [&](const f32x4 (&ac)[4], f32x4 (&bc)[NB][4], f32x4 (&an)[4], f32x4 (&bn)[NB][4]) {
      // top: A(t+1) and B(t) landed, the fragment reads of step t are back; A(t+2) (DA = 3) may still fly
      if (DA >= 3 && t + 2 < T)
        asm volatile("s_waitcnt vmcnt(4)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
      else
        asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
      tie(bc);
      __builtin_amdgcn_sched_barrier(0);
      mfma(ac, bc, 0, 4);
      __builtin_amdgcn_sched_barrier(0);
      if (t + 1 < T) {
        readfrag(arS, 0, an, bn);
        arS = arS + 1 == DA ? 0 : arS + 1;
        if (!dbg_noB) loadB(gB, bn);
        gen_next(gB);
      }
      const bool more = t + DA < T;
      if (more) loadNb(gA, nb);
      __builtin_amdgcn_sched_barrier(0);
      mfma(ac, bc, 4, 8);
      __builtin_amdgcn_sched_barrier(0);
      // ONE MFMA chain whatever the tail does (MFMAs duplicated into both arms of the branch made hipcc keep the
      // accumulator in two register ranges and copy it every iteration)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (more) issueA1(gA, nb, aS, i);
        __builtin_amdgcn_sched_barrier(0);
        mfma(ac, bc, 8 + i, 9 + i);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (more) {
        gen_next(gA);
        aS = aS + 1 == DA ? 0 : aS + 1;
      }
      if (FB) {
        // the next step's fragments (read behind MFMA 3) through the BatchNorm, in two halves inside the tail of the
        // chain: the VALU work runs in the shadow of this wave's own MFMAs and of the other waves of the SIMD
        const bool nxt = t + 1 < T;
        if (nxt) bnhalf(an, gR, 0);
        __builtin_amdgcn_sched_barrier(0);
        mfma(ac, bc, 12, 14);
        __builtin_amdgcn_sched_barrier(0);
        if (nxt) {
          bnhalf(an, gR, 1);
          gen_next(gR);
        }
        __builtin_amdgcn_sched_barrier(0);
        mfma(ac, bc, 14, 16);
      } else {
        mfma(ac, bc, 12, 16);
      }
      ++t;
    };
    while (t < T) {
      iter(a0, b0, a1, b1);
      if (t < T) iter(a1, b1, a0, b0);
    }
  } else if (T > 0) {
    // issue order (DA = 3): A0 B0 A1 B1 A2 | B2 A3 | B3 A4 | ...   iteration t issues B(t+2), A(t+DA);
    // at the top of iteration t the pieces younger than B(t+1) are A(t+2) .. A(t+DA-1): 4 (DA - 2) of them.
    // BD: A0 B0 A1 [A2] | B1 A(DA) | B2 A(DA+1) ...  iteration t loads the registers of B(t+1) and issues A(t+DA);
    // the same count holds (the compiler adds its own wait for the B registers where they are first used).
    constexpr int VM_TOP = 4 * (DA - 2);
    constexpr int VM_PRE = BD ? 4 * (DA - 1) : 4 + 4 * NB + 4 * (DA - 2);     // A0, B0 landed
    Gen gA, gB;
    gen_init(gA);
    gen_init(gB);
    int32_t nb[4];
    int aS = 0, bS = 0;          // ring slots the NEXT issue goes to
    auto issueA = [&]() {
      loadNb(gA, nb);
#pragma unroll
      for (int i = 0; i < 4; ++i) issueA1(gA, nb, aS, i);
      gen_next(gA);
      aS = aS + 1 == DA ? 0 : aS + 1;
    };
    auto issueB = [&]() {
#pragma unroll
      for (int i = 0; i < 4 * NB; ++i) issueB1(gB, bS, i);
      gen_next(gB);
      bS ^= 1;
    };
    f32x4 a0[4], a1[4], b0[NB][4], b1[NB][4];
    issueA();
    if (BD) {
      loadB(gB, b0);
      gen_next(gB);
      issueA();
    } else {
      issueB();
      issueA();
      issueB();
    }
    if (DA >= 3) issueA();
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM_PRE) : "memory");
    readfrag(0, 0, a0, b0);
    int arS = 1, brS = 1;        // ring slots the NEXT fragment read takes
    // one iteration: fragments of step s+1 <- LDS, DMA of B(s+2) and A(s+DA), MFMAs of step s -- interleaved by hand
    // (left alone hipcc puts all 16 MFMAs behind the whole issue block and waits lgkmcnt(0) after every table read)
    auto iter = [&](const f32x4 (&ac)[4], const f32x4 (&bc)[NB][4], f32x4 (&an)[4], f32x4 (&bn)[NB][4]) {
      // A(s+1), B(s+1) landed; the fragment reads of step s are back
      asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)" ::"n"(VM_TOP) : "memory");
      // the first MFMAs go ahead of the new LDS reads: hipcc cannot see that the asm wait above covered the current
      // fragments and puts its own lgkmcnt(0) in front of the first MFMA -- with nothing pending there it is free
      __builtin_amdgcn_sched_barrier(0);
      mfma(ac, bc, 0, 4);
      __builtin_amdgcn_sched_barrier(0);
      readfrag(arS, brS, an, bn);
      arS = arS + 1 == DA ? 0 : arS + 1;
      brS ^= 1;
      loadNb(gA, nb);
      __builtin_amdgcn_sched_barrier(0);
      if (BD) {
        loadB(gB, bn);
        __builtin_amdgcn_sched_barrier(0);
        mfma(ac, bc, 4, 8);
        __builtin_amdgcn_sched_barrier(0);
      } else {
#pragma unroll
        for (int i = 0; i < 4 * NB; ++i) {
          issueB1(gB, bS, i);
          if (i % NB == NB - 1) {
            __builtin_amdgcn_sched_barrier(0);
            mfma(ac, bc, 4 + i / NB, 5 + i / NB);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        bS ^= 1;
      }
      gen_next(gB);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        issueA1(gA, nb, aS, i);
        __builtin_amdgcn_sched_barrier(0);
        mfma(ac, bc, 8 + i, 9 + i);
        __builtin_amdgcn_sched_barrier(0);
      }
      gen_next(gA);
      aS = aS + 1 == DA ? 0 : aS + 1;
      mfma(ac, bc, 12, 16);
    };
    for (int s = 0; s < T; s += 2) {
      iter(a0, b0, a1, b1);
      if (s + 1 < T) iter(a1, b1, a0, b0);
    }
  }
  // every DMA (the dummies of the tail included) must have landed before this wave's LDS is reused or released
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  if (DIAG) d_t2 = __builtin_amdgcn_s_memtime();

  // ---- epilogue.  C/D map of the 32x32 tile: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * half
  const bool final_pass = gridDim.z == 1;
  float* const dst = final_pass ? out : partial + (int64_t)blockIdx.z * M_out * Cout;
  if (!final_pass) {
    bias = nullptr;
    residual = nullptr;
  }
  if (NW == 1) {
    // the residual test is hoisted over the whole tile and the values are pinned before the (row-masked) stores: with
    // a per-element "load or zero" select hipcc branches around every load and waits vmcnt(0) in every store block
    auto store_tile = [&](auto has_res) {
#pragma unroll
      for (int cb = 0; cb < NB; ++cb) {
        const int c = col0 + cb * 32 + r31;
        const float bv = bias ? bias[c] : 0.0f;
        int32_t rows[16];
        float val[16];
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) rows[reg] = rowId[(reg & 3) + 8 * (reg >> 2) + 4 * half];
        if (decltype(has_res)::value) {
          float rv[16];
#pragma unroll
          for (int reg = 0; reg < 16; ++reg)    // 16 loads in flight together (row 0 for the masked ones)
            rv[reg] = residual[(int64_t)(rows[reg] >= 0 ? rows[reg] : 0) * Cout + c];
#pragma unroll
          for (int reg = 0; reg < 16; ++reg) val[reg] = (acc[cb][reg] + bv) + rv[reg];
        } else {
#pragma unroll
          for (int reg = 0; reg < 16; ++reg) val[reg] = acc[cb][reg] + bv;
        }
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) asm volatile("" : "+v"(val[reg]));
#pragma unroll
        for (int reg = 0; reg < 16; ++reg)
          if (rows[reg] >= 0) dst[(int64_t)rows[reg] * Cout + c] = val[reg];
        if (stats && final_pass && epi.x) {     // BatchNorm-backward partials of the slice (see BnEpi)
          const BnCoef kc = bn_coef(epi, c);
          float xv[16];
#pragma unroll
          for (int reg = 0; reg < 16; ++reg) xv[reg] = epi.x[(int64_t)(rows[reg] >= 0 ? rows[reg] : 0) * Cout + c];
          float sa = 0.0f, sb = 0.0f;
#pragma unroll
          for (int reg = 0; reg < 16; ++reg) {
            float dz, dzx;
            bn_terms(kc, epi.relu, val[reg], xv[reg], dz, dzx);
            sa += rows[reg] >= 0 ? dz : 0.0f;
            sb += rows[reg] >= 0 ? dzx : 0.0f;
          }
          sa += __shfl_xor(sa, 32, 64);
          sb += __shfl_xor(sb, 32, 64);
          if (half == 0) {
            stats[((int64_t)blockIdx.x * 2 + 0) * Cout + c] = sa;
            stats[((int64_t)blockIdx.x * 2 + 1) * Cout + c] = sb;
          }
        } else if (stats && final_pass) {      // rows in register order, then the two halves: a fixed order
          float sa = 0.0f, sb = 0.0f;
#pragma unroll
          for (int reg = 0; reg < 16; ++reg) sa += rows[reg] >= 0 ? val[reg] : 0.0f;
          sa += __shfl_xor(sa, 32, 64);
          const int64_t left = M_out - t0;
          const float mean_s = sa / (float)(left < SL ? left : SL);     // rows of the slice (tile order: a prefix)
#pragma unroll
          for (int reg = 0; reg < 16; ++reg) {
            const float d = rows[reg] >= 0 ? val[reg] - mean_s : 0.0f;
            sb += d * d;
          }
          sb += __shfl_xor(sb, 32, 64);
          if (half == 0) {
            st_sc1(stats + ((int64_t)blockIdx.x * 2 + 0) * Cout + c, sa);
            st_sc1(stats + ((int64_t)blockIdx.x * 2 + 1) * Cout + c, sb);
          }
          if (NB == 1 && fin.chunk)
            stat_finish(fin, stats, M_out, Cout, (int64_t)blockIdx.x, col0, (int)gridDim.y, (int)blockIdx.y,
                        reinterpret_cast<double*>(Aring));
        }
      }
    };
    if (residual)
      store_tile(std::true_type{});
    else
      store_tile(std::false_type{});
  } else {
    // every global read of the epilogue is issued first (row 0 for the masked rows): they fly while the waves write
    // their accumulators and wait for the slowest of them; inside the element loop each would expose its latency
    const int32_t* rowId0 = rowId;
    constexpr int PER = (32 * NB * 32) / (64 * NW);      // elements of this thread (its column is fixed: 64 NW is a
    float keep[PER];                                     // multiple of NB * 32)
    bool live[PER];
    float sa = 0.0f, sq = 0.0f;
    const bool bn_mode = stats && final_pass && epi.x;
    BnCoef kc = {0.0f, 0.0f, 0.0f, 0.0f};
    const int cc = (int)(threadIdx.x % (NB * 32)), c = col0 + cc;
    if (bn_mode) kc = bn_coef(epi, c);
    const float bv = bias ? bias[c] : 0.0f;
    int64_t off[PER];
    float rv[PER], xv[PER];
#pragma unroll
    for (int it = 0; it < PER; ++it) {
      const int rr = (threadIdx.x + it * 64 * NW) / (NB * 32);
      const int32_t r = rowId0[rr];
      live[it] = r >= 0;
      off[it] = (int64_t)(r >= 0 ? r : 0) * Cout + c;
    }
#pragma unroll
    for (int it = 0; it < PER; ++it) rv[it] = residual ? residual[off[it]] : 0.0f;
#pragma unroll
    for (int it = 0; it < PER; ++it) xv[it] = bn_mode ? epi.x[off[it]] : 0.0f;
    // accumulators -> this wave's ring memory as [row][NB*32 cols]; then every thread adds the NW copies in wave order
    float* red = reinterpret_cast<float*>(Aring);
#pragma unroll
    for (int cb = 0; cb < NB; ++cb)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int rr = (reg & 3) + 8 * (reg >> 2) + 4 * half;
        red[rr * (NB * 32) + cb * 32 + r31] = acc[cb][reg];
      }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < PER; ++it) {
      const int e = threadIdx.x + it * 64 * NW;
      float v = 0.0f;
#pragma unroll
      for (int w = 0; w < NW; ++w)
        v += reinterpret_cast<const float*>(lds + HDR_BYTES + w * L::WAVE_BYTES)[e];
      if (bias) v += bv;
      if (residual) v += rv[it];
      keep[it] = live[it] ? v : 0.0f;
      if (live[it]) dst[off[it]] = v;
      if (bn_mode) {
        float dz, dzx;
        bn_terms(kc, epi.relu, v, xv[it], dz, dzx);
        sa += live[it] ? dz : 0.0f;
        sq += live[it] ? dzx : 0.0f;
      } else {
        sa += live[it] ? v : 0.0f;
      }
    }
    if (bn_mode) {                  // both sums are plain: one exchange
      __syncthreads();
      float* sred = reinterpret_cast<float*>(lds + HDR_BYTES);
      const int colw = NB * 32;
      sred[threadIdx.x] = sa;
      sred[64 * NW + threadIdx.x] = sq;
      __syncthreads();
      if (threadIdx.x < colw) {
        float ta = 0.0f, tb = 0.0f;
        for (int j = threadIdx.x; j < 64 * NW; j += colw) {
          ta += sred[j];
          tb += sred[64 * NW + j];
        }
        const int c = col0 + threadIdx.x;
        stats[((int64_t)blockIdx.x * 2 + 0) * Cout + c] = ta;
        stats[((int64_t)blockIdx.x * 2 + 1) * Cout + c] = tb;
      }
    } else if (stats && final_pass) {      // threads of one column: t, t + NB*32, ...; added in that order
      __syncthreads();              // the accumulator copies in the rings are no longer needed
      float* sred = reinterpret_cast<float*>(lds + HDR_BYTES);
      const int colw = NB * 32, me = threadIdx.x % colw;
      sred[threadIdx.x] = sa;
      __syncthreads();
      float ta = 0.0f;
      for (int j = me; j < 64 * NW; j += colw) ta += sred[j];         // every thread: its column's sum
      const int64_t left = M_out - t0;
      const float mean_s = ta / (float)(left < SL ? left : SL);
      float sb = 0.0f;
#pragma unroll
      for (int it = 0; it < PER; ++it) {
        const float d = live[it] ? keep[it] - mean_s : 0.0f;
        sb += d * d;
      }
      __syncthreads();
      sred[threadIdx.x] = sb;
      __syncthreads();
      if (threadIdx.x < colw) {
        float tb = 0.0f;
        for (int j = threadIdx.x; j < 64 * NW; j += colw) tb += sred[j];
        const int c = col0 + threadIdx.x;
        st_sc1(stats + ((int64_t)blockIdx.x * 2 + 0) * Cout + c, ta);
        st_sc1(stats + ((int64_t)blockIdx.x * 2 + 1) * Cout + c, tb);
      }
      if (NB == 1 && fin.chunk && wave == 0)      // (the partials of a 32-channel block are stored by lanes 0-31 of wave 0)
        stat_finish(fin, stats, M_out, Cout, (int64_t)blockIdx.x, col0, (int)gridDim.y, (int)blockIdx.y,
                    reinterpret_cast<double*>(lds + HDR_BYTES));
    }
  }
  if (DIAG && dbg && threadIdx.x == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t3 = __builtin_amdgcn_s_memtime();
    unsigned long long* d = dbg + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 8;
    d[0] = d_r0;
    d[1] = __builtin_amdgcn_s_memrealtime();
    d[2] = d_t1 - d_t0;
    d[3] = d_t2 - d_t1;
    d[4] = t3 - d_t2;
    d[5] = (unsigned long long)T;
    d[6] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);   // HW_REG_HW_ID, 32 bits
    d[7] = 0;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Persistent form of the multi-wave work item (the mid pyramid levels: enough slices for a few per workgroup, too few
// for one wave per slice).  spconv_fwd2_kernel pays per slice a workgroup launch, two dependent memory latencies (gather
// table, then the first gathered rows: ~5.8 us) and an epilogue, and all resident workgroups go through these phases
// together, so the MFMA pipe idles for more than half of the launch.  Here a workgroup of NW waves stays resident and
// walks slices blockIdx.x, + gridDim.x, ...; every wave keeps the offsets fwd2 gives it (k % ZS == z,
// (k / ZS) % NW == wave: identical order of additions, bit-identical output) and runs its own generator one step
// ahead of its MFMA chain ACROSS slice boundaries: header of the next slice (its <= 8 table rows + the row list) by
// LDS-DMA one slice ahead, gathered rows of the next step by buffer LDS-DMA (32-bit offsets, missing pair = out of
// range = zeros), weights of the next step straight to registers; every wait is a counted vmcnt.  Per slice the waves
// meet twice (accumulators -> LDS in the ring slot just consumed, sum in wave order, store / statistics).
constexpr int P3_GS = 8;                               // offsets per wave
// slice queue of a launch: one ticket counter per (output block, offset slab) in the caller's sync slot (SyncSlot::ctr;
// zero between launches: the workgroup that draws the last ticket of a launch resets it)
constexpr int P3_HDR_INTS = P3_GS * 32 + 64;           // nb[8][32] + rows[32] (+ 32 written by the upper half wave)
constexpr int P3_HDR = P3_HDR_INTS * 4;
constexpr int P3_WAVE = 2 * P3_HDR + 2 * A_BYTES;      // two headers, two-slot ring of gathered rows
constexpr int P3_WG = 64;                              // ring parities of the waves

template <int NW, bool DIAG>
__global__ __launch_bounds__(64 * NW, 12 / NW) void spconv_fwd3_kernel(
    const float* __restrict__ X, const int32_t* __restrict__ nbrS, const int32_t* __restrict__ order,
    const float* __restrict__ WT, const float* __restrict__ bias, const float* __restrict__ residual,
    float* __restrict__ out, float* __restrict__ partial, int64_t M_out, int K, int Cin, int Cout, int flip,
    uint32_t x_bytes, float* __restrict__ stats, BnEpi epi, unsigned* __restrict__ q_ctr,
    unsigned long long* __restrict__ dbg = nullptr) {
  unsigned long long d_t0 = 0, d_pro = 0, d_steps = 0, d_wait = 0, d_epi = 0, d_tmp = 0;
  unsigned d_nsteps = 0, d_nsl = 0;
  if (DIAG) d_t0 = __builtin_readcyclecounter();
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r31 = lane & 31, half = lane >> 5;
  int32_t* const par = reinterpret_cast<int32_t*>(lds);
  unsigned char* const my = lds + P3_WG + wave * P3_WAVE;
  int32_t* const hdr0 = reinterpret_cast<int32_t*>(my);
  unsigned char* const At = my + 2 * P3_HDR;

  const int ZS = gridDim.z, z = blockIdx.z;
  const int col0 = blockIdx.y * 32;
  const int nchunk = Cin >> 5;
  const int64_t n_slices = (M_out + 31) >> 5;
  // ---- slice queue: the workgroups of a (block, slab) draw slices from a ticket counter in tile order (heaviest
  // first), so they finish together whatever the spread of steps per slice (static strided dealing: steps per wave
  // 8 ... 30 around a mean of 12 at level 1 of the C2 scene).  A slice's result does not depend on who computes it, so
  // the output stays bit-reproducible.  Inside the workgroup the wave that first needs the i-th slice id draws it and
  // publishes it in LDS; the others read it (they all walk the same sequence).
  unsigned* const ctr = q_ctr + (blockIdx.z * gridDim.y + blockIdx.y);
  volatile int* const q_pub = reinterpret_cast<volatile int*>(lds + 16);
  int* const q_claim = reinterpret_cast<int*>(lds + 20);
  volatile int* const q_val = reinterpret_cast<volatile int*>(lds + 24);      // ring of 8
  if (threadIdx.x == 0) {
    *q_pub = 2;
    *q_claim = 2;
  }
  __syncthreads();
  // the first two slices of a workgroup are dealt statically (blockIdx.x, + gridDim.x): 2 x gridDim.x draws of one
  // counter at kernel start took 22 us (device-scope atomics on one address across 8 XCDs); from the third on the
  // draws are spread over the launch.  Ticket t = slice 2 P + t.
  const int64_t q_P = gridDim.x;
  const int64_t q_real = n_slices > 2 * q_P ? n_slices - 2 * q_P : 0;                 // slices handed out by tickets
  const int64_t q_draw = n_slices > q_P ? (n_slices - q_P < q_P ? n_slices - q_P : q_P) : 0;   // workgroups that draw
  const unsigned q_last = (unsigned)(q_real + q_draw - 1);
  auto q_get = [&](int i) -> int64_t {
    if (i < 2) {
      const int64_t sl = (int64_t)blockIdx.x + i * q_P;
      return sl < n_slices ? sl : n_slices;
    }
    for (;;) {
      if (__builtin_amdgcn_readfirstlane(*q_pub) > i) break;
      int won = 0;
      if (lane == 0) won = atomicCAS(q_claim, i, i + 1) == i;
      won = __builtin_amdgcn_readfirstlane(won);
      if (won) {
        unsigned t = 0;
        if (lane == 0) {
          t = atomicAdd(ctr, 1u);
          if (t == q_last) *ctr = 0u;                 // the last draw of the launch: ready for the next one
          const int64_t sl = 2 * q_P + (int64_t)t;
          q_val[i & 7] = (int)(sl < n_slices ? sl : n_slices);
          __threadfence_block();
          *q_pub = i + 1;
        }
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
    for (;;) {       // (the winner's own publication included)
      if (__builtin_amdgcn_readfirstlane(*q_pub) > i) break;
      __builtin_amdgcn_s_sleep(1);
    }
    return (int64_t)__builtin_amdgcn_readfirstlane(q_val[i & 7]);
  };
  // offset slot j of this wave: k = z + ZS * (wave + NW * j)
  const int k0 = z + ZS * wave, kstep = ZS * NW;

  const uint32_t a_pitch = (uint32_t)Cin * 4u;
  const rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X), (short)0, (int)x_bytes, 0x00020000);
  const rsrc_t rsN = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(nbrS), (short)0,
                                                       (int)((int64_t)K * M_out * 4), 0x00020000);
  const rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(order), (short)0, (int)(M_out * 4), 0x00020000);
  const int d_row = lane >> 3, d_piece = lane & 7;
  uint32_t a_po[4];                      // swizzled 16-byte piece of this lane in DMA instruction i
#pragma unroll
  for (int i = 0; i < 4; ++i) a_po[i] = (uint32_t)((d_piece ^ swz(i * 8 + d_row)) << 4);
  const uint32_t m_last = (uint32_t)(M_out - 1);
  const char* const Wb = reinterpret_cast<const char*>(WT);
  const uint32_t b_voff = (uint32_t)(r31 * Cin + half * 16) * 4u;

  auto issueH = [&](int64_t s, int32_t* hb) {          // 5 DMA instructions
    uint32_t t = (uint32_t)s * 32u + (uint32_t)r31;
    t = t < m_last ? t : m_last;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      int k = k0 + (2 * q + half) * kstep;
      k = k < K ? k : k0 % K;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsN, (__attribute__((address_space(3))) void*)(hb + q * 64), 4,
                                               (int)(((uint32_t)k * (uint32_t)M_out + t) * 4u), 0, 0, 0);
    }
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsO, (__attribute__((address_space(3))) void*)(hb + P3_GS * 32), 4,
                                             (int)(t * 4u), 0, 0, 0);
  };
  auto fix_tail = [&](int32_t* hb, int64_t s) {        // the last slice only: rows past M_out are missing
    if (s * 32 + 32 <= M_out) return;
    const int64_t t0 = s * 32 + (lane & 7) * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (t0 + e >= M_out) hb[lane * 4 + e] = -1;
    if (lane < 32 && s * 32 + lane >= M_out) hb[P3_GS * 32 + lane] = -1;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };
  auto readmask = [&](const int32_t* hb) -> uint32_t {
    const int4 v = *reinterpret_cast<const int4*>(hb + lane * 4);      // slot lane >> 3, rows (lane & 7) * 4 ..
    const bool any = (k0 + (lane >> 3) * kstep < K) & ((v.x & v.y & v.z & v.w) >= 0);
    unsigned long long b = __ballot(any);
    b |= b >> 4;
    b |= b >> 2;
    b |= b >> 1;
    uint32_t m = 0u;
#pragma unroll
    for (int j = 0; j < P3_GS; ++j) m |= (uint32_t)((b >> (8 * j)) & 1ull) << j;
    return m;
  };
  // gathered rows of (offset slot j, chunk c) -> ring slot; 4 DMA instructions
  auto issueA = [&](const int32_t* hb, int j, int c, unsigned char* dst) {
    const int32_t* p = hb + j * 32 + d_row;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      bdma16(rsX, (uint32_t)p[i * 8] * a_pitch + a_po[i], (uint32_t)c * 128u, dst + i * 1024);
  };
  // weights of (offset slot j, chunk c): the lane's 16 values of output column col0 + r31; 4 plain loads.  (Every
  // step top waits vmcnt(0) anyway -- one step of lookahead -- so the compiler's own wait at the first use costs
  // nothing; inline-asm loads with the wait "by hand" let hipcc copy the destination registers at control-flow merges
  // before the data had landed.)
  auto loadB = [&](int j, int c, f32x4 (&b)[4]) {
    const int k = k0 + j * kstep;
    const int kk = flip ? K - 1 - k : k;
    const char* base = Wb + (((int64_t)kk * Cout + col0) * Cin + c * 32) * 4 + b_voff;
#pragma unroll
    for (int q = 0; q < 4; ++q) b[q] = *reinterpret_cast<const f32x4*>(base + q * 16);
  };
  auto tie = [&](f32x4 (&b)[4]) {};
  auto readfragA = [&](const unsigned char* img, f32x4 (&a)[4]) {
    const unsigned char* arow = img + r31 * 128;
    const int sw = swz(r31);
#pragma unroll
    for (int q = 0; q < 4; ++q) a[q] = *reinterpret_cast<const f32x4*>(arow + (((half * 4 + q) ^ sw) << 4));
  };

  f32x16 acc;
  const bool final_pass = ZS == 1;
  float* const dst = final_pass ? out : partial + (int64_t)z * M_out * Cout;
  if (!final_pass) {
    bias = nullptr;
    residual = nullptr;
  }

  // ---- generator state (wave-uniform): position (slice, remaining offset slots, slot, chunk) of the NEXT step to issue
  int64_t g_slice = q_get(0);
  if (g_slice >= n_slices) return;       // (uniform: every wave reads the same id)
  const int64_t first_slice = g_slice;   // (the generator may enter the next slice before the compute loop starts)
  int64_t g_nxt = q_get(1);              // the slice after it: its header is prefetched
  int64_t slice_next = 0;
  int g_hb = 0;
  uint32_t g_rem = 0u;                   // offset slots of g_slice after g_j
  int g_j = 0, g_c = 0;
  bool g_live = false;                   // (g_j, g_c) is a step not yet issued
  bool g_has_next = g_nxt < n_slices;
  int gi = 0, ci = 0;                    // slices entered by the generator / the compute side
  uint32_t mask_next = 0u;
  int F = 0;                             // steps issued and not consumed (0 / 1)
  int c_rd = 0;                          // ring slot of the next step to consume
  bool parity = false;                   // which weight register set the next step to consume uses
  f32x4 bA[4], bB[4], afr[4];

  // first header, synchronously; the second one in flight
  issueH(g_slice, hdr0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  fix_tail(hdr0, g_slice);
  uint32_t c_mask = readmask(hdr0);
  bool h_pending = false;                // a header DMA issued and no vmcnt(0) since
  if (g_has_next) {
    issueH(g_nxt, hdr0 + P3_HDR_INTS);
    h_pending = true;
  }
  g_rem = c_mask;
  if (g_rem) {
    g_j = __builtin_ctz(g_rem);
    g_rem &= g_rem - 1u;
    g_c = 0;
    g_live = true;
  }
  // one advance of the generator: everything of the next step (and, at a slice change, header / bookkeeping of the
  // slice behind it).  Returns false when there is nothing it may issue now.
  auto advance = [&](f32x4 (&bdst)[4], int slot, int& n) -> bool {
    n = 8;
    if (!g_live) {
      // enter the next slice?  only once the compute side is in the slice before it (one spare header)
      if (gi != ci || !g_has_next) return false;
      // the next header was issued one slice ago; a wave without any step since then has not waited for it yet
      if (h_pending) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      h_pending = false;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      int32_t* const hb = hdr0 + g_hb * P3_HDR_INTS;
      int32_t* const hn = hdr0 + (g_hb ^ 1) * P3_HDR_INTS;
      g_slice = g_nxt;
      slice_next = g_slice;
      fix_tail(hn, g_slice);
      const uint32_t m = readmask(hn);
      g_hb ^= 1;
      ++gi;
      mask_next = m;
      g_nxt = q_get(gi + 1);
      g_has_next = g_nxt < n_slices;
      int nh = 0;
      if (g_has_next) {
        issueH(g_nxt, hb);                       // into the header just left (its rows are in registers: my_rows)
        h_pending = true;
        nh = 5;
      }
      g_rem = m;
      if (!g_rem) return false;                  // no pair for this wave in that slice: nothing but the header
      g_j = __builtin_ctz(g_rem);
      g_rem &= g_rem - 1u;
      g_c = 0;
      g_live = true;
      n = 8 + nh;
    }
    const int32_t* hbq = hdr0 + g_hb * P3_HDR_INTS;
    loadB(g_j, g_c, bdst);
    issueA(hbq, g_j, g_c, At + slot * A_BYTES);
    if (++g_c == nchunk) {
      g_c = 0;
      if (g_rem) {
        g_j = __builtin_ctz(g_rem);
        g_rem &= g_rem - 1u;
      } else {
        g_live = false;
      }
    }
    return true;
  };

  const int PER = (32 * 32) / (64 * NW);
  int32_t my_rows[(32 * 32) / (64 * NW)];
  auto load_rows = [&](const int32_t* hb) {
#pragma unroll
    for (int it = 0; it < PER; ++it) my_rows[it] = hb[P3_GS * 32 + (int)(threadIdx.x + it * 64 * NW) / 32];
  };
  load_rows(hdr0);
  int c_hb = 0;

  // first step of the wave (if its first slice has one)
  {
    int n;
    if (advance(bA, 0, n)) F = 1;
  }
  // one step on weight set `bc`, prefetching into `bn`
  auto step = [&](f32x4 (&bc)[4], f32x4 (&bn)[4]) {
    // the step's gathered rows and weights are the youngest loads in flight (one step of lookahead)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    h_pending = false;
    tie(bc);
    readfragA(At + c_rd * A_BYTES, afr);
    int n = 0;
    // the next step's loads are issued from inside the chain
#define WSIS_M3(s_) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(afr[(s_) >> 2][(s_) & 3], bc[(s_) >> 2][(s_) & 3], acc, 0, 0, 0)
    __builtin_amdgcn_sched_barrier(0);
    WSIS_M3(0); WSIS_M3(1); WSIS_M3(2); WSIS_M3(3);
    __builtin_amdgcn_sched_barrier(0);
    const bool issued = advance(bn, c_rd ^ 1, n);
    __builtin_amdgcn_sched_barrier(0);
    WSIS_M3(4); WSIS_M3(5); WSIS_M3(6); WSIS_M3(7);
    WSIS_M3(8); WSIS_M3(9); WSIS_M3(10); WSIS_M3(11);
    WSIS_M3(12); WSIS_M3(13); WSIS_M3(14); WSIS_M3(15);
    __builtin_amdgcn_sched_barrier(0);
#undef WSIS_M3
    F = issued ? 1 : 0;
    c_rd ^= 1;
    parity = !parity;
  };

  if (DIAG) d_pro = __builtin_readcyclecounter() - d_t0;
  for (int64_t cs = first_slice;;) {
    if (DIAG) d_tmp = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    int T = __builtin_popcount(c_mask) * nchunk;
    while (T > 0) {
      if (F == 0) {        // nothing in flight for this step: issue it now (after an empty slice)
        int n;
        const bool ok = parity ? advance(bB, c_rd, n) : advance(bA, c_rd, n);
        (void)ok;
        F = 1;
      }
      if (!parity)
        step(bA, bB);
      else
        step(bB, bA);
      --T;
      if (DIAG) ++d_nsteps;
    }
    // the generator may still have to enter the next slice (this wave had no step in the current one, or its steps
    // ended before the advance could cross): the next slice's first step must be in flight before the epilogue
    if (F == 0) {
      int n;
      const bool ok = parity ? advance(bB, c_rd, n) : advance(bA, c_rd, n);
      if (ok) F = 1;
    }
    // everything this wave has in flight lands before the epilogue: from here to the next step only the epilogue's own
    // loads and stores are issued (they are older than anything the next chain issues, so no count has to know them)
    asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    h_pending = false;

    // ---- epilogue of slice cs.  C/D map: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * half
    const int64_t t0 = cs * 32;
    {
      // the epilogue's global reads first: they fly while the waves meet
      float keep[PER];
      bool live[PER];
      float sa = 0.0f, sq = 0.0f;
      const bool bn_mode = stats && final_pass && epi.x;
      BnCoef kc = {0.0f, 0.0f, 0.0f, 0.0f};
      const int cc = (int)(threadIdx.x & 31), c = col0 + cc;
      if (bn_mode) kc = bn_coef(epi, c);
      const float bv = bias ? bias[c] : 0.0f;
      int64_t off[PER];
      float rv[PER], xv[PER];
#pragma unroll
      for (int it = 0; it < PER; ++it) {
        const int32_t r = my_rows[it];
        live[it] = r >= 0;
        off[it] = (int64_t)(r >= 0 ? r : 0) * Cout + c;
      }
#pragma unroll
      for (int it = 0; it < PER; ++it) rv[it] = residual ? residual[off[it]] : 0.0f;
#pragma unroll
      for (int it = 0; it < PER; ++it) xv[it] = bn_mode ? epi.x[off[it]] : 0.0f;
      float* red = reinterpret_cast<float*>(At + (c_rd ^ 1) * A_BYTES);       // the ring slot just consumed
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int rr = (reg & 3) + 8 * (reg >> 2) + 4 * half;
        red[rr * 32 + r31] = acc[reg];
      }
      if (lane == 0) par[wave] = c_rd ^ 1;
      if (DIAG) { const unsigned long long t = __builtin_readcyclecounter(); d_steps += t - d_tmp; d_tmp = t; }
      __syncthreads();
      if (DIAG) { const unsigned long long t = __builtin_readcyclecounter(); d_wait += t - d_tmp; d_tmp = t; }
#pragma unroll
      for (int it = 0; it < PER; ++it) {
        const int e = threadIdx.x + it * 64 * NW;
        float v = 0.0f;
#pragma unroll
        for (int w = 0; w < NW; ++w)
          v += reinterpret_cast<const float*>(lds + P3_WG + w * P3_WAVE + 2 * P3_HDR + par[w] * A_BYTES)[e];
        if (bias) v += bv;
        if (residual) v += rv[it];
        keep[it] = live[it] ? v : 0.0f;
        if (live[it]) dst[off[it]] = v;
        if (bn_mode) {
          float dz, dzx;
          bn_terms(kc, epi.relu, v, xv[it], dz, dzx);
          sa += live[it] ? dz : 0.0f;
          sq += live[it] ? dzx : 0.0f;
        } else {
          sa += live[it] ? v : 0.0f;
        }
      }
      __syncthreads();                 // the accumulator copies are no longer needed
      if (stats && final_pass) {
        float* sred = reinterpret_cast<float*>(At + (c_rd ^ 1) * A_BYTES);      // this wave's free slot: 64 floats x 2
        float* sall = reinterpret_cast<float*>(lds + P3_WG);
        // per-wave staging at a fixed place in every wave's free slot
        sred[lane] = sa;
        sred[64 + lane] = sq;
        __syncthreads();
        // threads of one column: t, t + 32, ... (the order of spconv_fwd2_kernel: thread id ascending)
        auto col_sum = [&](int which) -> float {
          float tsum = 0.0f;
          for (int j = cc; j < 64 * NW; j += 32) {
            const int w = j >> 6, l = j & 63;
            tsum += reinterpret_cast<const float*>(lds + P3_WG + w * P3_WAVE + 2 * P3_HDR + par[w] * A_BYTES)[which * 64 + l];
          }
          return tsum;
        };
        (void)sall;
        if (bn_mode) {
          if (threadIdx.x < 32) {
            stats[((int64_t)cs * 2 + 0) * Cout + c] = col_sum(0);
            stats[((int64_t)cs * 2 + 1) * Cout + c] = col_sum(1);
          }
          __syncthreads();
        } else {
          const float ta = col_sum(0);
          const int64_t left = M_out - t0;
          const float mean_s = ta / (float)(left < SL ? left : SL);
          float sb = 0.0f;
#pragma unroll
          for (int it = 0; it < PER; ++it) {
            const float d = live[it] ? keep[it] - mean_s : 0.0f;
            sb += d * d;
          }
          __syncthreads();
          sred[64 + lane] = sb;
          __syncthreads();
          if (threadIdx.x < 32) {
            stats[((int64_t)cs * 2 + 0) * Cout + c] = ta;
            stats[((int64_t)cs * 2 + 1) * Cout + c] = col_sum(1);
          }
          __syncthreads();
        }
      }
    }
    if (DIAG) { d_epi += __builtin_readcyclecounter() - d_tmp; ++d_nsl; }
    // next slice of the compute side
    if (gi == ci) break;                 // the generator found no further slice: the queue is empty
    {
      cs = slice_next;
      c_hb ^= 1;
      ++ci;
      c_mask = mask_next;
      // (the generator entered it: gi == ci now, its header is current and intact until the generator leaves it)
      load_rows(hdr0 + c_hb * P3_HDR_INTS);
    }
  }
  if (DIAG && dbg && lane == 0) {
    unsigned long long* d = dbg + (((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * NW * 8 + wave * 8;
    d[0] = __builtin_readcyclecounter() - d_t0;
    d[1] = d_pro;
    d[2] = d_steps;
    d[3] = d_wait;
    d[4] = d_epi;
    d[5] = d_nsteps;
    d[6] = d_nsl;
    d[7] = 0;
  }
}

// out = sum_z partial[z] (+ bias, + residual), four channels per thread, z order fixed
__global__ void spconv2_reduce_kernel(const float4* __restrict__ partial, const float4* __restrict__ bias,
                                      const float4* __restrict__ residual, float4* __restrict__ out, int64_t total4,
                                      int cout4, int zs) {
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total4;
       t += (int64_t)gridDim.x * blockDim.x) {
    float4 s = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll 4
    for (int z = 0; z < zs; ++z) {
      const float4 v = partial[(int64_t)z * total4 + t];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if (bias) {
      const float4 v = bias[t % cout4];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if (residual) {
      const float4 v = residual[t];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    out[t] = s;
  }
}

int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}

// the same sum for levels whose consumer is a BatchNorm: one workgroup per 32 output rows (the slice granularity of
// the statistics partials), thread = (4 channels, row lane); also writes the slice's (sum, sum of squares) partials
__global__ __launch_bounds__(256) void spconv2_reduce_stats_kernel(const float4* __restrict__ partial,
                                                                  const float4* __restrict__ bias,
                                                                  const float4* __restrict__ residual,
                                                                  float4* __restrict__ out, int64_t M_out, int cout4,
                                                                  int zs, float* __restrict__ stats, BnEpi epi) {
  __shared__ float sred[256 * 4];
  const int64_t total4 = M_out * cout4;
  // gridDim.y > 1: the workgroup takes the 32 channels (8 float4 columns) of block blockIdx.y -- 32 row lanes, one row
  // per thread, five times the workgroups at 160 channels (11 workgroups walked 6 rows x 8 slabs per thread: 10 us)
  const int cw = gridDim.y > 1 ? 8 : cout4, cb = gridDim.y > 1 ? (int)blockIdx.y * 8 : 0;
  const int lanes = 256 / cw;                            // row lanes (cw <= 64)
  const int c4l = threadIdx.x % cw, c4 = cb + c4l, rl = threadIdx.x / cw;
  const int64_t r0 = (int64_t)blockIdx.x * 32;
  const int nrows = (int)min((int64_t)32, M_out - r0);
  constexpr int MAXR = 8;                                // rows per thread: ceil(32 / lanes), lanes >= 4
  float4 keep[MAXR];
  float sa[4] = {0.f, 0.f, 0.f, 0.f};
  BnCoef kcs[4] = {};
  if (epi.x)
#pragma unroll
    for (int e = 0; e < 4; ++e) kcs[e] = bn_coef(epi, c4 * 4 + e);
#pragma unroll
  for (int it = 0; it < MAXR; ++it) {
    keep[it] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const int rr = rl + it * lanes;
    if (rl < lanes && rr < nrows) {
      const int64_t t = (r0 + rr) * cout4 + c4;
      float4 s = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      for (int z0 = 0; z0 < zs; z0 += 8) {      // the slabs of a row in flight together (zs <= 8 in every plan), added in z order
        float4 pv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) pv[j] = partial[(int64_t)(z0 + j < zs ? z0 + j : z0) * total4 + t];
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (z0 + j < zs) {
            s.x += pv[j].x; s.y += pv[j].y; s.z += pv[j].z; s.w += pv[j].w;
          }
      }
      if (bias) {
        const float4 v = bias[c4];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
      if (residual) {
        const float4 v = residual[t];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
      out[t] = s;
      keep[it] = s;
      if (epi.x) {      // BatchNorm-backward partials: keep <- dz * xhat, sa <- dz
        const float4 xv = reinterpret_cast<const float4*>(epi.x)[t];
        const float dyv[4] = {s.x, s.y, s.z, s.w}, xs[4] = {xv.x, xv.y, xv.z, xv.w};
        float q[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float dz;
          bn_terms(kcs[e], epi.relu, dyv[e], xs[e], dz, q[e]);
          sa[e] += dz;
        }
        keep[it] = make_float4(q[0], q[1], q[2], q[3]);
      } else {
        sa[0] += s.x; sa[1] += s.y; sa[2] += s.z; sa[3] += s.w;
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) sred[threadIdx.x * 4 + e] = sa[e];
  __syncthreads();
  float ta[4] = {0.f, 0.f, 0.f, 0.f};
  if (rl < lanes)
    for (int j = 0; j < lanes; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) ta[e] += sred[(j * cw + c4l) * 4 + e];
  float sb[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int it = 0; it < MAXR; ++it) {
    const int rr = rl + it * lanes;
    if (rl < lanes && rr < nrows) {
      if (epi.x) {
        sb[0] += keep[it].x; sb[1] += keep[it].y; sb[2] += keep[it].z; sb[3] += keep[it].w;
      } else {
        const float d0 = keep[it].x - ta[0] / (float)nrows, d1 = keep[it].y - ta[1] / (float)nrows;
        const float d2 = keep[it].z - ta[2] / (float)nrows, d3 = keep[it].w - ta[3] / (float)nrows;
        sb[0] += d0 * d0; sb[1] += d1 * d1; sb[2] += d2 * d2; sb[3] += d3 * d3;
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 4; ++e) sred[threadIdx.x * 4 + e] = sb[e];
  __syncthreads();
  if (threadIdx.x < cw) {
    float tb[4] = {0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < lanes; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) tb[e] += sred[(j * cw + threadIdx.x) * 4 + e];
    const int Cout = cout4 * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      stats[((int64_t)blockIdx.x * 2 + 0) * Cout + (cb + threadIdx.x) * 4 + e] = ta[e];
      stats[((int64_t)blockIdx.x * 2 + 1) * Cout + (cb + threadIdx.x) * 4 + e] = tb[e];
    }
  }
}

// NB: output blocks per work item; NW: waves per work item; ZS: offset groups over blockIdx.z (partial slabs);
// DA: depth of the gathered-row ring.  Aim for ~2 waves per SIMD over the whole launch.
struct Plan2 {
  int NB, NW, ZS, DA, BD;
};

// noslab: the product is finished by ONE launch whatever the level -- up to 16 waves of a workgroup split the offsets of
// a work item and add through LDS, no offset slabs and no second (reduce) launch; the form of the fused-BatchNorm
// convolutions (wsis_spconv_fwd_f) and, with WSIS_FWD2_NOSLAB=1 (default), of every launch
Plan2 plan2(int64_t M_out, int K, int Cin, int Cout, bool fused = false) {
  bool noslab = fused;
  static int nb_pref = -1, target = -1, nw_force = -1, zs_force = -1, da_pref = -1, nw_max = -1, bd_pref = -1;
  static int noslab_all = -1, nw_max_noslab = 16;
  if (noslab_all < 0) {
    noslab_all = env_int("WSIS_FWD2_NOSLAB", 1);
    nw_max_noslab = env_int("WSIS_FWD2_NW_MAX_NOSLAB", 16);
  }
  noslab = noslab || noslab_all != 0;
  // ... except where a launch has so few work items that ONE item per CU is its whole schedule: the 344-row level of
  // the C2 scene is 55 items of 135 steps each -- 16 waves of one CU need 16.5 us of matrix-pipe time for an item while
  // 200 CUs idle; its 3x3x3 products split over offset slabs (+ the fixed-order sum) take 14.5 us instead of 25.5.
  // The 2x2x2 strided products of such a level are faster without slabs (9.6 against 13.3 us).
  const char* slab_env = getenv("WSIS_FWD2_SLAB_ITEMS");      // (read per call)
  const int slab_items = slab_env ? atoi(slab_env) : 96;
  if (!fused && noslab && K >= 16 && ceil_div(M_out, SL) * (Cout / 32) <= slab_items) noslab = false;
  if (nb_pref < 0) {
    bd_pref = env_int("WSIS_FWD2_BD", 1);
    nb_pref = env_int("WSIS_FWD2_NB", 1);
    target = env_int("WSIS_FWD2_WAVES", 8192);
    nw_force = env_int("WSIS_FWD2_NW", 0);
    zs_force = env_int("WSIS_FWD2_ZS", 0);
    da_pref = env_int("WSIS_FWD2_DA", 2);
    nw_max = env_int("WSIS_FWD2_NW_MAX", 4);
  }
  Plan2 p;
  const int nblk = Cout / 32;
  p.NB = (!fused && nb_pref >= 2 && nblk % 2 == 0) ? 2 : 1;
  const int steps = K * (Cin / 32);     // steps of a dense work item
  const int nwm = noslab ? nw_max_noslab : nw_max;
  int nw = 1;
  int64_t items = 0;
  for (;;) {
    items = ceil_div(M_out, SL) * (nblk / p.NB);
    nw = 1;
    // up to 4 waves per work item while the launch stays below ~8192 waves (two rounds of the chip's ~4096 resident
    // waves), beyond 4 only while ALL workgroups are resident at once: 8-wave workgroups take 70 KB of LDS (2 per CU)
    // and a launch of 600 of them runs a second, mostly empty round (C2 level 2, 96 -> 96: 44.7 -> 38.3 us with 4 waves)
    while (nw < nwm && items * nw * 2 <= target && nw * 2 <= steps && (nw < 4 || items * nw * 2 <= target / 2)) nw *= 2;
    if (nw_force > 0) nw = nw_force;
    if (p.NB == 2 && nw > 4) {      // two output blocks per work item are built for up to 4 waves
      p.NB = 1;
      continue;
    }
    break;
  }
  int zs = 1;
  if (!noslab) {
    while (zs < 8 && items * nw * zs * 2 <= target && zs * 2 <= K && steps / (nw * zs * 2) >= 2) zs *= 2;
    if (zs_force > 0) zs = zs_force;
    if (zs > K) zs = K;
  }
  if (fused) p.NB = 1;       // (recomputed below: the fused kernels are built for one output block per work item)
  p.NW = nw;
  p.ZS = zs;
  p.DA = (da_pref == 2 || nw >= 4) ? 2 : 3;     // 4+ waves per workgroup: two workgroups per CU need the short ring
  p.BD = bd_pref ? 1 : 0;
  return p;
}

}  // namespace

extern "C" {

int32_t wsis_spconv_fwd_t_supported(int32_t K, int32_t Cin, int32_t Cout) {
  return (K >= 1 && K <= KMAX && Cin >= 32 && Cin % 32 == 0 && Cout >= 32 && Cout % 32 == 0) ? 1 : 0;
}

int32_t wsis_spconv_fwd_t_slabs(int64_t M_out, int32_t K, int32_t Cin, int32_t Cout) {
  if (M_out < 1 || !wsis_spconv_fwd_t_supported(K, Cin, Cout)) return 0;
  return plan2(M_out, K, Cin, Cout).ZS;
}

int64_t wsis_spconv_fwd_t_workspace_bytes(int64_t M_out, int32_t K, int32_t Cin, int32_t Cout) {
  if (M_out < 0 || !wsis_spconv_fwd_t_supported(K, Cin, Cout)) return -1;
  const Plan2 p = plan2(M_out, K, Cin, Cout);
  return p.ZS <= 1 ? 256 : (int64_t)p.ZS * M_out * Cout * (int64_t)sizeof(float) + 256;
}

static int spconv_fwd_t_impl(const float* d_X, const int32_t* d_nbr, const int32_t* d_order, const float* d_WT,
                             int32_t flip, const float* d_bias, const float* d_residual, float* d_out, float* d_stats,
                             const BnEpi& epi, int64_t M_in, int64_t M_out, int32_t K, int32_t Cin, int32_t Cout,
                             void* d_ws, int64_t ws_bytes, void* d_sync, void* stream, const wsis_bn_in* bn_in = nullptr,
                             const wsis_stat_target* targets = nullptr, int32_t n_targets = 0, bool fused = false);

int64_t wsis_spconv_fwd_f_workspace_bytes(int64_t M_out, int32_t K, int32_t Cin, int32_t Cout) {
  if (M_out < 0 || !wsis_spconv_fwd_t_supported(K, Cin, Cout)) return -1;
  const int64_t n_part = (M_out + 31) / 32;
  return (int64_t)bn_fin_chunks(n_part) * (Cout / 32) * 96 * (int64_t)sizeof(double) + 256;
}

int wsis_spconv_fwd_f(const float* d_X, const wsis_bn_in* bn_in, const int32_t* d_nbr, const int32_t* d_order,
                      const float* d_WT, int32_t flip, const float* d_bias, const float* d_residual, float* d_out,
                      float* d_stats, const wsis_stat_target* targets, int32_t n_targets, int64_t M_in, int64_t M_out,
                      int32_t K, int32_t Cin, int32_t Cout, void* d_ws, int64_t ws_bytes, void* d_sync, void* stream) {
  WSIS_REQUIRE(n_targets >= 0 && n_targets <= 2 && (n_targets == 0 || (targets && d_stats && d_sync)),
               "statistics targets need the partial buffer and a sync slot");
  WSIS_REQUIRE(!bn_in || (bn_in->mean && bn_in->var), "input BatchNorm without statistics");
  return spconv_fwd_t_impl(d_X, d_nbr, d_order, d_WT, flip, d_bias, d_residual, d_out, d_stats, BnEpi{}, M_in, M_out, K,
                           Cin, Cout, d_ws, ws_bytes, d_sync, stream, bn_in, targets, n_targets, true);
}

int wsis_spconv_fwd_t(const float* d_X, const int32_t* d_nbr, const int32_t* d_order, const float* d_WT, int32_t flip,
                      const float* d_bias, const float* d_residual, float* d_out, float* d_stats, int64_t M_in,
                      int64_t M_out, int32_t K, int32_t Cin, int32_t Cout, void* d_ws, int64_t ws_bytes,
                      void* d_sync, void* stream) {
  return spconv_fwd_t_impl(d_X, d_nbr, d_order, d_WT, flip, d_bias, d_residual, d_out, d_stats, BnEpi{}, M_in, M_out, K,
                           Cin, Cout, d_ws, ws_bytes, d_sync, stream);
}

int wsis_spconv_fwd_t_bn(const float* d_X, const int32_t* d_nbr, const int32_t* d_order, const float* d_WT, int32_t flip,
                         float* d_out, float* d_partials, const float* d_bn_x, const float* d_bn_mean,
                         const float* d_bn_var, const float* d_bn_gamma, const float* d_bn_beta, float eps, int32_t relu,
                         int64_t M_in, int64_t M_out, int32_t K, int32_t Cin, int32_t Cout, void* d_ws, int64_t ws_bytes,
                         void* d_sync, void* stream) {
  WSIS_REQUIRE(d_partials && d_bn_x && d_bn_mean && d_bn_var, "null pointer");
  WSIS_REQUIRE((reinterpret_cast<uintptr_t>(d_bn_x) & 15) == 0, "bn_x must be 16-byte aligned");
  BnEpi epi;
  epi.x = d_bn_x;
  epi.mean = d_bn_mean;
  epi.var = d_bn_var;
  epi.gamma = d_bn_gamma;
  epi.beta = d_bn_beta;
  epi.eps = eps;
  epi.relu = relu;
  return spconv_fwd_t_impl(d_X, d_nbr, d_order, d_WT, flip, nullptr, nullptr, d_out, d_partials, epi, M_in, M_out, K,
                           Cin, Cout, d_ws, ws_bytes, d_sync, stream);
}

static int spconv_fwd_t_impl(const float* d_X, const int32_t* d_nbr, const int32_t* d_order, const float* d_WT,
                             int32_t flip, const float* d_bias, const float* d_residual, float* d_out, float* d_stats,
                             const BnEpi& epi, int64_t M_in, int64_t M_out, int32_t K, int32_t Cin, int32_t Cout,
                             void* d_ws, int64_t ws_bytes, void* d_sync, void* stream, const wsis_bn_in* bn_in,
                             const wsis_stat_target* targets, int32_t n_targets, bool fused) {
  WSIS_REQUIRE(M_in >= 0 && M_out >= 0, "bad sizes");
  WSIS_REQUIRE(wsis_spconv_fwd_t_supported(K, Cin, Cout), "needs K <= 32 and channel counts that are multiples of 32");
  if (M_out == 0) return WSIS_OK;
  WSIS_REQUIRE(d_X && d_WT && d_out, "null pointer");
  WSIS_REQUIRE(d_nbr || (K == 1 && M_in == M_out), "nbr may be null only for the dense 1x1 case");
  WSIS_REQUIRE(M_out < (int64_t)1 << 31 && M_in < (int64_t)1 << 31, "row count exceeds int32");
  WSIS_REQUIRE(M_in * Cin * 4 < (int64_t)1 << 31, "input tensor of 2 GiB or more (32-bit gather offsets): use wsis_spconv_fwd");
  const uint32_t x_bytes = (uint32_t)(M_in * Cin * 4);
  WSIS_REQUIRE(((reinterpret_cast<uintptr_t>(d_X) | reinterpret_cast<uintptr_t>(d_WT)) & 15) == 0,
               "X and WT must be 16-byte aligned");
  const Plan2 p = plan2(M_out, K, Cin, Cout, fused);
  float* partial = nullptr;
  if (p.ZS > 1) {
    WSIS_REQUIRE(d_ws && ws_bytes >= (int64_t)p.ZS * M_out * Cout * (int64_t)sizeof(float), "workspace too small");
    WSIS_REQUIRE((reinterpret_cast<uintptr_t>(d_ws) & 15) == 0, "workspace must be 16-byte aligned");
    partial = static_cast<float*>(d_ws);
  }
  BnIn bin{};
  if (bn_in) {
    bin.mean = bn_in->mean;
    bin.var = bn_in->var;
    bin.gamma = bn_in->gamma;
    bin.beta = bn_in->beta;
    bin.eps = bn_in->eps;
    bin.relu = bn_in->relu;
    WSIS_REQUIRE(p.NB == 1 && p.BD && p.ZS == 1, "fused input BatchNorm: plan without slabs expected");
  }
  StatFin fin{};
  if (n_targets > 0) {
    const int64_t n_part = ceil_div(M_out, SL);
    WSIS_REQUIRE(p.NB == 1 && p.ZS == 1, "in-launch statistics finish: plan without slabs expected");
    WSIS_REQUIRE(d_ws && ws_bytes >= wsis_spconv_fwd_f_workspace_bytes(M_out, K, Cin, Cout), "workspace too small");
    fin.G = bn_fin_chunks(n_part);
    fin.per = (int)ceil_div(n_part, fin.G);
    WSIS_REQUIRE((int64_t)(fin.G + 1) * (Cout / 32) <= (int64_t)(sizeof(SyncSlot::fin) / sizeof(unsigned)), "too many tickets");
    fin.chunk = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(d_ws) + 255) & ~(uintptr_t)255);
    fin.tickets = static_cast<SyncSlot*>(d_sync)->fin;
    fin.n_targets = n_targets;
    for (int i = 0; i < n_targets; ++i) {
      WSIS_REQUIRE(targets[i].mean && targets[i].var, "statistics target without mean / var");
      WSIS_REQUIRE((targets[i].running_mean == nullptr) == (targets[i].running_var == nullptr), "running stats come in pairs");
      fin.mean[i] = targets[i].mean;
      fin.var[i] = targets[i].var;
      fin.rmean[i] = targets[i].running_mean;
      fin.rvar[i] = targets[i].running_var;
      fin.momentum[i] = targets[i].momentum;
    }
  }
  hipStream_t st = as_stream(stream);
  const dim3 grid((unsigned)ceil_div(M_out, SL), (unsigned)(Cout / 32 / p.NB), (unsigned)p.ZS);
  // WSIS_FWD2_DEAL=1 (read per call; default 0): active offsets dealt round-robin to the waves of a work item instead
  // of ownership by offset index.  Measured neutral on the C2 step (level 1: 1046 -> 1035-1050 us per step, level 2:
  // 855 -> 835) -- the waves of a work item are not what its lifetime waits for -- and it gives up the tile-order
  // independence of the results, so it stays off.
  const char* deal_env = getenv("WSIS_FWD2_DEAL");
  // (wave priorities by step count -- s_setprio 1..3 for waves with many steps, the launch lasts as long as its longest
  // wave -- measured neutral on every level: not kept)
  const int flip_deal = (flip ? 1 : 0) | (((deal_env ? atoi(deal_env) : 0) != 0 && p.ZS == 1) ? 2 : 0);
  ProfScope prof(0, st, /*exact_events=*/true);
  // the persistent form pays where a launch is split into offset slabs (deep levels: 15 % faster at level 3 of the
  // C2 scene); WSIS_FWD3=2 forces it wherever it applies, 0 disables it
  static int fwd3_on = -1, fwd3_wgs = 768, fwd3_min = 2;
  if (fwd3_on < 0) {
    fwd3_on = env_int("WSIS_FWD3", 1);
    fwd3_wgs = env_int("WSIS_FWD3_WGS", 768);        // resident 4-wave workgroups of the whole chip (3 per CU)
    fwd3_min = env_int("WSIS_FWD3_MIN_SLICES", 2);   // slices per workgroup below which the one-shot kernel is kept
  }
  // (the slice queue needs a sync slot with one counter per (output block, offset slab))
  if (fwd3_on && !bn_in && n_targets == 0 && d_sync && (Cout / 32) * p.ZS <= (int)(sizeof(SyncSlot::ctr) / sizeof(unsigned)) && p.NB == 1 && p.NW == 4 &&
      d_nbr && d_order && ceil_div(K, 4 * p.ZS) <= P3_GS && (int64_t)K * M_out * 4 < ((int64_t)1 << 31)) {
    const int64_t n_slices = ceil_div(M_out, SL);
    int64_t P = fwd3_wgs / ((Cout / 32) * p.ZS);
    if (P < 1) P = 1;
    if (P > n_slices) P = n_slices;
    // without slabs: in the per-layer benchmark level 1 of the C2 scene goes 48.0 -> 45.8 us with the slice queue, but
    // inside the training step (epilogues with statistics / residuals, K = 8 tables) the launches of that class average
    // slower than on the one-shot kernel (42.3 vs 40.6 us over the family), so only slab-split launches take this path
    if (n_slices >= fwd3_min * P && (p.ZS > 1 || fwd3_on >= 2)) {
      const dim3 g3((unsigned)P, (unsigned)(Cout / 32), (unsigned)p.ZS);
      const size_t ldsb = (size_t)P3_WG + (size_t)P3_WAVE * 4;
      static bool attr3 = false;
      if (!attr3) {
        WSIS_HIP_CHECK(hipFuncSetAttribute((const void*)spconv_fwd3_kernel<4, false>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
        attr3 = true;
      }
      prof.bracket();
      hipLaunchKernelGGL((spconv_fwd3_kernel<4, false>), g3, dim3(256), ldsb, st, d_X, d_nbr, d_order, d_WT, d_bias,
                         d_residual, d_out, partial, M_out, K, Cin, Cout, flip, x_bytes, d_stats, epi,
                         static_cast<SyncSlot*>(d_sync)->ctr, (unsigned long long*)nullptr);
      goto launched;
    }
  }
#define WSIS_F2X(nb, nw, da, bd, fb)                                                                             \
  do {                                                                                                           \
    const size_t ldsb = (size_t)HDR_BYTES + (size_t)Layout<nb, da, bd>::WAVE_BYTES * nw + (fb ? (size_t)Cin * 12 : 0); \
    static size_t attr_set = 0;                                                                                  \
    if (attr_set < ldsb) {                                                                                       \
      WSIS_HIP_CHECK(hipFuncSetAttribute((const void*)spconv_fwd2_kernel<nb, nw, da, bd, false, fb>,             \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));               \
      attr_set = ldsb;                                                                                           \
    }                                                                                                            \
    hipExtLaunchKernelGGL((spconv_fwd2_kernel<nb, nw, da, bd, false, fb>), grid, dim3(64 * nw), (uint32_t)ldsb, st, \
                          prof.ka(), prof.kb(), 0u, d_X, d_nbr, d_order, d_WT, d_bias, d_residual, d_out, partial,   \
                          M_out, K, Cin, Cout, flip_deal, x_bytes, d_stats, epi, bin, fin,                          \
                          (unsigned long long*)nullptr);                                                            \
  } while (0)
#define WSIS_F2(nb, nw, da)             \
  if (p.BD)                             \
    WSIS_F2X(nb, nw, da, true, false);  \
  else                                  \
    WSIS_F2X(nb, nw, da, false, false)
#define WSIS_F2B(nw) /* NB = 1, DA = 2, weights to registers: with or without the fused input BatchNorm */ \
  if (bn_in)                                                                                               \
    WSIS_F2X(1, nw, 2, true, true);                                                                        \
  else                                                                                                     \
    WSIS_F2X(1, nw, 2, true, false)
  if (p.NB == 1 && p.DA == 2 && p.BD && (bn_in || p.NW == 16)) {
    switch (p.NW) {
      case 1: WSIS_F2B(1); break;
      case 2: WSIS_F2B(2); break;
      case 4: WSIS_F2B(4); break;
      case 8: WSIS_F2B(8); break;
      case 16: WSIS_F2B(16); break;
      default: return fail(WSIS_ERR_ARG, "spconv_fwd_f: unsupported plan NW=%d", p.NW);
    }
  } else {
  WSIS_REQUIRE(!bn_in, "fused input BatchNorm: unsupported plan");
  const int key = p.NB * 100 + p.NW * 10 + p.DA;
  switch (key) {
    case 113: WSIS_F2(1, 1, 3); break;
    case 112: WSIS_F2(1, 1, 2); break;
    case 123: WSIS_F2(1, 2, 3); break;
    case 122: WSIS_F2(1, 2, 2); break;
    case 142: WSIS_F2(1, 4, 2); break;
    case 182: WSIS_F2(1, 8, 2); break;
    case 213: WSIS_F2(2, 1, 3); break;
    case 212: WSIS_F2(2, 1, 2); break;
    case 223: WSIS_F2(2, 2, 3); break;
    case 222: WSIS_F2(2, 2, 2); break;
    case 242: WSIS_F2(2, 4, 2); break;
    default:
      return fail(WSIS_ERR_ARG, "spconv_fwd_t: unsupported plan NB=%d NW=%d DA=%d", p.NB, p.NW, p.DA);
  }
  }
#undef WSIS_F2B
#undef WSIS_F2
#undef WSIS_F2X
launched:
  prof.stop();
  WSIS_LAUNCH_CHECK();
  if (p.ZS > 1 && d_stats) {
    hipLaunchKernelGGL(spconv2_reduce_stats_kernel, dim3((unsigned)ceil_div(M_out, SL), (unsigned)(Cout / 32)), dim3(256), 0, st,
                       reinterpret_cast<const float4*>(partial), reinterpret_cast<const float4*>(d_bias),
                       reinterpret_cast<const float4*>(d_residual), reinterpret_cast<float4*>(d_out), M_out, Cout / 4,
                       p.ZS, d_stats, epi);
    prof.tail();
    WSIS_LAUNCH_CHECK();
  } else if (p.ZS > 1) {
    const int64_t total4 = M_out * Cout / 4;
    hipLaunchKernelGGL(spconv2_reduce_kernel, dim3(grid_for(total4, 256)), dim3(256), 0, st,
                       reinterpret_cast<const float4*>(partial), reinterpret_cast<const float4*>(d_bias),
                       reinterpret_cast<const float4*>(d_residual), reinterpret_cast<float4*>(d_out), total4, Cout / 4,
                       p.ZS);
    prof.tail();
    WSIS_LAUNCH_CHECK();
  }
  return WSIS_OK;
}

// diagnostic (not part of the ABI header): the NW = 1 kernel with per-workgroup stamps, dbg[ceil(M/32) * Cout/32 * 8];
// variant 0: weights direct to registers, ring depth 2; 1: both operands through LDS rings, depth 3
int wsis_debug_spconv2_diag(const float* d_X, const int32_t* d_nbr, const int32_t* d_order, const float* d_WT,
                            float* d_out, int64_t M_out, int32_t K, int32_t Cin, int32_t Cout, int32_t variant,
                            unsigned long long* d_dbg, void* d_sync, void* stream) {
  WSIS_REQUIRE(wsis_spconv_fwd_t_supported(K, Cin, Cout) && d_dbg, "bad args");
  const dim3 grid((unsigned)ceil_div(M_out, SL), (unsigned)(Cout / 32), 1);
  hipStream_t st = as_stream(stream);
  const int dflags = (variant >> 8) << 4;      // experiment bits (see the kernel)
  variant &= 0xff;
  if (variant == 0) {
    const size_t ldsb = (size_t)HDR_BYTES + (size_t)Layout<1, 2, true>::WAVE_BYTES;
    hipLaunchKernelGGL((spconv_fwd2_kernel<1, 1, 2, true, true>), grid, dim3(64), ldsb, st, d_X, d_nbr, d_order, d_WT,
                       (const float*)nullptr, (const float*)nullptr, d_out, (float*)nullptr, M_out, K, Cin, Cout, dflags,
                       (uint32_t)(M_out * Cin * 4), (float*)nullptr, BnEpi{}, BnIn{}, StatFin{}, d_dbg);
  } else if (variant >= 100 && !d_sync) {
    return fail(WSIS_ERR_ARG, "the persistent form needs a sync slot");
  } else if (variant >= 100) {    // persistent form (spconv_fwd3_kernel), variant - 100 offset slabs, 768 resident workgroups;
                                  // stamps: 8 x u64 per wave {total, prologue, steps, barrier wait, epilogue, n steps, n slices}
    const int zs = variant - 100;
    int64_t P = 768 / ((Cout / 32) * zs);
    const int64_t n_slices = ceil_div(M_out, SL);
    if (P < 1) P = 1;
    if (P > n_slices) P = n_slices;
    const dim3 g3((unsigned)P, (unsigned)(Cout / 32), (unsigned)zs);
    const size_t ldsb = (size_t)P3_WG + (size_t)P3_WAVE * 4;
    WSIS_HIP_CHECK(hipFuncSetAttribute((const void*)spconv_fwd3_kernel<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)ldsb));
    hipLaunchKernelGGL((spconv_fwd3_kernel<4, true>), g3, dim3(256), ldsb, st, d_X, d_nbr, d_order, d_WT,
                       (const float*)nullptr, (const float*)nullptr, d_out, d_out, M_out, K, Cin, Cout, 0,
                       (uint32_t)(M_out * Cin * 4), (float*)nullptr, BnEpi{}, static_cast<SyncSlot*>(d_sync)->ctr, d_dbg);
  } else if (variant >= 2) {      // the production small-level form: 4 waves per work item, `variant - 1` offset slabs
    const dim3 g4((unsigned)ceil_div(M_out, SL), (unsigned)(Cout / 32), (unsigned)(variant - 1));
    const size_t ldsb = (size_t)HDR_BYTES + (size_t)Layout<1, 2, true>::WAVE_BYTES * 4;
    hipLaunchKernelGGL((spconv_fwd2_kernel<1, 4, 2, true, true>), g4, dim3(256), ldsb, st, d_X, d_nbr, d_order, d_WT,
                       (const float*)nullptr, (const float*)nullptr, d_out, d_out, M_out, K, Cin, Cout, dflags,
                       (uint32_t)(M_out * Cin * 4), (float*)nullptr, BnEpi{}, BnIn{}, StatFin{}, d_dbg);
  } else {
    const size_t ldsb = (size_t)HDR_BYTES + (size_t)Layout<1, 3, false>::WAVE_BYTES;
    hipLaunchKernelGGL((spconv_fwd2_kernel<1, 1, 3, false, true>), grid, dim3(64), ldsb, st, d_X, d_nbr, d_order, d_WT,
                       (const float*)nullptr, (const float*)nullptr, d_out, (float*)nullptr, M_out, K, Cin, Cout, 0,
                       (uint32_t)(M_out * Cin * 4), (float*)nullptr, BnEpi{}, BnIn{}, StatFin{}, d_dbg);
  }
  WSIS_LAUNCH_CHECK();
  return WSIS_OK;
}

}  // extern "C"
