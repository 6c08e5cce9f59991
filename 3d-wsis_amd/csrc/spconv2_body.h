// Body of the wave-autonomous sparse convolution (see spconv2.hip for the design notes): shared by the one-shot kernel
// spconv_fwd2_kernel (spconv2.hip) and the resident deep-level kernel (deep.hip).  Everything here lives in the
// including file's anonymous namespace.
#pragma once
#include <cstdlib>
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
constexpr int HDR_TR_BYTES = 256;             // TR (one-wave items, table entries in registers): rowId [32] only
constexpr int RG_SCRATCH_BYTES = 1024;        // RG (gathered rows straight to registers): no ring, epilogue scratch only

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

// synchronisation of the NW waves of a work item: the whole workgroup (one work item per workgroup: the one-shot kernel)
struct WgSync {
  __device__ __forceinline__ void operator()() const { __syncthreads(); }
};

// The body of one work item: 32 output rows (slice bx in tile order) x NB 32-channel blocks (block group by), offset slab
// bz of gz, by the NW waves tid / 64 of a group of 64 NW threads whose LDS region starts at `lds` and whose barrier is
// `sync`.  spconv_fwd2_kernel runs it once per workgroup; the resident deep-level kernel (deep.hip) runs it in a loop,
// several work items of one workgroup side by side.  WT: every store another workgroup of the SAME launch reads later
// (output rows, statistics partials) is write-through (sc1) -- the hand-off form of cdna_hip_programming.md Guideline 16.
// The operands only the epilogue needs (bias, residual, output, slab workspace, statistics partials, the BatchNorm of the
// backward reduction) come through `late`, asked for where the epilogue starts: as kernel arguments (LateVals) the
// compiler re-loads them from the kernarg segment at will; as fields of a phase record in memory (deep.hip) they would
// otherwise sit in ~24 scalar registers through the whole walk, and the resident kernel runs at the 102-SGPR limit.
struct LateVals {
  const float* bias_;
  const float* residual_;
  float* out_;
  float* partial_;
  float* stats_;
  BnEpi epi_;
  __device__ __forceinline__ const float* bias() const { return bias_; }
  __device__ __forceinline__ const float* residual() const { return residual_; }
  __device__ __forceinline__ float* out() const { return out_; }
  __device__ __forceinline__ float* partial() const { return partial_; }
  __device__ __forceinline__ float* stats() const { return stats_; }
  __device__ __forceinline__ BnEpi epi() const { return epi_; }
};

// flip_deal bits 8..15 (TR kernels built with RGH): items of at least that many ACTIVE OFFSETS take the register-gather
// walk (0: none); bit 3 (set by the kernel for workgroups dispatched after the first full round): this item does
__device__ __forceinline__ int rg_min_steps(int flip_deal) {
  const int m = (flip_deal >> 8) & 255;
  return m ? m : 0x00ffffff;
}

template <int NB, int NW, int DA, bool BD, bool DIAG, bool FB, bool WT, bool TR = false, bool RG = false, bool RGH = false,
          typename Sync, typename Late>
__device__ __forceinline__ void fwd2_body(
    const float* __restrict__ X, const int32_t* __restrict__ nbrS, const int32_t* __restrict__ order,
    const float* __restrict__ WTp, const Late late, int64_t M_out, int K, int Cin, int Cout, int flip_deal,
    uint32_t x_bytes, const BnIn& bin, const StatFin* const finp,      // finp: nullptr = no in-launch statistics finish
    unsigned long long* __restrict__ dbg, const int bx, const int by, const int bz, const int gy, const int gz,
    unsigned char* const lds, const int tid, Sync& sync) {
  static_assert(!FB || (BD && NB == 1), "the fused input BatchNorm is built for the weights-to-registers form");
  static_assert(!TR || (BD && NB == 1 && NW == 1 && (DA == 2 || DA == 3) && !FB),
                "table entries in registers: the one-wave form only");
  static_assert(!RG || (BD && NB == 1 && NW == 1 && DA == 2 && !FB && !TR), "gathered rows to registers: the one-wave form only");
  static_assert(!RGH || TR, "the per-item choice of the walk lives in the table-in-registers kernel");
  // flip_deal: bit 0 = offset k uses weight slice K - 1 - k; bit 1 = the waves of a work item are dealt the slice's
  // ACTIVE offsets round-robin (see the ownership block below)
  const int flip = flip_deal & 1;
  const bool deal_active = (flip_deal & 2) != 0;
  // stats (optional, final pass only): per-slice BatchNorm partials of the FINISHED output rows (bias and residual
  // included), stats[(slice * 2 + {0: sum, 1: sum of squared deviations from the SLICE mean}) * Cout + channel] -- the
  // statistics pass of the BatchNorm that consumes this tensor (sparse_unet3d.py:128-137) without re-reading it.
  // Centred per slice (and combined in fp64 by wsis_bn_stats_finalize): no E[x^2] - mean^2 cancellation in fp32.
  // DIAG build only (tools/conv2_stamps.py): per-workgroup stamps, dbg[bx * 8 + i] =
  // {realtime at entry, realtime at exit, cycles: prologue, walk, epilogue, steps, HW_ID, 0}
  unsigned long long d_t0 = 0, d_r0 = 0, d_t1 = 0, d_t2 = 0;
  if (DIAG) {
    d_r0 = __builtin_amdgcn_s_memrealtime();
    d_t0 = __builtin_amdgcn_s_memtime();
  }
  using L = Layout<NB, DA, BD>;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r31 = lane & 31, half = lane >> 5;
  // TR: the header is the slice's row ids alone; the gather table is staged in the (still empty) ring for the first
  // three steps and read from the packed table in memory, one iteration ahead, for the rest (see the TR walk)
  constexpr int HDR = TR ? HDR_TR_BYTES : HDR_BYTES;
  int32_t* const nbT = reinterpret_cast<int32_t*>(TR ? lds + HDR_TR_BYTES : lds);
  int32_t* const rowId = TR ? reinterpret_cast<int32_t*>(lds) : nbT + KMAX * 32;
  unsigned char* const Aring = lds + HDR + wave * L::WAVE_BYTES;
  unsigned char* const Bring = Aring + DA * A_BYTES;
  // FB: per-channel (mean, scale, shift) of the input BatchNorm behind the rings, 3 x Cin floats; every wave writes
  // the identical values itself (like the header: no workgroup barrier in the prologue)
  float* const coef = reinterpret_cast<float*>(lds + HDR + NW * L::WAVE_BYTES);
  if (FB) {
    for (int c = lane; c < Cin; c += 64) {
      coef[c] = bin.mean[c];
      coef[Cin + c] = (bin.gamma ? bin.gamma[c] : 1.0f) * rsqrtf(bin.var[c] + bin.eps);
      coef[2 * Cin + c] = bin.beta ? bin.beta[c] : 0.0f;
    }
  }

  const int64_t t0 = (int64_t)bx * SL;
  const int col0 = by * (NB * 32);
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
      nbT[k * 32 + r31] = TR ? g : g >= 0 ? (int32_t)((uint32_t)g * a_pitch32) : (int32_t)NO_ROW;
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
    const int zs = gz, z = bz;
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
  const char* const Wb = reinterpret_cast<const char*>(WTp);
  const char* const zrow = reinterpret_cast<const char*>(g_zero_row) + d_piece * 16;

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
  // ---- register-gather walk (RG): the whole walk of a one-wave work item with BOTH operands straight to registers.
  // Lane (row r31, half) loads the 64 bytes of its own gathered row that it multiplies -- exactly the fragment the LDS
  // ring hands it -- and its 64 bytes of the weight column, each as four 16-byte buffer loads (a missing pair / a
  // finished stream: offset past the end of the buffer, the load returns zero and touches no memory).  No LDS-DMA (an
  // LDS-DMA instruction costs its wave 100-185 cycles to issue, four per step: a wave alone on its SIMD -- the long items
  // and the last round of a launch -- makes a step per 2,200-2,700 cycles against 1,024 of MFMAs), no ring, no fragment
  // reads.  Two register sets per operand, refilled a QUARTER at a time: the four MFMAs of quarter q of step t read
  // a[t&1][q], b[t&1][q]; right behind them the same registers are loaded for step t + 2 -- every load has two full
  // steps to land.  Loads go out in a fixed pattern (B then A per quarter, dummies past the end), so before any quarter
  // exactly 14 younger loads may be in flight: one constant counted wait.  Same operands, same chain: bit-identical to
  // the ring form.  The price: a row-per-lane load touches four cache lines per quad of lanes where the coalesced
  // LDS-DMA piece touches one -- with every wave of a full round walking this way the texture path saturates (7,760
  // cycles per step against 5,300), so it is the walk of the waves that run ALONE (see the TR block).
  // `entry_at(i)`: byte offset of the row under table image entry i (0x80000000 for a missing pair).
  auto rg_walk = [&](auto entry_at) {
      Gen g;
      gen_init(g);
      typedef int v4i __attribute__((ext_vector_type(4)));
      auto make_rs = [&](const void* p, uint32_t bytes) {
        const uint64_t pa = reinterpret_cast<uint64_t>(p);
        v4i r;
        r[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)pa);
        r[1] = __builtin_amdgcn_readfirstlane((int)((uint32_t)(pa >> 32) & 0xffffu));
        r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
        r[3] = __builtin_amdgcn_readfirstlane(0x00020000);
        return r;
      };
      const v4i rsA = make_rs(X, x_bytes);
      const v4i rsW = make_rs(WTp, (uint32_t)((int64_t)K * Cout * Cin * 4));
      const uint32_t a_lane = (uint32_t)half * 64u;
      const uint32_t b_lane = (uint32_t)(r31 * Cin + half * 16) * 4u;
      // per step: lane offsets of the two streams and their scalar parts (chunk of the row / block of the weights)
      struct Step {
        uint32_t offa, offb, soa, sow;
      };
      bool t_dbg = false;                             // (DIAG: the weights are loaded for the first step only)
      auto entry_of = [&](const Gen& gg) { return dbg_row0 ? (uint32_t)r31 * a_pitch32 : entry_at(gg.k * 32 + r31); };
      auto step_of = [&](const Gen& gg, uint32_t e) {
        Step st;
        st.offa = gg.valid ? e + a_lane : NO_ROW;
        st.offb = (gg.valid && !(dbg_noB && t_dbg)) ? b_lane : NO_ROW;
        const int kk = flip ? K - 1 - gg.k : gg.k;
        st.soa = __builtin_amdgcn_readfirstlane((uint32_t)gg.c * 128u);
        st.sow = __builtin_amdgcn_readfirstlane((uint32_t)((((int64_t)kk * Cout + col0) * Cin + gg.c * 32) * 4));
        return st;
      };
#define WSIS_RG_LOAD(nop, dst, off, rs, so, imm) \
  asm volatile(nop "buffer_load_dwordx4 %0, %1, %2, %3 offen offset:" #imm : "=v"(dst) : "v"(off), "s"(rs), "s"(so) : "memory")
      auto fill = [&](const Step& st, f32x4 (&a)[4], f32x4 (&b)[NB][4], int q) {
        if (q == 0) {
          WSIS_RG_LOAD("s_nop 4\n\t", b[0][0], st.offb, rsW, st.sow, 0);
          WSIS_RG_LOAD("", a[0], st.offa, rsA, st.soa, 0);
        } else if (q == 1) {
          WSIS_RG_LOAD("s_nop 4\n\t", b[0][1], st.offb, rsW, st.sow, 16);
          WSIS_RG_LOAD("", a[1], st.offa, rsA, st.soa, 16);
        } else if (q == 2) {
          WSIS_RG_LOAD("s_nop 4\n\t", b[0][2], st.offb, rsW, st.sow, 32);
          WSIS_RG_LOAD("", a[2], st.offa, rsA, st.soa, 32);
        } else {
          WSIS_RG_LOAD("s_nop 4\n\t", b[0][3], st.offb, rsW, st.sow, 48);
          WSIS_RG_LOAD("", a[3], st.offa, rsA, st.soa, 48);
        }
      };
      f32x4 a0[4], a1[4], b0[NB][4], b1[NB][4];
      uint32_t e_next;                                // table entry of the step the next iteration loads
      {
        const Step s0 = step_of(g, entry_of(g));
        gen_next(g);
        t_dbg = true;
        const Step s1 = step_of(g, entry_of(g));
        gen_next(g);
        e_next = entry_of(g);
#pragma unroll
        for (int q = 0; q < 4; ++q) fill(s0, a0, b0, q);
#pragma unroll
        for (int q = 0; q < 4; ++q) fill(s1, a1, b1, q);
      }
      int t = 0;
      auto iter = [&](f32x4 (&a)[4], f32x4 (&b)[NB][4]) {
        const Step st = step_of(g, e_next);           // step t + 2 (a finished stream: dummies)
        gen_next(g);
        e_next = entry_of(g);                         // (an LDS read: back long before the next iteration)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
          asm volatile("" : "+v"(a[q]), "+v"(b[0][q])::"memory");
          __builtin_amdgcn_sched_barrier(0);
          mfma(a, b, 4 * q, 4 * q + 4);
          __builtin_amdgcn_sched_barrier(0);
          fill(st, a, b, q);
          __builtin_amdgcn_sched_barrier(0);
        }
        ++t;
      };
      while (t < T) {
        iter(a0, b0);
        if (t < T) iter(a1, b1);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the dummies of the last two steps
#undef WSIS_RG_LOAD
  };
  if constexpr (RG) {
    if (T > 0) rg_walk([&](int i) { return (uint32_t)nbT[i]; });
  } else if constexpr (TR) {
    // ---- one-wave work items with the gather-table entries in REGISTERS: the 4 KB table image is what kept a work
    // item at 12.3 KB of LDS (12 waves per CU); without it 8.25 KB -> 16 waves per CU, the register file's own limit.
    // The lane that issues piece (i, lane) of a step needs the entry of row i*8 + (lane >> 3) under the step's offset:
    // 4 registers per step.  Steps 0 .. 2 take them from the table image the prologue staged in the still empty ring
    // (no second trip to memory before the first gather); step s >= 3 loads them from the packed table itself during
    // iteration s - 3 (the lines were read by the prologue: L2 hits) -- they ride the vmcnt(0) the next iteration
    // starts with.  Issue order of iteration t: B(t+1), T(t+3), A(t+2).  Same MFMA chain, same operands: results
    // bit-identical to the table-in-LDS form.
    // Per ITEM (flip_deal bit 3: this workgroup was dispatched after the launch's first full round; or a long item):
    // the waves that end up alone on their SIMD take the register-gather walk above -- it reads the same staged image
    // (row indices; nothing overwrites it without a ring in use).
    const bool rg_item = RGH && T > 0 && (T >= rg_min_steps(flip_deal) * nchunk || (flip_deal & 8));
    if (rg_item) {
      rg_walk([&](int i) {
        const int32_t g = nbT[i];
        return g >= 0 ? __umul24((uint32_t)g, a_pitch32) : NO_ROW;
      });
    } else if (T > 0) {
      Gen gA, gB, gT;
      gen_init(gA);
      gen_init(gB);
      int aS = 0, arS = 0;
      const int rows = (int)(M_out - t0 < (int64_t)SL ? M_out - t0 : (int64_t)SL);      // rows of a ragged last slice
      const uint32_t b_voff = (uint32_t)(r31 * Cin + half * 16) * 4u;
      auto loadBr = [&](const Gen& g, f32x4 (&b)[NB][4]) {
        const int kk = flip ? K - 1 - g.k : g.k;
        const uint64_t bp = reinterpret_cast<uint64_t>(Wb) + (uint64_t)((((int64_t)kk * Cout + col0) * Cin + g.c * 32) * 4);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)bp);
        const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(bp >> 32));
        const char* base = reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
        asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=v"(b[0][0]) : "v"(b_voff), "s"(base) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:16" : "=v"(b[0][1]) : "v"(b_voff), "s"(base) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:32" : "=v"(b[0][2]) : "v"(b_voff), "s"(base) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:48" : "=v"(b[0][3]) : "v"(b_voff), "s"(base) : "memory");
      };
      auto tieB = [&](f32x4 (&b)[NB][4]) {
        asm volatile("" : "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[0][2]), "+v"(b[0][3])::"memory");
      };
      // the 4 entries of a step: rows i*8 + d_row, clamped to the slice's last row (a ragged last slice must not read
      // past the table; the entries of rows that do not exist are masked at the issue)
      auto loadT = [&](const Gen& g, int32_t (&n)[4]) {
        const uint64_t tp = reinterpret_cast<uint64_t>(nbrS) + (uint64_t)(((int64_t)g.k * M_out + t0) * 4);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)tp);
        const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(tp >> 32));
        const char* base = reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
        uint32_t vo[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int r = i * 8 + d_row;
          vo[i] = (uint32_t)(r < rows ? r : rows - 1) * 4u;
        }
        asm volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2" : "=v"(n[0]) : "v"(vo[0]), "s"(base) : "memory");
        asm volatile("global_load_dword %0, %1, %2" : "=v"(n[1]) : "v"(vo[1]), "s"(base) : "memory");
        asm volatile("global_load_dword %0, %1, %2" : "=v"(n[2]) : "v"(vo[2]), "s"(base) : "memory");
        asm volatile("global_load_dword %0, %1, %2" : "=v"(n[3]) : "v"(vo[3]), "s"(base) : "memory");
      };
      auto tieT = [&](int32_t (&n)[4]) { asm volatile("" : "+v"(n[0]), "+v"(n[1]), "+v"(n[2]), "+v"(n[3])::"memory"); };
      auto stagedT = [&](const Gen& g, int32_t (&n)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) n[i] = nbT[g.k * 32 + i * 8 + d_row];
      };
      auto issueAr = [&](const Gen& g, const int32_t (&n)[4], int slot, int i) {
        const int32_t e = dbg_row0 ? i * 8 + d_row : n[i];
        const bool ok = g.valid && e >= 0 && i * 8 + d_row < rows;
        const uint32_t off = ok ? __umul24((uint32_t)e, a_pitch32) + a_po[i] : NO_ROW;
        bdma16(rsX, off, (uint32_t)g.c * 128u, Aring + slot * A_BYTES + i * 1024);
      };
      f32x4 a0[4], a1[4], b0[NB][4], b1[NB][4];
      int32_t nA[4], nB[4];
      {
        // steps 0 .. DA from the staged image; the ring may be written only once these reads are back
        int32_t n0[4], n1[4], n2[4];
        gT = gA;
        stagedT(gT, n0);
        gen_next(gT);
        stagedT(gT, n1);
        gen_next(gT);
        if (DA == 3) {
          stagedT(gT, n2);
          gen_next(gT);
        }
        stagedT(gT, nA);
        gen_next(gT);                                 // gT: step DA + 1
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) issueAr(gA, n0, 0, i);      // A0
        gen_next(gA);
        loadBr(gB, b0);                                         // B0
        gen_next(gB);
        if (T > 1) {
#pragma unroll
          for (int i = 0; i < 4; ++i) issueAr(gA, n1, 1, i);    // A1
          gen_next(gA);
        }
        if (DA == 3 && T > 2) {
#pragma unroll
          for (int i = 0; i < 4; ++i) issueAr(gA, n2, 2, i);    // A2
          gen_next(gA);
        }
      }
      {   // A0, B0 landed; A1 [A2] may fly
        const int young = (T > DA ? DA : T) - 1;
        if (young >= 2)
          asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (young == 1)
          asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      tieB(b0);
      readfrag(0, 0, a0, b0);
      arS = 1;
      int t = 0;
      auto iter = [&](const f32x4 (&ac)[4], f32x4 (&bc)[NB][4], f32x4 (&an)[4], f32x4 (&bn)[NB][4], int32_t (&nc)[4],
                      int32_t (&nn)[4]) {
        // top: A(t+1), B(t), T(t+DA) landed, the fragment reads of step t are back; DA = 3: A(t+2), the pieces issued
        // last in the previous iteration, may still fly
        if (DA == 3 && t + 2 < T)
          asm volatile("s_waitcnt vmcnt(4)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
        else
          asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
        tieB(bc);
        tieT(nc);
        __builtin_amdgcn_sched_barrier(0);
        mfma(ac, bc, 0, 4);
        __builtin_amdgcn_sched_barrier(0);
        if (t + 1 < T) {
          readfrag(arS, 0, an, bn);
          arS = arS + 1 == DA ? 0 : arS + 1;
          if (!dbg_noB) loadBr(gB, bn);
          gen_next(gB);
        }
        if (t + DA + 1 < T) loadT(gT, nn);
        gen_next(gT);
        const bool more = t + DA < T;
        __builtin_amdgcn_sched_barrier(0);
        mfma(ac, bc, 4, 8);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (more) issueAr(gA, nc, aS, i);
          __builtin_amdgcn_sched_barrier(0);
          mfma(ac, bc, 8 + i, 9 + i);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (more) {
          gen_next(gA);
          aS = aS + 1 == DA ? 0 : aS + 1;
        }
        mfma(ac, bc, 12, 16);
        ++t;
      };
      while (t < T) {
        iter(a0, b0, a1, b1, nA, nB);
        if (t < T) iter(a1, b1, a0, b0, nB, nA);
      }
    }
  } else if (BD && T > 0) {
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
    if (DA >= 4 && T > 3) issueA_all();             // A3
    {   // A0, B0 landed; A1 [A2 [A3]] may fly
      const int young = (T > DA ? DA : T) - 1;
      if (young >= 3)
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else if (young == 2)
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
      // top: A(t+1) and B(t) landed, the fragment reads of step t are back; A(t+DA-1) (DA >= 3: the pieces issued
      // behind B(t) in the previous iteration) may still fly
      if (DA >= 3 && t + DA - 1 < T)
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
  const float* bias = late.bias();
  const float* residual = late.residual();
  float* const stats = late.stats();
  const BnEpi epi = late.epi();
  const bool final_pass = gz == 1;
  float* const dst = final_pass ? late.out() : late.partial() + (int64_t)bz * M_out * Cout;
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
          if (rows[reg] >= 0) {
            if (WT)
              st_sc1(dst + (int64_t)rows[reg] * Cout + c, val[reg]);
            else
              dst[(int64_t)rows[reg] * Cout + c] = val[reg];
          }
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
            if (WT) {
              st_sc1(stats + ((int64_t)bx * 2 + 0) * Cout + c, sa);
              st_sc1(stats + ((int64_t)bx * 2 + 1) * Cout + c, sb);
            } else {
              stats[((int64_t)bx * 2 + 0) * Cout + c] = sa;
              stats[((int64_t)bx * 2 + 1) * Cout + c] = sb;
            }
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
            st_sc1(stats + ((int64_t)bx * 2 + 0) * Cout + c, sa);
            st_sc1(stats + ((int64_t)bx * 2 + 1) * Cout + c, sb);
          }
          if (NB == 1 && finp && finp->chunk)
            stat_finish(*finp, stats, M_out, Cout, (int64_t)bx, col0, gy, by,
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
    const int cc = (int)(tid % (NB * 32)), c = col0 + cc;
    if (bn_mode) kc = bn_coef(epi, c);
    const float bv = bias ? bias[c] : 0.0f;
    int64_t off[PER];
    float rv[PER], xv[PER];
#pragma unroll
    for (int it = 0; it < PER; ++it) {
      const int rr = (tid + it * 64 * NW) / (NB * 32);
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
    sync();
#pragma unroll
    for (int it = 0; it < PER; ++it) {
      const int e = tid + it * 64 * NW;
      float v = 0.0f;
#pragma unroll
      for (int w = 0; w < NW; ++w)
        v += reinterpret_cast<const float*>(lds + HDR + w * L::WAVE_BYTES)[e];
      if (bias) v += bv;
      if (residual) v += rv[it];
      keep[it] = live[it] ? v : 0.0f;
      if (live[it]) {
        if (WT)
          st_sc1(dst + off[it], v);
        else
          dst[off[it]] = v;
      }
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
      sync();
      float* sred = reinterpret_cast<float*>(lds + HDR);
      const int colw = NB * 32;
      sred[tid] = sa;
      sred[64 * NW + tid] = sq;
      sync();
      if (tid < colw) {
        float ta = 0.0f, tb = 0.0f;
        for (int j = tid; j < 64 * NW; j += colw) {
          ta += sred[j];
          tb += sred[64 * NW + j];
        }
        const int c = col0 + tid;
        if (WT) {
          st_sc1(stats + ((int64_t)bx * 2 + 0) * Cout + c, ta);
          st_sc1(stats + ((int64_t)bx * 2 + 1) * Cout + c, tb);
        } else {
          stats[((int64_t)bx * 2 + 0) * Cout + c] = ta;
          stats[((int64_t)bx * 2 + 1) * Cout + c] = tb;
        }
      }
    } else if (stats && final_pass) {      // threads of one column: t, t + NB*32, ...; added in that order
      sync();              // the accumulator copies in the rings are no longer needed
      float* sred = reinterpret_cast<float*>(lds + HDR);
      const int colw = NB * 32, me = tid % colw;
      sred[tid] = sa;
      sync();
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
      sync();
      sred[tid] = sb;
      sync();
      if (tid < colw) {
        float tb = 0.0f;
        for (int j = tid; j < 64 * NW; j += colw) tb += sred[j];
        const int c = col0 + tid;
        st_sc1(stats + ((int64_t)bx * 2 + 0) * Cout + c, ta);
        st_sc1(stats + ((int64_t)bx * 2 + 1) * Cout + c, tb);
      }
      if (NB == 1 && finp && finp->chunk && wave == 0)      // (the partials of a 32-channel block are stored by lanes 0-31 of wave 0)
        stat_finish(*finp, stats, M_out, Cout, (int64_t)bx, col0, gy, by,
                    reinterpret_cast<double*>(lds + HDR));
    }
  }
  if (DIAG && dbg && tid == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t3 = __builtin_amdgcn_s_memtime();
    unsigned long long* d = dbg + ((int64_t)by * ((M_out + SL - 1) / SL) + bx) * 8;
    d[0] = d_r0;
    d[1] = __builtin_amdgcn_s_memrealtime();
    d[2] = d_t1 - d_t0;
    d[3] = d_t2 - d_t1;
    d[4] = t3 - d_t2;
    d[5] = (unsigned long long)T;
    d[6] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);   // HW_REG_HW_ID, 32 bits
    d[7] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);  // HW_REG_XCC_ID
  }
}

// write-through (sc1) store of four floats: two 8-byte agent-scope stores (deep.hip: rows another workgroup of the same
// launch reads after the next grid barrier)
__device__ __forceinline__ void st4_sc1(float* p, const float4 v) {
  union {
    float f[2];
    unsigned long long u;
  } a, c;
  a.f[0] = v.x; a.f[1] = v.y; c.f[0] = v.z; c.f[1] = v.w;
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), a.u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p) + 1, c.u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// out = sum_z partial[z] (+ bias, + residual), four channels per thread, z order fixed; elements t0, t0 + stride, ...
template <bool WT>
__device__ __forceinline__ void reduce_body(const float4* __restrict__ partial, const float4* __restrict__ bias,
                                            const float4* __restrict__ residual, float4* __restrict__ out, int64_t total4,
                                            int cout4, int zs, int64_t t0, int64_t stride) {
  for (int64_t t = t0; t < total4; t += stride) {
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
    if (WT)
      st4_sc1(reinterpret_cast<float*>(out + t), s);
    else
      out[t] = s;
  }
}

int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}

// the same sum for levels whose consumer is a BatchNorm: one workgroup per 32 output rows (the slice granularity of
// the statistics partials), thread = (4 channels, row lane); also writes the slice's (sum, sum of squares) partials
// (body of spconv2_reduce_stats_kernel: virtual block (bx, by) of gy, 256 threads tid, 4 KB of LDS at sred)
template <bool WT, typename Sync>
__device__ __forceinline__ void reduce_stats_body(const float4* __restrict__ partial, const float4* __restrict__ bias,
                                                  const float4* __restrict__ residual, float4* __restrict__ out,
                                                  int64_t M_out, int cout4, int zs, float* __restrict__ stats,
                                                  const BnEpi& epi, const int bx, const int by, const int gy, const int tid,
                                                  float* const sred, Sync& sync, const bool live = true) {
  const int64_t total4 = M_out * cout4;
  // gy > 1: the workgroup takes the 32 channels (8 float4 columns) of block by -- 32 row lanes, one row
  // per thread, five times the workgroups at 160 channels (11 workgroups walked 6 rows x 8 slabs per thread: 10 us)
  const int cw = gy > 1 ? 8 : cout4, cb = gy > 1 ? by * 8 : 0;
  const int lanes = 256 / cw;                            // row lanes (cw <= 64)
  const int c4l = tid % cw, c4 = cb + c4l, rl = tid / cw;
  const int64_t r0 = (int64_t)bx * 32;
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
      if (WT)
        st4_sc1(reinterpret_cast<float*>(out + t), s);
      else
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
  for (int e = 0; e < 4; ++e) sred[tid * 4 + e] = sa[e];
  sync();
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
  sync();
#pragma unroll
  for (int e = 0; e < 4; ++e) sred[tid * 4 + e] = sb[e];
  sync();
  if (tid < cw && live) {
    float tb[4] = {0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < lanes; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) tb[e] += sred[(j * cw + tid) * 4 + e];
    const int Cout = cout4 * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (WT) {
        st_sc1(stats + ((int64_t)bx * 2 + 0) * Cout + (cb + tid) * 4 + e, ta[e]);
        st_sc1(stats + ((int64_t)bx * 2 + 1) * Cout + (cb + tid) * 4 + e, tb[e]);
      } else {
        stats[((int64_t)bx * 2 + 0) * Cout + (cb + tid) * 4 + e] = ta[e];
        stats[((int64_t)bx * 2 + 1) * Cout + (cb + tid) * 4 + e] = tb[e];
      }
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
    noslab_all = tune_int("WSIS_FWD2_NOSLAB", 1);
    nw_max_noslab = tune_int("WSIS_FWD2_NW_MAX_NOSLAB", 16);
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
    bd_pref = tune_int("WSIS_FWD2_BD", 1);
    nb_pref = env_int("WSIS_FWD2_NB", 1);
    target = tune_int("WSIS_FWD2_WAVES", 8192);
    nw_force = tune_int("WSIS_FWD2_NW", 0);
    zs_force = tune_int("WSIS_FWD2_ZS", 0);
    da_pref = env_int("WSIS_FWD2_DA", 2);
    nw_max = tune_int("WSIS_FWD2_NW_MAX", 4);
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
  if (nw == 4 && p.NB == 1 && p.BD) {
    // 4-wave items (levels 1-2 of a scene): ring depth 3 = 52 KB of LDS per workgroup, three per CU.  The launch lasts
    // as long as its heaviest waves, and with depth 2 a wave makes a step per gather latency (~3,400 cycles) even once
    // it has its SIMD to itself; with depth 3 the gather of step t + 2 has two steps to land (read per call)
    const int d = tune_int("WSIS_FWD2_DA_NW4", 2);
    if (d >= 2 && d <= 4) p.DA = d;
  }
  p.BD = bd_pref ? 1 : 0;
  return p;
}


}  // namespace
