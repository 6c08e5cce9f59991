// Resident kernel for the deep levels of the sparse UNet (sparse_unet3d.py:321-350 walked at levels of a few thousand
// rows; SURVEY 7 "Hard parts": "deep levels are launch-latency-bound -> ... persistent kernels").
//
// On the C2 scene levels 2-4 hold 6,572 / 1,520 / 344 voxels: their 54 convolution products and ~100 BatchNorm launches
// per step are chains of dependent launches of 5-40 us each whose work would take a few microseconds at the roofline.
// This file runs a CONTIGUOUS RUN of the executor's op list (csrc/executor.hip) whose tensors have at most
// WSIS_DEEP_ROWS rows as ONE launch: 256 workgroups of 1024 threads (one per CU, all resident) walk a device-side phase
// table; a phase is what used to be a launch, the launch boundary becomes a grid barrier.
//
//   * convolution phases run fwd2_body (spconv2_body.h) -- the SAME code, launch plan (waves per work item, offset
//     slabs) and therefore the same order of additions as the one-shot spconv_fwd2_kernel: results are bit-identical.
//     A workgroup runs up to four work items side by side (sub-groups of NW waves with an LDS arrival counter as their
//     barrier), work items are dealt so that every CU gets the same number of busy sub-groups;
//   * BatchNorm phases: tile = (256 rows, 32 channels); every tile finishes the statistics (or the backward sums) of ITS
//     channel group itself from the slice partials the convolution epilogues wrote -- the arithmetic and order of
//     bn_chunk_centred_stage / bn_sum_chunk_stage (bn.hip), chunks included -- and applies; row block 0 writes
//     mean / var / running statistics (dgamma / dbeta).  No finish launch, no hand-off inside the phase;
//   * slab sums (levels with <= 96 work items keep their offset slabs), concat / split copies: phases of their own.
//
// Hand-off between phases (cdna_hip_programming.md Guideline 16): every byte another workgroup reads in a later phase is
// stored write-through (sc1), every storing wave drains (s_waitcnt vmcnt(0)), then the grid barrier: eight agent-scope
// arrival counters (one per blockIdx.x % 8) that eight lanes of every workgroup poll together; WSIS_DEEP_FENCE bit 2
// selects the XCC-aware variant (L2-local counter + generation word per XCC, one fabric hop by the XCC's last arriver),
// which measures the same.  Every buffer written inside the launch is written ONCE and only read in later phases
// (private workspace per phase), so no line can be resident anywhere before it is final.  All waits are bounded (2 s) and set an error
// word the host checks.
#include <mutex>
#include <vector>

#include <cstddef>
#include <cstring>

#include "deep.h"
#include "spconv2_body.h"

namespace {

constexpr int DEEP_THREADS = 1024;
constexpr int DEEP_LDS_CTRL = 256;                                   // sub-group arrival counters (64 bytes apart)
constexpr int DEEP_ITEM_WAVE = Layout<1, 2, true>::WAVE_BYTES;       // 8 KB: two-slot ring of gathered rows per wave
constexpr int DEEP_BN_CHUNKS = 16;                                   // two-level finish of at most 16 chunks: <= 34,784 rows
constexpr int DEEP_BN_LDS = 6912 + 3 * DEEP_BN_CHUNKS * 32 * 8;      // per 256-thread quarter: red, coefficients, chunk sums
constexpr int DEEP_LDS_BYTES = DEEP_LDS_CTRL + 4 * (HDR_BYTES + 4 * DEEP_ITEM_WAVE);      // NW = 4 x 4 sub-groups: the maximum

struct DeepSync {             // zeroed by the host before every launch
  unsigned grp[8][32];        // arrival counters of the 8 workgroup groups (blockIdx.x % 8), one 128-byte line each
  unsigned members[32];       // XCC-aware form: workgroups resident on XCC x (counted at the start of the launch)
  unsigned xcnt[8][32];       //   arrival counter of XCC x, touched by that XCC only: atomics in its own L2
  unsigned xgen[8][32];       //   generation word of XCC x, written by the XCC's last arriver, polled through L2
  unsigned top[32];           //   the eight XCC leaders (agent scope)
};
// the error word of a bounded wait that ran out: word 19 of the slot, where wsis_native.sync_errors() looks (SyncSlot::err)
#define DEEP_ERR_WORD(s) (&(s)->grp[0][19])
static_assert(sizeof(DeepSync) <= kDeepSyncBytes, "fits a caller sync slot");

// bn_fin_chunks (common.h) on the device: chunks of the two-level statistics finish -- must stay identical
__device__ __forceinline__ int deep_fin_chunks(int n_part) {
  int g = n_part / 64;
  if (g < 1) g = 1;
  if (g > kBnFinChunks) g = kBnFinChunks;
  return g;
}

constexpr unsigned DEEP_SPIN_LIMIT = 4000000u;      // polls of ~0.5 us each: ~2 s

__device__ __forceinline__ unsigned ld_u32_sc1(const unsigned* p) {
  return __hip_atomic_load(const_cast<unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// grid barrier number `epoch` (1, 2, ...) of this launch.  Every workgroup adds to the counter of its group
// (blockIdx.x % 8: eight counters on lines of their own, so the 256 arrivals serialise 32 deep instead of 256 deep) with
// an atomic that returns nothing, then eight lanes of its first wave poll the eight counters together (sc1 loads) until
// all have reached epoch x group size: one one-way trip + one poll round trip, no top-level counter, no second hop.
// fence bit 0: agent-scope acquire behind the barrier, bit 1: release in front (the payload is stored write-through and
// every buffer is written once per launch, so neither is needed: see the file header; kept as switches)
__device__ __forceinline__ void deep_grid_barrier(DeepSync* s, unsigned epoch, int fence) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every wave: its (write-through) stores have left
  __syncthreads();
  if (threadIdx.x < 64) {
    const unsigned lane = threadIdx.x;
    const unsigned nwg = gridDim.x, g = blockIdx.x & 7u;
    const unsigned ngrp = nwg < 8u ? nwg : 8u;
    if (fence & 2) {
      if (lane == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (lane == 0) __hip_atomic_fetch_add(&s->grp[g][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned mine = lane < ngrp ? lane : 0u;
    const unsigned need = epoch * ((nwg - mine + 7u) / 8u);
    unsigned spins = 0;
    for (;;) {
      const unsigned v = ld_u32_sc1(&s->grp[mine][0]);
      if (__all((int)(v - need) >= 0)) break;
      __builtin_amdgcn_s_sleep(1);
      if (++spins > DEEP_SPIN_LIMIT) {      // ~2 s of polling: a workgroup of this launch never became resident
        if (lane == 0) __hip_atomic_store(DEEP_ERR_WORD(s), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
    if (fence & 1) {
      if (lane == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  __syncthreads();
}


// ---- XCC-aware grid barrier (MI355X_MICROARCH.md, price list "barrier-xcd").  The workgroups of one XCC share an L2:
// their arrival counter and generation word never leave it (atomic without sc1, polls that bypass L1 only), one returned
// L2 atomic tells the last arriver of the XCC, which alone crosses the fabric (agent-scope add + poll of the top counter)
// and then releases its XCC.  Placement is read from HW_REG_XCC_ID, never assumed; the member counts are taken at the start
// of every launch behind one flat barrier.
__device__ __forceinline__ unsigned deep_xcc_id() { return __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20) & 7u; }
__device__ __forceinline__ unsigned l2_atomic_inc(unsigned* p) {
  unsigned old;
  asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(old) : "v"(p), "v"(1u) : "memory");
  return old;
}
__device__ __forceinline__ unsigned l2_load(const unsigned* p) {
  unsigned v;
  asm volatile("global_load_dword %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
  return v;
}
struct XccState {
  unsigned xcc, members, nxcc;
};
__device__ __forceinline__ void deep_xcc_barrier(DeepSync* s, unsigned epoch, const XccState& x) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned t = l2_atomic_inc(&s->xcnt[x.xcc][0]);
    unsigned spins = 0;
    if (t == epoch * x.members - 1u) {
      __hip_atomic_fetch_add(&s->top[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while ((int)(ld_u32_sc1(&s->top[0]) - epoch * x.nxcc) < 0) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > DEEP_SPIN_LIMIT) {
          __hip_atomic_store(DEEP_ERR_WORD(s), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      }
      *reinterpret_cast<volatile unsigned*>(&s->xgen[x.xcc][0]) = epoch;      // (L1 is write-through: the word lands in L2)
    } else {
      while ((int)(l2_load(&s->xgen[x.xcc][0]) - epoch) < 0) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > DEEP_SPIN_LIMIT) {
          __hip_atomic_store(DEEP_ERR_WORD(s), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      }
    }
  }
  __syncthreads();
}

// barrier of the NW waves that share a work item (a sub-group of the workgroup): arrival counter in LDS
template <int NW>
struct SubSync {
  unsigned* ctr;
  unsigned target;
  __device__ __forceinline__ void operator()() {
    target += (unsigned)NW;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    while ((int)(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) - target) < 0)
      __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
};

// a field of the phase record by a scalar load the optimiser cannot move or keep (see LateVals in spconv2_body.h)
template <int BYTE_OFF>
__device__ __forceinline__ unsigned long long deep_ld64(const DeepOp* op) {
  unsigned long long v;
  asm volatile("s_load_dwordx2 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(op), "i"(BYTE_OFF) : "memory");
  return v;
}
template <int BYTE_OFF>
__device__ __forceinline__ unsigned deep_ld32(const DeepOp* op) {
  unsigned v;
  asm volatile("s_load_dword %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(op), "i"(BYTE_OFF) : "memory");
  return v;
}
constexpr int DEEP_P_OFF = 72;      // offsetof(DeepOp, p)
static_assert(offsetof(DeepOp, p) == DEEP_P_OFF && offsetof(DeepOp, eps) == 56 && offsetof(DeepOp, relu) == 12, "phase record");
struct LateOp {
  const DeepOp* op;
  __device__ __forceinline__ const float* bias() const { return (const float*)deep_ld64<DEEP_P_OFF + 4 * 8>(op); }
  __device__ __forceinline__ const float* residual() const { return (const float*)deep_ld64<DEEP_P_OFF + 5 * 8>(op); }
  __device__ __forceinline__ float* out() const { return (float*)deep_ld64<DEEP_P_OFF + 6 * 8>(op); }
  __device__ __forceinline__ float* partial() const { return (float*)deep_ld64<DEEP_P_OFF + 7 * 8>(op); }
  __device__ __forceinline__ float* stats() const { return (float*)deep_ld64<DEEP_P_OFF + 8 * 8>(op); }
  __device__ __forceinline__ BnEpi epi() const {
    BnEpi e;
    e.x = (const float*)deep_ld64<DEEP_P_OFF + 9 * 8>(op);
    e.mean = (const float*)deep_ld64<DEEP_P_OFF + 10 * 8>(op);
    e.var = (const float*)deep_ld64<DEEP_P_OFF + 11 * 8>(op);
    e.gamma = (const float*)deep_ld64<DEEP_P_OFF + 12 * 8>(op);
    e.beta = (const float*)deep_ld64<DEEP_P_OFF + 13 * 8>(op);
    e.eps = __uint_as_float(deep_ld32<56>(op));
    e.relu = (int)deep_ld32<12>(op);
    return e;
  }
};

__device__ __forceinline__ BnEpi deep_epi(const DeepOp& op) {
  BnEpi e;
  e.x = (const float*)op.p[9];
  e.mean = (const float*)op.p[10];
  e.var = (const float*)op.p[11];
  e.gamma = (const float*)op.p[12];
  e.beta = (const float*)op.p[13];
  e.eps = op.eps;
  e.relu = op.relu;
  return e;
}

// ---- convolution phase: work items (slice bx, block by, slab bz), SG = min(16 / NW, 4) side by side per workgroup
template <int NW>
__device__ __forceinline__ void deep_conv(const DeepOp& op, unsigned char* lds, unsigned* sub_epoch) {
  constexpr int SG = (16 / NW) < 4 ? (16 / NW) : 4;
  constexpr int ITEM_BYTES = HDR_BYTES + NW * DEEP_ITEM_WAVE;
  static_assert(DEEP_LDS_CTRL + SG * ITEM_BYTES <= DEEP_LDS_BYTES, "LDS budget");
  const int sub = __builtin_amdgcn_readfirstlane((int)threadIdx.x / (64 * NW));
  if (sub >= SG) return;
  const int ltid = (int)threadIdx.x - sub * (64 * NW);
  const int gx = (int)((op.M_out + SL - 1) / SL), gy = op.Cout / 32, gz = op.ZS;
  const int total = gx * gy * gz;
  unsigned char* base = lds + DEEP_LDS_CTRL + sub * ITEM_BYTES;
  const LateOp late{&op};
  const BnIn bin{};
  const int nwg = (int)gridDim.x;
  if (NW == 16) {
    WgSync sync;
    for (int item = (int)blockIdx.x; item < total; item += nwg) {
      const int bx = item % gx, r = item / gx;
      // nothing of one work item is kept in registers for the next: without the two statements below hipcc hoists
      // every lane constant of the body out of the item AND the phase loop and spills them around the MFMA walk
      asm volatile("" ::: "memory");
      int t2 = ltid;
      asm volatile("" : "+v"(t2));
      fwd2_body<1, NW, 2, true, false, false, true>(
          (const float*)op.p[0], (const int32_t*)op.p[1], (const int32_t*)op.p[2], (const float*)op.p[3], late, op.M_out,
          op.K, op.Cin, op.Cout, op.flip, op.x_bytes, bin, nullptr, nullptr, bx, r % gy, r / gy, gy, gz, base, t2, sync);
      sync();        // the next work item re-writes the header and the rings
    }
  } else {
    unsigned* ctr = reinterpret_cast<unsigned*>(lds + sub * 64);
    SubSync<NW> sync{ctr, *sub_epoch};
    // item = round * (nwg SG) + sub * nwg + workgroup: every CU gets the same number of busy sub-groups
    for (int item = sub * nwg + (int)blockIdx.x; item < total; item += nwg * SG) {
      const int bx = item % gx, r = item / gx;
      // nothing of one work item is kept in registers for the next: without the two statements below hipcc hoists
      // every lane constant of the body out of the item AND the phase loop and spills them around the MFMA walk
      asm volatile("" ::: "memory");
      int t2 = ltid;
      asm volatile("" : "+v"(t2));
      fwd2_body<1, NW, 2, true, false, false, true>(
          (const float*)op.p[0], (const int32_t*)op.p[1], (const int32_t*)op.p[2], (const float*)op.p[3], late, op.M_out,
          op.K, op.Cin, op.Cout, op.flip, op.x_bytes, bin, nullptr, nullptr, bx, r % gy, r / gy, gy, gz, base, t2, sync);
      if (NW > 1) sync();
    }
    *sub_epoch = sync.target;
  }
}

// ---- slab sum: out = sum_z partial[z] (+ bias + residual), with the slice partials of the BatchNorm behind it
__device__ __forceinline__ void deep_reduce(const DeepOp& op, unsigned char* lds) {
  const float4* partial = (const float4*)op.p[7];
  const float4* bias = (const float4*)op.p[4];
  const float4* residual = (const float4*)op.p[5];
  float4* out = (float4*)const_cast<void*>(op.p[6]);
  float* stats = (float*)const_cast<void*>(op.p[8]);
  const int cout4 = op.Cout / 4;
  if (!stats) {
    reduce_body<true>(partial, bias, residual, out, op.M_out * cout4, cout4, op.ZS,
                      (int64_t)blockIdx.x * DEEP_THREADS + threadIdx.x, (int64_t)gridDim.x * DEEP_THREADS);
    return;
  }
  // spconv2_reduce_stats_kernel: grid (ceil(M / 32), Cout / 32) of 256 threads; four virtual blocks per workgroup, all
  // four walk the same number of rounds (the barriers inside are workgroup barriers)
  const BnEpi epi = deep_epi(op);
  int tid_ = (int)threadIdx.x;
  asm volatile("" : "+v"(tid_));
  const int sub = tid_ >> 8, tid = tid_ & 255;
  float* sred = reinterpret_cast<float*>(lds + DEEP_LDS_CTRL) + sub * 1024;
  const int gx = (int)((op.M_out + 31) / 32), gy = op.Cout / 32;
  const int total = gx * gy, nwg = (int)gridDim.x;
  WgSync sync;
  for (int base = 0; base < total; base += nwg * 4) {
    const int item = base + sub * nwg + (int)blockIdx.x;
    // (a virtual block past the end runs on block 0's rows with its stores masked: M_out = 0 rows masks everything)
    const bool live = item < total;
    const int bx = live ? item % gx : 0, by = live ? item / gx : 0;
    reduce_stats_body<true>(partial, bias, residual, out, live ? op.M_out : (int64_t)0, cout4, op.ZS, stats, epi, bx, by, gy, tid,
                            sred, sync, live);
    __syncthreads();
  }
}

// ---- BatchNorm forward: statistics finish (training) + apply; tile = (256 rows, 32 channels) per 256-thread quarter
__device__ __forceinline__ void deep_bn_fwd(const DeepOp& op, unsigned char* lds) {
  int tid_ = (int)threadIdx.x;
  asm volatile("" : "+v"(tid_));      // (lane constants stay inside the phase: see deep_conv)
  const int sub = tid_ >> 8, tid = tid_ & 255;
  double(*red)[8][33] = reinterpret_cast<double(*)[8][33]>(lds + DEEP_LDS_CTRL + sub * DEEP_BN_LDS);      // [3][8][33] doubles
  float* s_mu = reinterpret_cast<float*>(lds + DEEP_LDS_CTRL + sub * DEEP_BN_LDS + 6400);
  float* s_sc = s_mu + 32;
  float* s_bt = s_sc + 32;
  double* chunkS = reinterpret_cast<double*>(lds + DEEP_LDS_CTRL + sub * DEEP_BN_LDS + 6912);      // [3][16][32] doubles
  const int C = op.Cin, C0 = op.C0;
  const int64_t M = op.M_in;
  const float* x = (const float*)op.p[0];
  const float* gamma = (const float*)op.p[1];
  const float* beta = (const float*)op.p[2];
  float* rmean = (float*)const_cast<void*>(op.p[3]);
  float* rvar = (float*)const_cast<void*>(op.p[4]);
  float* y = (float*)const_cast<void*>(op.p[7]);
  float* mean = (float*)const_cast<void*>(op.p[8]);
  float* var = (float*)const_cast<void*>(op.p[9]);
  const int nblk = (int)((M + 31) / 32);
  const int G = deep_fin_chunks(nblk), per = (nblk + G - 1) / G;
  const int CG = C / 32, RB = (int)((M + 255) / 256);
  const int total = y ? RB * CG : CG;            // statistics only: one tile per channel group
  const int nwg = (int)gridDim.x;
  const int cl = tid & 31, pl = tid >> 5;
  for (int base = 0; base < total; base += nwg * 4) {
    const int item = base + sub * nwg + (int)blockIdx.x;
    const bool live = item < total;
    const int cgi = live ? item % CG : 0, rb = live ? item / CG : 0;
    const int c = cgi * 32 + cl;
    if (op.training) {
      // partial source of this channel group (a concatenation has two producers)
      const bool first = c < C0;
      const float* part = (const float*)(first ? op.p[5] : op.p[6]);
      const int Cs = first ? C0 : C - C0, cs = first ? c : c - C0;
      // level 1: the G chunk sums of this channel (fixed order inside a chunk) -> chunkS[3][G][32] in LDS
      for (int g = 0; g < G; ++g) {
        const int lo = g * per, hi = lo + per < nblk ? lo + per : nblk;
        double s = 0.0, q = 0.0, w = 0.0;
        if (live) {
          // the rows lo + pl, + 8, ... of the chunk, sixteen loads in flight at a time, added in ascending order
          for (int b0 = lo + pl; b0 < hi; b0 += 8 * 16) {
            float sf[16], qf[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
              const int b = b0 + 8 * u;
              const int bb = b < hi ? b : lo;
              sf[u] = part[(int64_t)bb * 2 * Cs + cs];
              qf[u] = part[(int64_t)bb * 2 * Cs + Cs + cs];
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
              const int b = b0 + 8 * u;
              if (b < hi) {
                const int64_t left = M - (int64_t)b * 32;
                const double si = sf[u];
                s += si;
                q += qf[u];
                w += si * si * (left < 32 ? 1.0 / (double)left : 0.03125);
              }
            }
          }
        }
        __syncthreads();
        red[0][pl][cl] = s;
        red[1][pl][cl] = q;
        red[2][pl][cl] = w;
        __syncthreads();
        if (pl == 0) {
          double S = 0.0, Q = 0.0, W = 0.0;
#pragma unroll
          for (int j = 0; j < 8; ++j) {      // fixed order (bn_chunk_centred_stage)
            S += red[0][j][cl];
            Q += red[1][j][cl];
            W += red[2][j][cl];
          }
          chunkS[(0 * DEEP_BN_CHUNKS + g) * 32 + cl] = S;
          chunkS[(1 * DEEP_BN_CHUNKS + g) * 32 + cl] = Q;
          chunkS[(2 * DEEP_BN_CHUNKS + g) * 32 + cl] = W;
        }
      }
      double S = 0.0, Q = 0.0, W = 0.0;
      if (G > 1) {
        // level 2: partial lane pl adds chunks pl, pl + 8, ...; the eight lanes are added in order
        __syncthreads();
        double s2 = 0.0, q2 = 0.0, w2 = 0.0;
        for (int g = pl; g < G; g += 8) {
          s2 += chunkS[(0 * DEEP_BN_CHUNKS + g) * 32 + cl];
          q2 += chunkS[(1 * DEEP_BN_CHUNKS + g) * 32 + cl];
          w2 += chunkS[(2 * DEEP_BN_CHUNKS + g) * 32 + cl];
        }
        red[0][pl][cl] = s2;
        red[1][pl][cl] = q2;
        red[2][pl][cl] = w2;
        __syncthreads();
        if (pl == 0) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            S += red[0][j][cl];
            Q += red[1][j][cl];
            W += red[2][j][cl];
          }
        }
      } else if (pl == 0) {
        S = chunkS[(0 * DEEP_BN_CHUNKS + 0) * 32 + cl];
        Q = chunkS[(1 * DEEP_BN_CHUNKS + 0) * 32 + cl];
        W = chunkS[(2 * DEEP_BN_CHUNKS + 0) * 32 + cl];
      }
      if (pl == 0) {
        const double n = (double)M;        // (bn_finish_centred)
        const double mu = S / n;
        double v = (Q + (W - n * mu * mu)) / n;
        if (v < 0.0) v = 0.0;
        const float muf = (float)mu, vf = (float)v;
        if (live && rb == 0) {
          mean[c] = muf;
          var[c] = vf;
          if (rmean) {
            const double unb = n > 1 ? v * n / (n - 1) : v;
            rmean[c] = (float)((1.0 - op.momentum) * rmean[c] + op.momentum * mu);
            rvar[c] = (float)((1.0 - op.momentum) * rvar[c] + op.momentum * unb);
          }
        }
        s_mu[cl] = muf;
        s_sc[cl] = (gamma ? gamma[c] : 1.0f) * rsqrtf(vf + op.eps);
        s_bt[cl] = beta ? beta[c] : 0.0f;
      }
    } else if (pl == 0) {        // evaluation: the running statistics (op.p[3], op.p[4])
      s_mu[cl] = rmean[c];
      s_sc[cl] = (gamma ? gamma[c] : 1.0f) * rsqrtf(rvar[c] + op.eps);
      s_bt[cl] = beta ? beta[c] : 0.0f;
    }
    __syncthreads();
    if (y && live) {
      const int q4 = tid & 7, rl = tid >> 3;
      const int c4 = cgi * 32 + q4 * 4;
      float mu4[4], sc4[4], bt4[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        mu4[e] = s_mu[q4 * 4 + e];
        sc4[e] = s_sc[q4 * 4 + e];
        bt4[e] = s_bt[q4 * 4 + e];
      }
      const int64_t r0 = (int64_t)rb * 256;
      float4 vx[8];
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int64_t r = r0 + it * 32 + rl;
        vx[it] = *reinterpret_cast<const float4*>(x + (r < M ? r : 0) * C + c4);
      }
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int64_t r = r0 + it * 32 + rl;
        if (r < M) {
          float o[4] = {vx[it].x, vx[it].y, vx[it].z, vx[it].w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float z = __builtin_fmaf(o[e] - mu4[e], sc4[e], bt4[e]);      // (bn_apply_body)
            if (op.relu) z = fmaxf(z, 0.0f);
            o[e] = z;
          }
          st4_sc1(y + r * C + c4, make_float4(o[0], o[1], o[2], o[3]));
        }
      }
    }
    __syncthreads();
  }
}

// ---- BatchNorm backward from the slice partials (sum dz, sum dz xhat) of the dIn epilogue: finish + dx (+ addend)
__device__ __forceinline__ void deep_bn_bwd(const DeepOp& op, unsigned char* lds) {
  int tid_ = (int)threadIdx.x;
  asm volatile("" : "+v"(tid_));
  const int sub = tid_ >> 8, tid = tid_ & 255;
  double(*red)[8][33] = reinterpret_cast<double(*)[8][33]>(lds + DEEP_LDS_CTRL + sub * DEEP_BN_LDS);      // [2][8][33]
  float* s_k1 = reinterpret_cast<float*>(lds + DEEP_LDS_CTRL + sub * DEEP_BN_LDS + 6400);
  float* s_k2 = s_k1 + 32;
  double* chunkS = reinterpret_cast<double*>(lds + DEEP_LDS_CTRL + sub * DEEP_BN_LDS + 6912);
  const int C = op.Cin;
  const int64_t M = op.M_in;
  const float* x = (const float*)op.p[0];
  const float* dy = (const float*)op.p[1];
  const float* mean = (const float*)op.p[2];
  const float* var = (const float*)op.p[3];
  const float* gamma = (const float*)op.p[4];
  const float* beta = (const float*)op.p[5];
  const float* addend = (const float*)op.p[6];
  const float* part = (const float*)op.p[7];
  float* dx = (float*)const_cast<void*>(op.p[8]);
  float* dgamma = (float*)const_cast<void*>(op.p[9]);
  float* dbeta = (float*)const_cast<void*>(op.p[10]);
  const int nblk = (int)((M + 31) / 32);
  const int G = deep_fin_chunks(nblk), per = (nblk + G - 1) / G;
  const int CG = C / 32, RB = (int)((M + 255) / 256);
  const int total = RB * CG, nwg = (int)gridDim.x;
  const int cl = tid & 31, pl = tid >> 5;
  for (int base = 0; base < total; base += nwg * 4) {
    const int item = base + sub * nwg + (int)blockIdx.x;
    const bool live = item < total;
    const int cgi = live ? item % CG : 0, rb = live ? item / CG : 0;
    const int c = cgi * 32 + cl;
    for (int g = 0; g < G; ++g) {
      const int lo = g * per, hi = lo + per < nblk ? lo + per : nblk;
      double a = 0.0, b = 0.0;
      if (live) {
        for (int k0 = lo + pl; k0 < hi; k0 += 8 * 16) {      // sixteen loads in flight, added in ascending order
          float af[16], bf[16];
#pragma unroll
          for (int u = 0; u < 16; ++u) {
            const int k = k0 + 8 * u;
            const int kk = k < hi ? k : lo;
            af[u] = part[(int64_t)kk * 2 * C + c];
            bf[u] = part[(int64_t)kk * 2 * C + C + c];
          }
#pragma unroll
          for (int u = 0; u < 16; ++u)
            if (k0 + 8 * u < hi) {
              a += af[u];
              b += bf[u];
            }
        }
      }
      __syncthreads();
      red[0][pl][cl] = a;
      red[1][pl][cl] = b;
      __syncthreads();
      if (pl == 0) {
        double A = 0.0, B = 0.0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {      // fixed order (bn_sum_chunk_stage)
          A += red[0][j][cl];
          B += red[1][j][cl];
        }
        chunkS[(0 * DEEP_BN_CHUNKS + g) * 32 + cl] = A;
        chunkS[(1 * DEEP_BN_CHUNKS + g) * 32 + cl] = B;
      }
    }
    double A = 0.0, B = 0.0;
    if (G > 1) {
      __syncthreads();
      double a2 = 0.0, b2 = 0.0;
      for (int g = pl; g < G; g += 8) {
        a2 += chunkS[(0 * DEEP_BN_CHUNKS + g) * 32 + cl];
        b2 += chunkS[(1 * DEEP_BN_CHUNKS + g) * 32 + cl];
      }
      red[0][pl][cl] = a2;
      red[1][pl][cl] = b2;
      __syncthreads();
      if (pl == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          A += red[0][j][cl];
          B += red[1][j][cl];
        }
      }
    } else if (pl == 0) {
      A = chunkS[(0 * DEEP_BN_CHUNKS + 0) * 32 + cl];
      B = chunkS[(1 * DEEP_BN_CHUNKS + 0) * 32 + cl];
    }
    if (pl == 0) {
      const float db = (float)A, dg = (float)B;
      if (live && rb == 0) {
        dbeta[c] = db;
        dgamma[c] = dg;
      }
      const float inv_m = 1.0f / (float)M;      // (bn_bwd_apply_body)
      s_k1[cl] = db * inv_m;
      s_k2[cl] = dg * inv_m;
    }
    __syncthreads();
    if (live) {
      const int q4 = tid & 7, rl = tid >> 3;
      const int c4 = cgi * 32 + q4 * 4;
      float mu[4], rstd[4], gm[4], bt[4], k1[4], k2[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int cc = c4 + e;
        mu[e] = mean[cc];
        rstd[e] = rsqrtf(var[cc] + op.eps);
        gm[e] = gamma ? gamma[cc] : 1.0f;
        bt[e] = beta ? beta[cc] : 0.0f;
        k1[e] = s_k1[q4 * 4 + e];
        k2[e] = s_k2[q4 * 4 + e];
      }
      const int64_t r0 = (int64_t)rb * 256;
#pragma unroll 2
      for (int it = 0; it < 8; it += 2) {
        float4 vx[2], vd[2], va[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int64_t r = r0 + (it + u) * 32 + rl;
          const int64_t rr = r < M ? r : 0;
          vx[u] = *reinterpret_cast<const float4*>(x + rr * C + c4);
          vd[u] = *reinterpret_cast<const float4*>(dy + rr * C + c4);
          va[u] = addend ? *reinterpret_cast<const float4*>(addend + rr * C + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int64_t r = r0 + (it + u) * 32 + rl;
          if (r < M) {
            const float xv[4] = {vx[u].x, vx[u].y, vx[u].z, vx[u].w}, dv[4] = {vd[u].x, vd[u].y, vd[u].z, vd[u].w};
            const float av[4] = {va[u].x, va[u].y, va[u].z, va[u].w};
            float ov[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float xh = (xv[e] - mu[e]) * rstd[e];
              float dz = dv[e];
              if (op.relu && xh * gm[e] + bt[e] <= 0.0f) dz = 0.0f;
              const float rr2 = dz - k1[e] - xh * k2[e];
              ov[e] = gm[e] * rstd[e] * rr2;
              if (addend) ov[e] += av[e];
            }
            st4_sc1(dx + r * C + c4, make_float4(ov[0], ov[1], ov[2], ov[3]));
          }
        }
      }
    }
    __syncthreads();
  }
}

// out[r] = [a[r] | b[r]]  /  a[r], b[r] = halves of in[r]   (float4 granules)
__device__ __forceinline__ void deep_cat(const DeepOp& op, bool split) {
  const int Ca4 = op.Cin / 4, C4 = (op.Cin + op.Cout) / 4;
  const int64_t total = op.M_in * C4;
  const float* a = (const float*)op.p[0];
  const float* b = (const float*)op.p[1];
  float* o = (float*)const_cast<void*>(op.p[2]);
  for (int64_t t = (int64_t)blockIdx.x * DEEP_THREADS + threadIdx.x; t < total; t += (int64_t)gridDim.x * DEEP_THREADS) {
    const int64_t r = t / C4;
    const int c = (int)(t - r * C4);
    if (!split) {
      const float* src = c < Ca4 ? a + (r * Ca4 + c) * 4 : b + (r * (C4 - Ca4) + (c - Ca4)) * 4;
      st4_sc1(o + t * 4, *reinterpret_cast<const float4*>(src));
    } else {      // split: p0 = in, p1 = first half out, p2 = second half out
      float* dst = c < Ca4 ? const_cast<float*>(b) + (r * Ca4 + c) * 4 : o + (r * (C4 - Ca4) + (c - Ca4)) * 4;
      st4_sc1(dst, *reinterpret_cast<const float4*>(a + t * 4));
    }
  }
}

__global__ __launch_bounds__(DEEP_THREADS) void deep_run_kernel(const DeepOp* __restrict__ ops, int n, DeepSync* sync,
                                                                   int fence, unsigned long long* __restrict__ stamps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  if (threadIdx.x < DEEP_LDS_CTRL / 4) reinterpret_cast<unsigned*>(lds)[threadIdx.x] = 0u;
  __syncthreads();
  unsigned sub_epoch = 0u;           // arrivals so far at this wave's sub-group counter (all sub-group sizes share it)
  // fence bit 2: the XCC-aware barrier.  Member counts of the XCCs first (one flat barrier per launch)
  XccState xs{0u, 1u, 1u};
  if (fence & 4) {
    xs.xcc = deep_xcc_id();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(&sync->members[xs.xcc], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    deep_grid_barrier(sync, 1u, 0);
    unsigned mine = 0u, any = 0u;
    if (threadIdx.x < 64) {
      const unsigned m = threadIdx.x < 8 ? ld_u32_sc1(&sync->members[threadIdx.x]) : 0u;
      any = (unsigned)__builtin_popcountll(__ballot(m != 0u));
      mine = __shfl(m, (int)xs.xcc, 64);
    }
    xs.members = __builtin_amdgcn_readfirstlane(mine);
    xs.nxcc = __builtin_amdgcn_readfirstlane(any);
    // (threads of the other waves: through LDS)
    unsigned* xw = reinterpret_cast<unsigned*>(lds) + 32;
    if (threadIdx.x == 0) {
      xw[0] = xs.members;
      xw[1] = xs.nxcc;
    }
    __syncthreads();
    xs.members = xw[0];
    xs.nxcc = xw[1];
    __syncthreads();
  }
  if (stamps && blockIdx.x == 0 && threadIdx.x == 0) stamps[0] = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < n; ++i) {
    const DeepOp& op = ops[i];
    switch (op.kind) {
      case DK_CONV:
        // (the arrival counter is shared by sub-groups of different sizes over the phases: it restarts per phase)
        if (op.NW == 16) {
          deep_conv<16>(op, lds, &sub_epoch);
        } else {
          __syncthreads();
          if (threadIdx.x < DEEP_LDS_CTRL / 4) reinterpret_cast<unsigned*>(lds)[threadIdx.x] = 0u;
          __syncthreads();
          sub_epoch = 0u;
          if (op.NW == 8) deep_conv<8>(op, lds, &sub_epoch);
          else deep_conv<4>(op, lds, &sub_epoch);
        }
        break;
      case DK_REDUCE: deep_reduce(op, lds); break;
      case DK_BN_FWD: deep_bn_fwd(op, lds); break;
      case DK_BN_BWD: deep_bn_bwd(op, lds); break;
      case DK_CAT: deep_cat(op, false); break;
      case DK_SPLIT: deep_cat(op, true); break;
      default: break;
    }
    if (i + 1 < n) {
      if (fence & 4)
        deep_xcc_barrier(sync, (unsigned)(i + 1), xs);
      else
        deep_grid_barrier(sync, (unsigned)(i + 1), fence);
    }
    if (stamps && op.stamp >= 0 && blockIdx.x == 0 && threadIdx.x == 0) {
      // (the last phase: its stores are complete when this workgroup's are -- a lower bound by at most one item)
      if (i + 1 == n) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      stamps[op.stamp] = __builtin_amdgcn_s_memrealtime();
    }
  }
}

}  // namespace

// -------------------------------------------------------------------------------------------------------------------
// host side: called by the executor (csrc/executor.hip) with a run of ops it found eligible
namespace wsis {

// default OFF: measured on the C2 scene the resident launches take 1.41 + 1.45 ms against ~2.45 ms of the same ops launch by
// launch (kernels + boundaries) -- every phase, however small, costs a cold load, a write-through store and the barrier
// (~7 us), where a launch boundary costs ~1.7 us on top of the same latencies
bool deep_enabled() {
  const char* e = getenv("WSIS_DEEP");
  return e ? atoi(e) != 0 : false;
}

int64_t deep_max_rows() {      // (read per call: a test raises it to put a whole small network into one launch)
  const char* e = getenv("WSIS_DEEP_ROWS");
  return e ? (int64_t)atoll(e) : (int64_t)8192;
}

// BatchNorm phases finish the statistics in at most DEEP_BN_CHUNKS chunks (LDS)
bool deep_bn_rows_ok(int64_t M) { return M >= 1 && bn_fin_chunks((M + 31) / 32) <= DEEP_BN_CHUNKS; }

// launch plan of a product inside the resident kernel = the plan of the one-shot kernel (same order of additions)
bool deep_conv_plan(int64_t M_out, int K, int Cin, int Cout, int* NW, int* ZS) {
  if (M_out < 1 || K < 1 || K > KMAX || Cin < 32 || Cin % 32 || Cout < 32 || Cout % 32) return false;
  const Plan2 p = plan2(M_out, K, Cin, Cout);
  if (p.NB != 1 || p.DA != 2 || !p.BD) return false;
  *ZS = p.ZS;
  if (K == 1) {      // one offset: wave 0 of the work item does all the steps whatever NW is (the others add zeros)
    *NW = 4;
    return p.ZS == 1;
  }
  if (p.NW != 4 && p.NW != 8 && p.NW != 16) return false;      // (1 and 2 waves per work item: > 1024 work items)
  *NW = p.NW;
  return true;
}

int64_t g_deep_launches = 0, g_deep_phases = 0;

namespace {
std::mutex g_deep_mu;
struct PinnedRing {
  static constexpr int N = 16;
  static constexpr size_t BYTES = 64 * 1024;
  char* buf[N] = {};
  hipEvent_t ev[N] = {};
  bool used[N] = {};
  int next = 0;
};
PinnedRing g_ring;
}  // namespace

// ops: the phase table (host); runs it on `st`.  d_table: device memory for the table (n * sizeof(DeepOp)), d_sync: a
// zero-able 4 KiB slot, d_stamps: optional device buffer of (max stamp index + 1) u64
int deep_launch(const DeepOp* h_ops, int n, void* d_table, void* d_sync, unsigned long long* d_stamps, hipStream_t st) {
  if (n == 0) return WSIS_OK;
  WSIS_REQUIRE((size_t)n * sizeof(DeepOp) <= PinnedRing::BYTES, "too many phases for one resident launch");
  static int cus = 0;           // one process per GPU (include/wsis_hip.h): the device of the first call
  if (!cus) {
    int dev = 0;
    WSIS_HIP_CHECK(hipGetDevice(&dev));
    WSIS_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    WSIS_HIP_CHECK(hipFuncSetAttribute((const void*)deep_run_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, DEEP_LDS_BYTES));
    const char* e = getenv("WSIS_DEEP_WGS");
    if (e && atoi(e) > 0 && atoi(e) < cus) cus = atoi(e);
  }
  char* staging = nullptr;
  hipEvent_t ev = nullptr;
  {
    std::lock_guard<std::mutex> lock(g_deep_mu);
    const int slot = g_ring.next;
    g_ring.next = (g_ring.next + 1) % PinnedRing::N;
    if (!g_ring.buf[slot]) {
      WSIS_HIP_CHECK(hipHostMalloc((void**)&g_ring.buf[slot], PinnedRing::BYTES, hipHostMallocDefault));
      WSIS_HIP_CHECK(hipEventCreateWithFlags(&g_ring.ev[slot], hipEventDisableTiming));
    }
    if (g_ring.used[slot]) WSIS_HIP_CHECK(hipEventSynchronize(g_ring.ev[slot]));      // the copy of 16 launches ago
    g_ring.used[slot] = true;
    staging = g_ring.buf[slot];
    ev = g_ring.ev[slot];
  }
  memcpy(staging, h_ops, (size_t)n * sizeof(DeepOp));
  WSIS_HIP_CHECK(hipMemcpyAsync(d_table, staging, (size_t)n * sizeof(DeepOp), hipMemcpyHostToDevice, st));
  WSIS_HIP_CHECK(hipEventRecord(ev, st));
  WSIS_HIP_CHECK(hipMemsetAsync(d_sync, 0, sizeof(DeepSync), st));
  static int fence = -1;
  if (fence < 0) {
    const char* e = getenv("WSIS_DEEP_FENCE");
    fence = e ? atoi(e) : 0;
  }
  hipLaunchKernelGGL(deep_run_kernel, dim3((unsigned)cus), dim3(DEEP_THREADS), DEEP_LDS_BYTES, st,
                     static_cast<const DeepOp*>(d_table), n, static_cast<DeepSync*>(d_sync), fence, d_stamps);
  WSIS_LAUNCH_CHECK();
  ++g_deep_launches;
  g_deep_phases += n;
  return WSIS_OK;
}

}  // namespace wsis

extern "C" {
// resident launches / phases issued by this process so far (tests: the resident path was really taken)
int64_t wsis_deep_launches(void) { return wsis::g_deep_launches; }
int64_t wsis_deep_phases(void) { return wsis::g_deep_phases; }
}
