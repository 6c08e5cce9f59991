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
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "common.h"

using namespace wsis;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int TM = 128;       // output rows per workgroup
constexpr int CK = 32;        // input channels staged per step
constexpr int A_STRIDE = 36;  // words; 16-B aligned rows, conflict-free ds_read_b128
constexpr int KG = 16;        // kernel offsets handled per group (LDS slice 16 x 128 ints = 8 KB)

// Workgroups are dispatched round-robin over the 8 XCDs (linear id % 8), each with its own 4 MB L2.  With
// WSIS_XCD_AWARE=1 the work items (tile-major) are dealt so that XCD c owns one contiguous run of tiles, whose
// overlapping gathers then share that XCD's L2.  Measured on the C2 pyramid it is 5-10 % SLOWER than the plain
// round-robin (75.6 -> 81.1 us at level 0, 48.9 -> 54.5 us at level 2): the gathers are not L2-miss bound, and
// contiguous runs lose the statistical load balance of dealing heavy / light tiles over all XCDs.  Default off.
// Bijection lin -> w for any total (q = total / 8, r = total % 8).
constexpr int N_XCD = 8;
__device__ __forceinline__ unsigned xcd_deal(unsigned lin, unsigned total) {
  const unsigned xcd = lin % N_XCD, idx = lin / N_XCD;
  const unsigned q = total / N_XCD, r = total % N_XCD;
  return xcd * q + (xcd < r ? xcd : r) + idx;
}

// ------------------------------------------------------------------------------------------
// forward / dIn kernel
// ------------------------------------------------------------------------------------------
// MATH 0: v_mfma_f32_32x32x2_f32, exact fp32 (a k-ordered fma chain).
// MATH 1: every fp32 operand split into two bf16 terms (hi = bf16(x), lo = bf16(x - hi)) when it is staged to LDS and
//         the product evaluated as lo*hi + hi*lo + hi*hi on v_mfma_f32_32x32x16_bf16 with fp32 accumulation:
//         6 MFMAs of 32 cycles per (32 rows x 32 ch x 32 cols) step instead of 16 of 64 cycles; the dropped lo*lo
//         term and the 16-bit operand mantissa bound the error at ~2^-16 per product (NOT bit-identical to fp32).
template <int NB, bool VEC4, bool DIAG = false, int MATH = 0>
__global__ __launch_bounds__(256, (NB <= 1 ? 5 : (NB <= 2 ? 4 : (NB <= 3 ? 3 : 2)))) void spconv_fwd_kernel(
    const float* __restrict__ X, const int32_t* __restrict__ nbrS, const int32_t* __restrict__ order,
    const float* __restrict__ W, const float* __restrict__ bias, const float* __restrict__ residual,
    float* __restrict__ out, float* __restrict__ partial, int64_t M_in, int64_t M_out, int K, int Cin, int Cout,
    int k_per, int xcd_aware, unsigned long long* __restrict__ dbg = nullptr) {
  // DIAG build only: per-phase cycle sums of wave 0 (s_memtime), written to dbg[blockIdx.x*8 + phase]
  unsigned long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tlast = 0;
  auto stamp = [&](int ph) {
    if (DIAG) {
      const unsigned long long now = __builtin_amdgcn_s_memtime();
      tph[ph] += now - tlast;
      tlast = now;
    }
  };
  if (DIAG) tlast = __builtin_amdgcn_s_memtime();
  // fp32 mode: As [TM][A_STRIDE] floats, Bs [CK][NB*32] floats.
  // bf16 mode: a row of A (tile row) / of B (output column) is 2 terms x 32 channels of bf16 = 128 B + 16 B pad
  //            (the same 144-B pitch as A_STRIDE words, so the ds_read_b128 fragment reads stay conflict-free).
  constexpr int ROWB = A_STRIDE * 4;                                   // 144 bytes
  constexpr int ROWB3 = 3 * 64 + 16;                                   // MATH 2: three terms per column row
  constexpr int KGL = MATH == 2 ? 8 : KG;                              // offsets per table group
  constexpr int BS_BYTES = MATH == 0 ? CK * NB * 32 * 4 : (MATH == 1 ? NB * 32 * ROWB : NB * 32 * ROWB3);
  __shared__ __attribute__((aligned(16))) float As[TM * A_STRIDE];
  __shared__ __attribute__((aligned(16))) unsigned char BsRaw[BS_BYTES];
  float* const Bs = reinterpret_cast<float*>(BsRaw);
  __shared__ int32_t nbT[KGL * TM];   // [offset in group][tile row] -> input row or -1
  __shared__ int32_t rowId[TM];
  __shared__ uint32_t grpMask[4];    // per 32-row slice: bit k set iff some row of the slice uses offset k

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int half = lane >> 5;
  const int l31 = lane & 31;
  unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (xcd_aware) {
    const unsigned per_tile = gridDim.y * gridDim.z;
    const unsigned w = xcd_deal(bx + gridDim.x * (by + gridDim.y * bz), gridDim.x * per_tile);
    bx = w / per_tile;
    const unsigned rest = w - bx * per_tile;
    bz = rest / gridDim.y;
    by = rest - bz * gridDim.y;
  }
  const int64_t tile0 = (int64_t)bx * TM;
  const int col0 = by * (NB * 32);
  const int ncols = min(NB * 32, Cout - col0);

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
  const int k_begin = bz * k_per;
  const int k_end = min(K, k_begin + k_per);

  // per-thread staging coordinates (constant over the walk)
  int a_row[4], a_c4[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int f = tid + 256 * j;
    a_row[j] = f >> 3;
    a_c4[j] = (f & 7) * 4;
  }
  int b_kr[NB], b_c4[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int f = tid + 256 * j;
    b_kr[j] = f / (NB * 8);
    b_c4[j] = (f - b_kr[j] * (NB * 8)) * 4;
  }

  for (int kg0 = k_begin; kg0 < k_end; kg0 += KGL) {
    const int kcount = min(KGL, k_end - kg0);
    __syncthreads();  // previous group's nbT / grpMask readers are done (also publishes rowId)
    // ---- tile slice of the packed gather table: nbrS[k][tile0 + t], coalesced 512-B pieces.  All loads are
    //      issued before any of them is consumed: one memory latency for the whole slice.
    if (tid < TM) {
      // loads in batches of 8 offsets (independent, one latency per batch); batches beyond kcount are skipped,
      // which matters when the offsets are split over blockIdx.z (kcount is 1..6 there)
      int32_t v[KGL];
#pragma unroll
      for (int g8 = 0; g8 < KGL; g8 += 8) {
        if (g8 < kcount) {
#pragma unroll
          for (int kk = g8; kk < g8 + 8; ++kk) {
            const bool ok = kk < kcount && my_row >= 0 && nbrS != nullptr;
            v[kk] = nbrS ? nbrS[ok ? (int64_t)(kg0 + kk) * M_out + tile0 + tid : 0] : 0;
            if (!ok) v[kk] = (nbrS == nullptr && kk < kcount) ? my_row : -1;
          }
        } else {
#pragma unroll
          for (int kk = g8; kk < g8 + 8; ++kk) v[kk] = -1;
        }
      }
      uint32_t bits_lo = 0u, bits_hi = 0u;
#pragma unroll
      for (int g8 = 0; g8 < KGL; g8 += 8) {
        if (g8 < kcount) {
#pragma unroll
          for (int kk = g8; kk < g8 + 8; ++kk) {
            nbT[kk * TM + tid] = v[kk];
            const unsigned long long b = __ballot(v[kk] >= 0);
            if ((uint32_t)b) bits_lo |= 1u << kk;
            if ((uint32_t)(b >> 32)) bits_hi |= 1u << kk;
          }
        }
      }
      if (lane == 0) {
        grpMask[wave * 2] = bits_lo;
        grpMask[wave * 2 + 1] = bits_hi;
      }
    }
    __syncthreads();
    stamp(0);   // prologue: table slice
    const uint32_t my_mask = grpMask[wave];
    uint32_t tile_mask = grpMask[0] | grpMask[1] | grpMask[2] | grpMask[3];
    if (tile_mask == 0u) continue;
    {
      // Tiles with many active offsets are the tail of the launch (all workgroups are resident in one round and it
      // lasts as long as its slowest tile): their waves get a higher priority in the SIMD's MFMA / VALU arbitration
      // against the light workgroups sharing the SIMD (level 0: 75.5 -> 70.7 us, 64->32 layer 134.8 -> 124.5 us;
      // thresholds 12/8/4 and 10/6/3 measure the same, priority by REMAINING offsets per step measured worse).
      const int act = __builtin_popcount(tile_mask) * (KG / KGL);
      if (act >= 14)
        __builtin_amdgcn_s_setprio(3);
      else if (act >= 10)
        __builtin_amdgcn_s_setprio(2);
      else if (act >= 6)
        __builtin_amdgcn_s_setprio(1);
      else
        __builtin_amdgcn_s_setprio(0);
    }

    // ---- pipelined walk over (active offset, channel chunk) steps, register prefetch TWO steps ahead.
    // fetch(): branch-free -- every lane always issues its loads (row 0 / channel 0 when masked) so that the
    // gathers of a step are independent and in flight together; masking to zero happens when the registers are
    // written to LDS, which keeps the loads in flight under the MFMAs of the two steps before.
    struct Stage {
      f32x4 ra[4];
      f32x4 rb[NB];
      uint32_t ok_bits;   // bit j: ra[j] valid, bit 8+j: rb[j] valid
      int kk, cin_here;
    };
    auto fetch = [&](Stage& st, int kk, int ci0) {
      const int cin_here = min(CK, Cin - ci0);
      st.kk = kk;
      st.cin_here = cin_here;
      uint32_t okb = 0u;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int32_t g = nbT[kk * TM + a_row[j]];
        const bool ok = g >= 0 && a_c4[j] < cin_here;
        const float* src = X + (int64_t)(ok ? g : 0) * Cin + ci0 + (ok ? a_c4[j] : 0);
        if (VEC4) {
          st.ra[j] = *reinterpret_cast<const f32x4*>(src);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const bool oke = ok && a_c4[j] + e < cin_here;
            const float t = src[oke ? e : 0];
            st.ra[j][e] = oke ? t : 0.0f;
          }
        }
        okb |= (ok ? 1u : 0u) << j;
      }
      const float* Wk = W + (int64_t)(kg0 + kk) * Cin * Cout + col0;
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const bool okr = b_kr[j] < cin_here;
        const float* src = Wk + (int64_t)(ci0 + (okr ? b_kr[j] : 0)) * Cout;
        if (VEC4) {
          const bool okc = b_c4[j] + 3 < ncols;
          st.rb[j] = *reinterpret_cast<const f32x4*>(src + (okc ? b_c4[j] : 0));
          okb |= ((okr && okc) ? 1u : 0u) << (8 + j);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const bool okc = okr && b_c4[j] + e < ncols;
            const float t = src[okc ? b_c4[j] + e : 0];
            st.rb[j][e] = okc ? t : 0.0f;
          }
          okb |= 1u << (8 + j);
        }
      }
      st.ok_bits = okb;
    };
    // step iterator over (active offset, chunk)
    int it_kk = __builtin_ctz(tile_mask);
    uint32_t it_mask = tile_mask & (tile_mask - 1);
    int it_ci = 0;
    bool it_valid = true;
    auto advance = [&]() {
      it_ci += CK;
      if (it_ci >= Cin) {
        it_ci = 0;
        if (it_mask) {
          it_kk = __builtin_ctz(it_mask);
          it_mask &= it_mask - 1;
        } else {
          it_valid = false;
        }
      }
    };
    auto compute = [&](Stage& st, bool prefetch_more) {
      const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
      if (MATH == 0 || MATH == 2) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          *reinterpret_cast<f32x4*>(&As[a_row[j] * A_STRIDE + a_c4[j]]) = ((st.ok_bits >> j) & 1u) ? st.ra[j] : zero4;
      }
      if (MATH == 0) {
#pragma unroll
        for (int j = 0; j < NB; ++j)
          *reinterpret_cast<f32x4*>(&Bs[b_kr[j] * (NB * 32) + b_c4[j]]) =
              ((st.ok_bits >> (8 + j)) & 1u) ? st.rb[j] : zero4;
      } else if (MATH == 2) {
#pragma unroll
        for (int j = 0; j < NB; ++j) {   // W[k = b_kr][4 columns] -> column rows of three 32-channel bf16 terms
          const f32x4 v = ((st.ok_bits >> (8 + j)) & 1u) ? st.rb[j] : zero4;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const __bf16 t1 = (__bf16)v[e];
            const float r1 = v[e] - (float)t1;
            const __bf16 t2 = (__bf16)r1;
            const __bf16 t3 = (__bf16)(r1 - (float)t2);
            unsigned char* dst = BsRaw + (b_c4[j] + e) * ROWB3 + b_kr[j] * 2;
            *reinterpret_cast<__bf16*>(dst) = t1;
            *reinterpret_cast<__bf16*>(dst + 64) = t2;
            *reinterpret_cast<__bf16*>(dst + 128) = t3;
          }
        }
      } else {
        unsigned char* const A8 = reinterpret_cast<unsigned char*>(As);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x4 v = ((st.ok_bits >> j) & 1u) ? st.ra[j] : zero4;
          bf16x4 hi, lo;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            hi[e] = (__bf16)v[e];
            lo[e] = (__bf16)(v[e] - (float)hi[e]);
          }
          unsigned char* dst = A8 + a_row[j] * ROWB + a_c4[j] * 2;
          *reinterpret_cast<bf16x4*>(dst) = hi;
          *reinterpret_cast<bf16x4*>(dst + 64) = lo;
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {   // W[k = b_kr][4 columns] -> transposed: column rows, channel k inside
          const f32x4 v = ((st.ok_bits >> (8 + j)) & 1u) ? st.rb[j] : zero4;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const __bf16 hi = (__bf16)v[e];
            const __bf16 lo = (__bf16)(v[e] - (float)hi);
            unsigned char* dst = BsRaw + (b_c4[j] + e) * ROWB + b_kr[j] * 2;
            *reinterpret_cast<__bf16*>(dst) = hi;
            *reinterpret_cast<__bf16*>(dst + 64) = lo;
          }
        }
      }
      const int cur_kk = st.kk;
      const int cin_here = st.cin_here;
      stamp(1);   // wait for the stage's loads + LDS write
      __syncthreads();
      stamp(2);   // barrier 1
      if (prefetch_more) {   // refill this register stage with the step two ahead
        fetch(st, it_kk, it_ci);
        advance();
      }
      stamp(3);   // issue prefetch
      if ((my_mask >> cur_kk) & 1u) {
        if (MATH == 0) {
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
        } else if (MATH == 2) {
          // the lane's 16 fp32 values (channels half*16 .. +15 of its row); MFMA j takes its channels half*16 + j*8
          // .. +7 as the lane's 8 k-slots, B rows are laid out the same way
          const float* arow = &As[(wave * 32 + l31) * A_STRIDE + half * 16];
          bf16x8 a1[2], a2[2], a3[2];
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const f32x4 u0 = *reinterpret_cast<const f32x4*>(arow + 8 * j);
            const f32x4 u1 = *reinterpret_cast<const f32x4*>(arow + 8 * j + 4);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float x = e < 4 ? u0[e] : u1[e - 4];
              const __bf16 t1 = (__bf16)x;
              const float r1 = x - (float)t1;
              const __bf16 t2 = (__bf16)r1;
              a1[j][e] = t1;
              a2[j][e] = t2;
              a3[j][e] = (__bf16)(r1 - (float)t2);
            }
          }
#pragma unroll
          for (int cb = 0; cb < NB; ++cb) {
            const unsigned char* brow = BsRaw + (cb * 32 + l31) * ROWB3 + half * 32;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              if (j * 8 < cin_here) {      // channels of both halves beyond cin_here are zero
                const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(brow + j * 16);
                const bf16x8 b2 = *reinterpret_cast<const bf16x8*>(brow + 64 + j * 16);
                const bf16x8 b3 = *reinterpret_cast<const bf16x8*>(brow + 128 + j * 16);
                acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[j], b3, acc[cb], 0, 0, 0);
                acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[j], b1, acc[cb], 0, 0, 0);
                acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[j], b2, acc[cb], 0, 0, 0);
                acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[j], b1, acc[cb], 0, 0, 0);
                acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[j], b2, acc[cb], 0, 0, 0);
                acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[j], b1, acc[cb], 0, 0, 0);
              }
            }
          }
        } else {
          // MFMA j covers channels j*16 .. j*16+15; lane (row / column l31, half) supplies the 8 channels
          // j*16 + half*8 .. +7 of its row -- one ds_read_b128 per term.  Channels >= cin_here are zero in A and B.
          const unsigned char* arow = reinterpret_cast<const unsigned char*>(As) + (wave * 32 + l31) * ROWB + half * 16;
          bf16x8 a_hi[2], a_lo[2];
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            a_hi[j] = *reinterpret_cast<const bf16x8*>(arow + j * 32);
            a_lo[j] = *reinterpret_cast<const bf16x8*>(arow + 64 + j * 32);
          }
#pragma unroll
          for (int cb = 0; cb < NB; ++cb) {
            const unsigned char* brow = BsRaw + (cb * 32 + l31) * ROWB + half * 16;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              if (j * 16 < cin_here) {
                const bf16x8 b_hi = *reinterpret_cast<const bf16x8*>(brow + j * 32);
                const bf16x8 b_lo = *reinterpret_cast<const bf16x8*>(brow + 64 + j * 32);
                acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo[j], b_hi, acc[cb], 0, 0, 0);
                acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi[j], b_lo, acc[cb], 0, 0, 0);
                acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi[j], b_hi, acc[cb], 0, 0, 0);
              }
            }
          }
        }
      }
      stamp(4);   // fragment reads + MFMA issue
      __syncthreads();
      stamp(5);   // barrier 2
      if (DIAG) tph[7] += 1;
    };

    Stage s0;
    fetch(s0, it_kk, it_ci);
    advance();
    for (;;) {
      const bool pf = it_valid;
      compute(s0, pf);
      if (!pf) break;
    }
  }

  if (DIAG && dbg && tid == 0) {
    stamp(6);
    for (int i = 0; i < 8; ++i) dbg[((int64_t)bz * gridDim.x + bx) * 8 + i] = tph[i];
  }
  // ---- epilogue: C/D map of the 32x32 tile: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*half
  float* dst = (gridDim.z > 1) ? partial + (int64_t)bz * M_out * Cout : out;
  const bool final_pass = gridDim.z == 1;
#pragma unroll
  for (int cb = 0; cb < NB; ++cb) {
    const int c = cb * 32 + l31;
    if (c >= ncols) continue;
    const float bv = (final_pass && bias) ? bias[col0 + c] : 0.0f;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int rr = (reg & 3) + 8 * (reg >> 2) + 4 * half;
      const int32_t r = rowId[wave * 32 + rr];
      if (r < 0) continue;
      const int64_t o = (int64_t)r * Cout + col0 + c;
      float v = acc[cb][reg] + bv;
      if (final_pass && residual) v += residual[o];
      dst[o] = v;
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

// the same sum, four channels per thread (Cout % 4 == 0, 16-byte aligned buffers): same order of additions per element
__global__ void spconv_reduce4_kernel(const float4* __restrict__ partial, const float4* __restrict__ bias,
                                      const float4* __restrict__ residual, float4* __restrict__ out, int64_t total4,
                                      int cout4, int ksplit) {
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total4;
       t += (int64_t)gridDim.x * blockDim.x) {
    float4 s = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll 4
    for (int z = 0; z < ksplit; ++z) {
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

// ---- live kernel timing for bench.py's roofline: HIP events recorded on the launch stream directly around
// the dominant kernels (not around the host wrapper), enabled by wsis_prof_enable().
// output-channel blocks (of 32) one workgroup owns.  Levels with <= 100 row tiles run one block per workgroup:
// more workgroups and fewer MFMAs per step on each workgroup's latency chain (measured on the C2 pyramid:
// 96 ch 55->48 us, 128 ch 44->33 us, 160 ch 41->25 us).  WSIS_FWD_NB_SMALL / WSIS_FWD_SMALL_TILES: tuning knobs.
int fwd_nb(int64_t M_out, int Cout) {
  static int nb_small = -1, small_tiles = -1;
  if (nb_small < 0) {
    const char* e = tune_env("WSIS_FWD_NB_SMALL");
    nb_small = e ? atoi(e) : 1;
    e = tune_env("WSIS_FWD_SMALL_TILES");
    small_tiles = e ? atoi(e) : 100;
  }
  const int nblk = (Cout + 31) / 32;
  int NB = nblk <= 5 ? nblk : 4;
  if (ceil_div(M_out, TM) <= small_tiles && NB > nb_small) {
    // balanced groups: e.g. 5 blocks with cap 2 -> 3 groups of 2,2,1 ; 4 blocks cap 3 -> 2 groups of 2
    const int groups = (nblk + nb_small - 1) / nb_small;
    NB = (nblk + groups - 1) / groups;
  }
  return NB;
}

int fwd_ksplit(int64_t M_out, int K, int Cout) {
  static int target = -1;   // WSIS_KSPLIT_TARGET: workgroups aimed for when splitting the offsets (tuning knob)
  if (target < 0) {
    const char* e = tune_env("WSIS_KSPLIT_TARGET");
    target = e ? atoi(e) : 1024;
  }
  const int NB = fwd_nb(M_out, Cout);
  const int64_t wgs = ceil_div(M_out, TM) * ceil_div(Cout, NB * 32);
  if (wgs >= target / 2 || K == 1) return 1;
  int64_t ks = ceil_div(target, wgs);
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

// the same for up to 16 tables in one launch (all tables of a pyramid)
constexpr int PACK_MAX = 16;
struct PackBatch {
  const int32_t* nbr[PACK_MAX];
  const int32_t* order[PACK_MAX];
  int32_t* out[PACK_MAX];
  int64_t M[PACK_MAX];
  int64_t base[PACK_MAX + 1];   // first element (K*M entries per table) in the concatenated index space
  int n;
};

__global__ void rulebook_pack_batch_kernel(PackBatch b) {
  const int64_t total = b.base[b.n];
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    int t = 0;
#pragma unroll 1
    while (t + 1 < b.n && b.base[t + 1] <= e) ++t;
    const int64_t u = e - b.base[t], M = b.M[t];
    const int64_t k = u / M;
    const int64_t c = u - k * M;
    b.out[t][u] = b.nbr[t][k * M + b.order[t][c]];
  }
}

// ------------------------------------------------------------------------------------------
// dW kernel: grid (work item, ci-block); a work item is (offset k, row chunk j of n_k).  The MFMA A operand is
// X^T (lane = input channel, the two k-slots = two tile rows), B is dY -- both are coalesced 128-B row segments
// loaded straight from global memory, no LDS transposition.
// A wave scans 64 tile rows at a time (one coalesced table read + ballot, the next block's read already in
// flight), then walks only the ACTIVE rows two at a time (scalar bit scan + v_readlane), so inactive rows cost no
// vector work.  Row bases are computed on the scalar unit (the row of a k-slot is wave-uniform) and addressed
// as 32-bit byte offsets from the tensor base -- the vector unit only selects the half's offset and adds the
// lane's channel (the first version spent ~40 VALU ops per MFMA on 64-bit address arithmetic and was issue-bound:
// switching the MFMAs off saved 25 %, making every gather cache-hot saved nothing).
//
// Work split: the 27 offsets of a 3x3x3 submanifold table are very unevenly filled (C2 scene, level 0: the
// centre offset pairs every row, a corner offset 1.5 % of them), so the rows of offset k are cut into n_k chunks
// with n_k proportional to a static weight of the offset type (centre 4 : face 3 : edge 2 : corner 1); other
// kernels (2x2x2 strided, 1x1x1) use equal chunks.  Each item writes one [ci_pad][Cout] slab; dw_reduce_kernel
// adds the slabs of an offset in item order (fixed order -> deterministic).
constexpr int DW_MAX_K = 128;
struct DwPlan {
  int begin[DW_MAX_K + 1];   // items of offset k: [begin[k], begin[k+1])
};

template <int NBO, bool DIAG = false>
__global__ __launch_bounds__(256) void spconv_dw_kernel(
    const float* __restrict__ X, const int32_t* __restrict__ nbrS, const int32_t* __restrict__ order,
    const float* __restrict__ dY, float* __restrict__ partial, int64_t M_in, int64_t M_out, int K, int Cin,
    int Cout, DwPlan plan, int n_cib, int co0, unsigned long long* __restrict__ dbg = nullptr) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // [4][NBO][1024]
  // DIAG build only: per-phase cycle sums (s_memtime) of every wave, dbg[(item*4 + wave)*8 + phase]
  unsigned long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tlast = 0;
  auto stamp = [&](int ph) {
    if (DIAG) {
      const unsigned long long now = __builtin_amdgcn_s_memtime();
      tph[ph] += now - tlast;
      tlast = now;
    }
  };
  if (DIAG) tlast = __builtin_amdgcn_s_memtime();
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int half = lane >> 5;
  const int l31 = lane & 31;
  const int item = blockIdx.x;
  const int cib = blockIdx.y;
  int k = 0;
  while (k + 1 < K && plan.begin[k + 1] <= item) ++k;
  const int n_k = plan.begin[k + 1] - plan.begin[k];
  const int j = item - plan.begin[k];
  const int ci = cib * 32 + l31;
  const bool ci_ok = ci < Cin;

  // the 64-row blocks of the table are dealt round-robin: block b belongs to chunk b % n_k (every chunk samples
  // the whole scene, so the chunks of one offset carry the same load whatever the local density), and inside the
  // workgroup the chunk's blocks go round-robin to the 4 waves
  const int64_t n_blocks_all = (M_out + 63) >> 6;
  const int64_t n_blocks = n_blocks_all > j ? (n_blocks_all - j + n_k - 1) / n_k : 0;   // blocks j, j+n_k, ...

  f32x16 acc[NBO];
#pragma unroll
  for (int cb = 0; cb < NBO; ++cb)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[cb][i] = 0.0f;

  const int32_t* nbk = nbrS ? nbrS + (int64_t)k * M_out : nullptr;
  // lane parts of the byte offsets (clamped to a valid column; out-of-range lanes are zeroed after the load)
  const uint32_t a_lane = (uint32_t)(ci_ok ? ci : 0) * 4u;
  uint32_t b_lane[NBO];
  bool cob_ok[NBO];
#pragma unroll
  for (int cb = 0; cb < NBO; ++cb) {
    const int co = co0 + cb * 32 + l31;
    cob_ok[cb] = co < Cout;
    b_lane[cb] = (uint32_t)(cob_ok[cb] ? co : 0) * 4u;
  }
  const uint32_t a_pitch = (uint32_t)Cin * 4u, b_pitch = (uint32_t)Cout * 4u;
  const char* Xb = reinterpret_cast<const char*>(X);
  const char* Yb = reinterpret_cast<const char*>(dY);

  constexpr int U = (NBO <= 2) ? 8 : 4;   // MFMAs (row pairs) per group
  constexpr int QCAP = 2 * U + 64;
  // Per-wave queue of active rows, as byte offsets (x row, dY row).  A scanned block appends its active rows with
  // one ballot + mbcnt + LDS write (vector compaction: a scalar ctz/readlane walk cost ~1 us per 16 pairs and was
  // the largest phase of the first version); whenever 2U rows are queued a full group of U MFMAs runs off the
  // top of the queue, so sparsely filled offsets do not issue mostly-empty groups.  The queue order depends only
  // on the table and the plan -> deterministic summation order.
  __shared__ uint2 queue[4][QCAP];
  uint2* q = queue[wave];
  int qn = 0;   // wave-uniform

  auto run_group = [&](int base, int count, auto partial_tag) {
    constexpr bool PARTIAL = decltype(partial_tag)::value;
    float a[U];
    float b[U][NBO];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = 2 * u + half;
      ok[u] = !PARTIAL || e < count;
      uint2 off = make_uint2(0u, 0u);
      if (ok[u]) off = q[base + e];
      a[u] = *reinterpret_cast<const float*>(Xb + (off.x + a_lane));
#pragma unroll
      for (int cb = 0; cb < NBO; ++cb) b[u][cb] = *reinterpret_cast<const float*>(Yb + (off.y + b_lane[cb]));
    }
    stamp(3);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (PARTIAL && 2 * u >= count) continue;   // wave-uniform
      const float av = (ok[u] && ci_ok) ? a[u] : 0.0f;
#pragma unroll
      for (int cb = 0; cb < NBO; ++cb) {
        const float bv = (ok[u] && cob_ok[cb]) ? b[u][cb] : 0.0f;
        acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[cb], 0, 0, 0);
      }
    }
    stamp(4);
  };

  // (contiguous row ranges left both the chunks of an offset and the waves of a workgroup unevenly loaded: the
  // barrier before the cross-wave reduce alone was 28 % of the mean wave time)
  auto scan = [&](int64_t blk, int32_t& yr, int32_t& nb) {
    const int64_t t = ((int64_t)j + blk * n_k) * 64 + lane;
    yr = 0;
    nb = -1;
    if (t < M_out) {
      yr = order ? order[t] : (int32_t)t;
      nb = nbk ? nbk[t] : yr;
    }
  };
  int32_t yr_n = 0, nb_n = -1;
  if (wave < n_blocks) scan(wave, yr_n, nb_n);
  stamp(0);
  for (int64_t blk = wave; blk < n_blocks; blk += 4) {
    const int32_t yr = yr_n, nb = nb_n;
    if (blk + 4 < n_blocks) scan(blk + 4, yr_n, nb_n);   // next block's table entries fly during this block's work
    const unsigned long long m = __ballot(nb >= 0);
    stamp(1);
    if (DIAG) tph[6] += 1;
    if (m) {
      const int pos = qn + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
      if (nb >= 0) q[pos] = make_uint2((uint32_t)nb * a_pitch, (uint32_t)yr * b_pitch);
      qn += __builtin_popcountll(m);
    }
    stamp(2);
    while (qn >= 2 * U) {
      if (DIAG) tph[7] += 1;
      qn -= 2 * U;
      run_group(qn, 2 * U, std::false_type{});
    }
  }
  if (qn > 0) {
    if (DIAG) tph[7] += 1;
    run_group(0, qn, std::true_type{});
  }

#pragma unroll
  for (int cb = 0; cb < NBO; ++cb)
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int row = (reg & 3) + 8 * (reg >> 2) + 4 * half;
      red[(wave * NBO + cb) * 1024 + row * 32 + l31] = acc[cb][reg];
    }
  __syncthreads();
  if (DIAG && dbg) {
    stamp(5);
    if (lane == 0)
      for (int i = 0; i < 8; ++i) dbg[(((int64_t)item * gridDim.y + cib) * 4 + wave) * 8 + i] = tph[i];
  }
  const int ci_pad = n_cib * 32;
  float* dst = partial + ((int64_t)item * ci_pad + cib * 32) * (int64_t)Cout;
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

__global__ void dw_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dW, DwPlan plan, int K,
                                 int Cin, int Cout, int ci_pad) {
  const int64_t total = (int64_t)K * Cin * Cout;
  const int64_t slab = (int64_t)ci_pad * Cout;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int co = (int)(t % Cout);
    const int64_t u = t / Cout;
    const int ci = (int)(u % Cin);
    const int k = (int)(u / Cin);
    const float* src = partial + (int64_t)ci * Cout + co;
    float s = 0.0f;
    const int e = plan.begin[k + 1];
#pragma unroll 4
    for (int it = plan.begin[k]; it < e; ++it) s += src[it * slab];
    dW[t] = s;
  }
}

// four output channels per thread (Cout % 4 == 0, aligned dW): the slab walk issues eight 16-byte loads at a time
__global__ void dw_reduce4_kernel(const float4* __restrict__ partial, float4* __restrict__ dW, DwPlan plan, int K,
                                  int Cin, int cout4, int ci_pad) {
  const int64_t total4 = (int64_t)K * Cin * cout4;
  const int64_t slab4 = (int64_t)ci_pad * cout4;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total4;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int co = (int)(t % cout4);
    const int64_t u = t / cout4;
    const int ci = (int)(u % Cin);
    const int k = (int)(u / Cin);
    const float4* src = partial + (int64_t)ci * cout4 + co;
    float4 s = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const int e = plan.begin[k + 1];
#pragma unroll 8
    for (int it = plan.begin[k]; it < e; ++it) {
      const float4 v = src[it * slab4];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    dW[t] = s;
  }
}

// items per offset: base chunks (~64 at the large levels, >= 256 rows each) scaled by the offset-type weight
int dw_base_chunks(int64_t M_out) {
  static int div = -1;   // WSIS_DW_DIV: rows per unit-weight chunk below which a level gets fewer chunks (tuning knob)
  if (div < 0) {
    const char* e = tune_env("WSIS_DW_DIV");
    div = e ? atoi(e) : 256;
    if (div < 64) div = 64;
  }
  int64_t c = ceil_div(M_out, div);
  if (c > 32) c = 32;
  if (c < 1) c = 1;
  return (int)c;
}

void dw_make_plan(int64_t M_out, int K, DwPlan& plan) {
  const int base = dw_base_chunks(M_out);
  const int64_t max_chunks = M_out <= 0 ? 1 : ceil_div(M_out, 256);   // >= 64 rows per wave
  plan.begin[0] = 0;
  for (int k = 0; k < K; ++k) {
    int w = 2;   // equal chunks (2 * base ~ 64) for anything but the 3x3x3 table
    if (K == 27) {
      const int nz = (k / 9 != 1) + ((k / 3) % 3 != 1) + (k % 3 != 1);   // 0 centre, 1 face, 2 edge, 3 corner
      w = 4 - nz;
    }
    int64_t n = (int64_t)w * base;
    if (n > max_chunks) n = max_chunks;
    if (n < 1) n = 1;
    plan.begin[k + 1] = plan.begin[k] + (int)n;
  }
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

int wsis_rulebook_pack_batch(int32_t n, const void* const* h_nbr, const void* const* h_order,
                             void* const* h_nbr_packed, const int64_t* h_M, const int32_t* h_K, void* stream) {
  WSIS_REQUIRE(n >= 0 && n <= PACK_MAX, "at most 16 tables per call");
  if (n == 0) return WSIS_OK;
  WSIS_REQUIRE(h_nbr && h_order && h_nbr_packed && h_M && h_K, "null pointer");
  PackBatch b;
  b.n = 0;
  b.base[0] = 0;
  for (int t = 0; t < n; ++t) {
    WSIS_REQUIRE(h_M[t] >= 0 && h_K[t] >= 1, "bad sizes");
    if (h_M[t] == 0) continue;
    WSIS_REQUIRE(h_nbr[t] && h_order[t] && h_nbr_packed[t], "null table pointer");
    b.nbr[b.n] = static_cast<const int32_t*>(h_nbr[t]);
    b.order[b.n] = static_cast<const int32_t*>(h_order[t]);
    b.out[b.n] = static_cast<int32_t*>(h_nbr_packed[t]);
    b.M[b.n] = h_M[t];
    b.base[b.n + 1] = b.base[b.n] + h_M[t] * h_K[t];
    ++b.n;
  }
  if (b.n == 0) return WSIS_OK;
  hipLaunchKernelGGL(rulebook_pack_batch_kernel, dim3(grid_for(b.base[b.n], 256)), dim3(256), 0, as_stream(stream), b);
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
  if (d_nbr && spconv_in_supported(K, Cin, Cout) && (reinterpret_cast<uintptr_t>(d_X) & 7) == 0) {
    // the 6 -> 32 input convolution: im2col on the matrix cores, operands in registers (csrc/spconv_in.hip)
    hipStream_t st_in = as_stream(stream);
    ProfScope prof_in(0, st_in, /*exact_events=*/true);
    const int rc = spconv_in_launch(d_X, d_nbr, d_order, d_W, d_bias, d_residual, d_out, M_out, prof_in.ka(), prof_in.kb(), st_in);
    prof_in.stop();
    return rc;
  }
  const int NB = fwd_nb(M_out, Cout);
  const int ksplit = fwd_ksplit(M_out, K, Cout);
  const int k_per = (K + ksplit - 1) / ksplit;
  const int kz = (K + k_per - 1) / k_per;   // blocks along z that own at least one offset
  float* partial = nullptr;
  if (kz > 1) {
    WSIS_REQUIRE(d_ws && ws_bytes >= (int64_t)kz * M_out * Cout * (int64_t)sizeof(float), "workspace too small");
    partial = static_cast<float*>(d_ws);
  }
  hipStream_t st = as_stream(stream);
  const bool vec_ok = (Cin % 4 == 0) && (Cout % 4 == 0) && ((reinterpret_cast<uintptr_t>(d_X) & 15) == 0) &&
                      ((reinterpret_cast<uintptr_t>(d_W) & 15) == 0);
  const dim3 grid((unsigned)ceil_div(M_out, TM), (unsigned)ceil_div(Cout, NB * 32), (unsigned)kz);
  static int xcd_aware = -1;
  if (xcd_aware < 0) {
    const char* e = tune_env("WSIS_XCD_AWARE");
    xcd_aware = e ? atoi(e) : 0;
  }
  // WSIS_CONV_MATH (read per call, so a process can switch): 0 = exact fp32 MFMA (default), 1 = split-bf16 products
  const char* math_env = getenv("WSIS_CONV_MATH");
  const int conv_math = math_env ? atoi(math_env) : 0;
  ProfScope prof(0, st);
#define WSIS_FWD_CASE(n)                                                                          \
  case n:                                                                                         \
    if (vec_ok && conv_math == 1)                                                                 \
      hipLaunchKernelGGL((spconv_fwd_kernel<n, true, false, 1>), grid, dim3(256), 0, st, d_X, d_nbr, d_order, d_W, \
                         d_bias, d_residual, d_out, partial, M_in, M_out, K, Cin, Cout, k_per, xcd_aware); \
    else if (vec_ok && conv_math == 2)                                                            \
      hipLaunchKernelGGL((spconv_fwd_kernel<n, true, false, 2>), grid, dim3(256), 0, st, d_X, d_nbr, d_order, d_W, \
                         d_bias, d_residual, d_out, partial, M_in, M_out, K, Cin, Cout, k_per, xcd_aware); \
    else if (vec_ok)                                                                              \
      hipLaunchKernelGGL((spconv_fwd_kernel<n, true>), grid, dim3(256), 0, st, d_X, d_nbr, d_order, d_W,  \
                         d_bias, d_residual, d_out, partial, M_in, M_out, K, Cin, Cout, k_per, xcd_aware); \
    else                                                                                          \
      hipLaunchKernelGGL((spconv_fwd_kernel<n, false>), grid, dim3(256), 0, st, d_X, d_nbr, d_order, d_W, \
                         d_bias, d_residual, d_out, partial, M_in, M_out, K, Cin, Cout, k_per, xcd_aware); \
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
  prof.stop();
  WSIS_LAUNCH_CHECK();
  if (kz > 1) {
    const uintptr_t al = reinterpret_cast<uintptr_t>(partial) | reinterpret_cast<uintptr_t>(d_bias) |
                         reinterpret_cast<uintptr_t>(d_residual) | reinterpret_cast<uintptr_t>(d_out);
    if ((Cout & 3) == 0 && (al & 15) == 0) {
      const int64_t total4 = M_out * Cout / 4;
      hipLaunchKernelGGL(spconv_reduce4_kernel, dim3(grid_for(total4, 256)), dim3(256), 0, st,
                         reinterpret_cast<const float4*>(partial), reinterpret_cast<const float4*>(d_bias),
                         reinterpret_cast<const float4*>(d_residual), reinterpret_cast<float4*>(d_out), total4,
                         Cout / 4, kz);
    } else {
      hipLaunchKernelGGL(spconv_reduce_kernel, dim3(grid_for(M_out * Cout, 256)), dim3(256), 0, st, partial, d_bias,
                         d_residual, d_out, M_out, Cout, kz);
    }
    prof.tail();
    WSIS_LAUNCH_CHECK();
  }
  return WSIS_OK;
}

int wsis_prof_enable(int32_t on) {
  if (on && !g_prof_on && g_prof[0].empty() && g_prof[1].empty() && g_prof[2].empty()) {      // a new session: the stamp buffers of the last one
    for (void* b : g_prof_bufs) (void)hipFree(b);
    g_prof_bufs.clear();
  }
  g_prof_on = on != 0;
  return WSIS_OK;
}

int wsis_prof_summary(int32_t which, double* total_ms, int64_t* launches) {
  WSIS_REQUIRE(which >= 0 && which < 3 && total_ms && launches, "bad args");
  double ms = 0.0;
  for (ProfRec& r : g_prof[which]) {
    if (r.d_stamps) {
      unsigned long long t[2];
      WSIS_HIP_CHECK(hipDeviceSynchronize());
      WSIS_HIP_CHECK(hipMemcpy(&t[0], r.d_stamps + r.s0, 8, hipMemcpyDeviceToHost));
      WSIS_HIP_CHECK(hipMemcpy(&t[1], r.d_stamps + r.s1, 8, hipMemcpyDeviceToHost));
      ms += (double)(t[1] - t[0]) * 1e-5;      // 100 MHz ticks -> ms
      continue;
    }
    hipEvent_t last = r.c ? r.c : r.b;
    WSIS_HIP_CHECK(hipEventSynchronize(last));
    float t = 0.0f;
    WSIS_HIP_CHECK(hipEventElapsedTime(&t, r.a, last));
    ms += t;
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
    if (r.c) (void)hipEventDestroy(r.c);
  }
  *total_ms = ms;
  *launches = (int64_t)g_prof[which].size();
  g_prof[which].clear();
  return WSIS_OK;
}

int wsis_prof_records(int32_t which, double* h_main_ms, double* h_total_ms, int64_t cap, int64_t* n) {
  WSIS_REQUIRE(which >= 0 && which < 3 && h_main_ms && h_total_ms && n && cap >= 0, "bad args");
  WSIS_REQUIRE((int64_t)g_prof[which].size() <= cap, "record buffer too small");
  int64_t i = 0;
  for (ProfRec& r : g_prof[which]) {
    if (r.d_stamps) {
      unsigned long long t[3];
      WSIS_HIP_CHECK(hipDeviceSynchronize());
      WSIS_HIP_CHECK(hipMemcpy(&t[0], r.d_stamps + r.s0, 8, hipMemcpyDeviceToHost));
      WSIS_HIP_CHECK(hipMemcpy(&t[1], r.d_stamps + r.sm, 8, hipMemcpyDeviceToHost));
      WSIS_HIP_CHECK(hipMemcpy(&t[2], r.d_stamps + r.s1, 8, hipMemcpyDeviceToHost));
      h_main_ms[i] = (double)(t[1] - t[0]) * 1e-5;
      h_total_ms[i] = (double)(t[2] - t[0]) * 1e-5;
      ++i;
      continue;
    }
    hipEvent_t last = r.c ? r.c : r.b;
    WSIS_HIP_CHECK(hipEventSynchronize(last));
    float t0 = 0.0f, t1 = 0.0f;
    WSIS_HIP_CHECK(hipEventElapsedTime(&t0, r.a, r.b));
    WSIS_HIP_CHECK(hipEventElapsedTime(&t1, r.a, last));
    h_main_ms[i] = t0;
    h_total_ms[i] = t1;
    ++i;
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
    if (r.c) (void)hipEventDestroy(r.c);
  }
  *n = i;
  g_prof[which].clear();
  return WSIS_OK;
}

// diagnostic (not part of the ABI header): VEC4 kernels with phase stamps, dbg[tiles*kz*8]
int wsis_debug_spconv_diag(const float* d_X, const int32_t* d_nbr, const int32_t* d_order, const float* d_W,
                           float* d_out, float* d_partial, int64_t M_out, int32_t K, int32_t Cin, int32_t Cout,
                           int32_t kz, unsigned long long* d_dbg, void* stream) {
  WSIS_REQUIRE(Cin % 4 == 0 && Cout % 32 == 0 && Cout <= 160, "diag supports Cout in {32..160}, Cin%4==0");
  const int nb = Cout / 32;
  const int k_per = (K + kz - 1) / kz;
  const dim3 grid((unsigned)ceil_div(M_out, TM), 1, (unsigned)kz);
  hipStream_t st = as_stream(stream);
#define WSIS_DIAG_CASE(n)                                                                                   \
  case n:                                                                                                   \
    hipLaunchKernelGGL((spconv_fwd_kernel<n, true, true>), grid, dim3(256), 0, st, d_X, d_nbr, d_order, d_W, \
                       (const float*)nullptr, (const float*)nullptr, d_out, d_partial, M_out, M_out, K, Cin, Cout, \
                       k_per, 0, d_dbg);                                                                       \
    break;
  switch (nb) {
    WSIS_DIAG_CASE(1)
    WSIS_DIAG_CASE(2)
    WSIS_DIAG_CASE(3)
    WSIS_DIAG_CASE(4)
    WSIS_DIAG_CASE(5)
  }
#undef WSIS_DIAG_CASE
  WSIS_LAUNCH_CHECK();
  return WSIS_OK;
}

// diagnostic (not part of the ABI header): dW kernel with phase stamps, dbg[items * n_cib * 4 waves * 8]; returns the
// number of work items through *n_items.  Cout <= 64.
int wsis_debug_dw_diag(const float* d_X, const int32_t* d_nbr, const int32_t* d_order, const float* d_dY,
                       float* d_partial, int64_t M_in, int64_t M_out, int32_t K, int32_t Cin, int32_t Cout,
                       unsigned long long* d_dbg, int32_t* n_items, void* stream) {
  WSIS_REQUIRE(K <= DW_MAX_K && Cout <= 64 && n_items, "diag supports Cout <= 64");
  DwPlan plan;
  dw_make_plan(M_out, K, plan);
  *n_items = plan.begin[K];
  if (!d_dbg) return WSIS_OK;
  const int n_cib = (int)ceil_div(Cin, 32);
  const int nbo = (Cout + 31) / 32;
  const dim3 grid((unsigned)plan.begin[K], (unsigned)n_cib, 1);
  const size_t lds = (size_t)4 * nbo * 1024 * sizeof(float);
  hipStream_t st = as_stream(stream);
  if (nbo == 1)
    hipLaunchKernelGGL((spconv_dw_kernel<1, true>), grid, dim3(256), lds, st, d_X, d_nbr, d_order, d_dY, d_partial, M_in,
                       M_out, K, Cin, Cout, plan, n_cib, 0, d_dbg);
  else
    hipLaunchKernelGGL((spconv_dw_kernel<2, true>), grid, dim3(256), lds, st, d_X, d_nbr, d_order, d_dY, d_partial, M_in,
                       M_out, K, Cin, Cout, plan, n_cib, 0, d_dbg);
  WSIS_LAUNCH_CHECK();
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
  if (K > DW_MAX_K) return -1;
  DwPlan plan;
  dw_make_plan(M_out, K, plan);
  const int64_t ci_pad = ceil_div(Cin, 32) * 32;
  const int64_t old_bytes = (int64_t)plan.begin[K] * ci_pad * Cout * (int64_t)sizeof(float) + 256;
  if (dw2_supported(K, Cin, Cout)) return std::max(old_bytes, dw2_workspace_bytes(M_out, K, Cin, Cout));
  if (spconv_in_supported(K, Cin, Cout)) return std::max(old_bytes, spconv_in_dw_workspace_bytes(M_out));
  return old_bytes;
}

int32_t wsis_spconv_dw_bn_supported(int32_t K, int32_t Cin, int32_t Cout) {
  return (dw2_supported(K, Cout, Cin) && Cin % 32 == 0 && Cout % 32 == 0) ? 1 : 0;
}

int64_t wsis_spconv_dw_bn_workspace_bytes(int64_t M_in, int32_t K, int32_t Cin, int32_t Cout) {
  if (M_in < 0 || !wsis_spconv_dw_bn_supported(K, Cin, Cout)) return -1;
  return dw2_workspace_bytes(M_in, K, Cout, Cin);
}

int wsis_spconv_dw_bn(const float* d_X, const float* d_mean, const float* d_var, const float* d_gamma,
                      const float* d_beta, float eps, int32_t relu, const int32_t* d_nbr_b, const int32_t* d_order_b,
                      int32_t flip, const float* d_dY, float* d_dW, int64_t M_in, int64_t M_out, int32_t K, int32_t Cin,
                      int32_t Cout, void* d_ws, int64_t ws_bytes, void* stream) {
  WSIS_REQUIRE(M_in >= 0 && M_out >= 0 && d_dW, "bad args");
  WSIS_REQUIRE(wsis_spconv_dw_bn_supported(K, Cin, Cout), "needs K <= 32 and channel counts that are multiples of 32");
  hipStream_t st = as_stream(stream);
  if (M_in == 0 || M_out == 0) {
    WSIS_HIP_CHECK(hipMemsetAsync(d_dW, 0, sizeof(float) * (size_t)K * Cin * Cout, st));
    return WSIS_OK;
  }
  WSIS_REQUIRE(d_X && d_dY && d_ws && d_nbr_b && d_order_b, "null pointer");
  WSIS_REQUIRE((d_mean == nullptr) == (d_var == nullptr), "mean and var come in pairs");
  WSIS_REQUIRE(dw2_fits(M_out, M_in, K, Cout, Cin), "tensor too large for 32-bit gather offsets");
  WSIS_REQUIRE(ws_bytes >= wsis_spconv_dw_bn_workspace_bytes(M_in, K, Cin, Cout), "workspace too small");
  WSIS_REQUIRE(((reinterpret_cast<uintptr_t>(d_X) | reinterpret_cast<uintptr_t>(d_dY) | reinterpret_cast<uintptr_t>(d_dW) |
                 reinterpret_cast<uintptr_t>(d_ws)) & 15) == 0, "16-byte alignment");
  return dw2_launch_swapped(d_X, d_mean, d_var, d_gamma, d_beta, eps, (int)relu, d_nbr_b, d_order_b, (int)flip, d_dY, d_dW,
                            M_in, M_out, K, Cin, Cout, d_ws, st);
}

int wsis_hint_batch_rows(int64_t rows) {
  WSIS_REQUIRE(rows >= 0, "bad row count");
  dw2_set_batch_rows(rows);
  return WSIS_OK;
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
  WSIS_REQUIRE(K <= DW_MAX_K, "dW supports kernel volumes up to 128 offsets");
  WSIS_REQUIRE(M_in * Cin * 4 < ((int64_t)1 << 32) && M_out * Cout * 4 < ((int64_t)1 << 32),
               "dW addresses features as 32-bit byte offsets: a feature matrix must stay below 4 GiB");
  WSIS_REQUIRE(ws_bytes >= wsis_spconv_dw_workspace_bytes(M_out, K, Cin, Cout), "workspace too small");
  if (dw2_supported(K, Cin, Cout) && dw2_fits(M_in, M_out, K, Cin, Cout) &&
      ((d_nbr && d_order) || (!d_nbr && K == 1 && M_in == M_out)) && ((reinterpret_cast<uintptr_t>(d_X) | reinterpret_cast<uintptr_t>(d_dY) |
                                       reinterpret_cast<uintptr_t>(d_dW) | reinterpret_cast<uintptr_t>(d_ws)) & 15) == 0)
    return dw2_launch(d_X, d_nbr, d_order, d_dY, d_dW, M_in, M_out, K, Cin, Cout, d_ws, st);
  if (d_nbr && spconv_in_supported(K, Cin, Cout) &&
      ((reinterpret_cast<uintptr_t>(d_X) & 7) | (reinterpret_cast<uintptr_t>(d_dW) & 15) | (reinterpret_cast<uintptr_t>(d_ws) & 15)) == 0) {
    // the 6-channel input convolution: im2col on the matrix cores (csrc/spconv_in.hip)
    ProfScope prof_in(1, st, true);
    const int rc = spconv_in_dw_launch(d_X, d_nbr, d_order, d_dY, d_dW, M_out, d_ws, prof_in.ka(), prof_in.kb(), st);
    prof_in.stop();
    prof_in.tail();
    return rc;
  }
  DwPlan plan;
  dw_make_plan(M_out, K, plan);
  const int n_cib = (int)ceil_div(Cin, 32);
  const int ci_pad = n_cib * 32;
  float* partial = static_cast<float*>(d_ws);
  const int nblk = (Cout + 31) / 32;
  const dim3 grid((unsigned)plan.begin[K], (unsigned)n_cib, 1);
  // output-channel blocks are processed in groups of <= 5 per launch (register budget)
  ProfScope prof(1, st);
  for (int cb0 = 0; cb0 < nblk; cb0 += 5) {
    const int nbo = min(5, nblk - cb0);
    const int co0 = cb0 * 32;
    const size_t lds = (size_t)4 * nbo * 1024 * sizeof(float);
#define WSIS_DW_CASE(n)                                                                                  \
  case n:                                                                                                \
    WSIS_HIP_CHECK(hipFuncSetAttribute((const void*)spconv_dw_kernel<n, false>,                          \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));           \
    hipLaunchKernelGGL((spconv_dw_kernel<n, false>), grid, dim3(256), lds, st, d_X, d_nbr, d_order, d_dY, \
                       partial, M_in, M_out, K, Cin, Cout, plan, n_cib, co0, nullptr);                   \
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
  prof.stop();
  const int64_t total = (int64_t)K * Cin * Cout;
  if ((Cout & 3) == 0 && ((reinterpret_cast<uintptr_t>(partial) | reinterpret_cast<uintptr_t>(d_dW)) & 15) == 0) {
    hipLaunchKernelGGL(dw_reduce4_kernel, dim3(grid_for(total / 4, 256)), dim3(256), 0, st,
                       reinterpret_cast<const float4*>(partial), reinterpret_cast<float4*>(d_dW), plan, K, Cin, Cout / 4,
                       ci_pad);
  } else {
    hipLaunchKernelGGL(dw_reduce_kernel, dim3(grid_for(total, 256)), dim3(256), 0, st, partial, d_dW, plan, K, Cin,
                       Cout, ci_pad);
  }
  prof.tail();
  WSIS_LAUNCH_CHECK();
  return WSIS_OK;
}

}  // extern "C"
