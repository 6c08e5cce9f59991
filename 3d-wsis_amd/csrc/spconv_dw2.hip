// Weight gradient of the sparse convolutions, wave-autonomous form:
//   dW[k][ci][co] = sum over rows r with nbr[k][r] >= 0 of  X[nbr[k][r]][ci] * dY[r][co]
// for layers with Cin % 32 == 0, Cout % 4 == 0 and K <= 32 (every UNet layer but the 6-channel input conv, plus the
// point-level Linear layers as dense K = 1 products).
//
// One wave = one worker with a FIXED (offset group, 32-channel chunk of Cin, 32-channel block of Cout): it owns up to
// 8 kernel offsets (k = og, og + NOG, ...) whose 32x32 accumulators stay in registers for the whole launch (8 x 16
// AGPRs), and streams through its share of the 32-row slices of the output (tile order, strided so that every worker
// samples the whole scene).  Per slice: the dY rows of the slice (the B operand, 32 rows x 128 B) are gathered ONCE into
// LDS by LDS-DMA and kept as 16 fragment registers for all offsets of the slice; per active offset the X rows it pairs
// with (the A operand) are gathered the same way (full 128-byte lines, 8 rows per instruction -- the round-1 kernel
// fetched both operands with 4-byte loads, two rows per instruction) and read back transposed (lane = input channel)
// with conflict-free ds_read_b32; 16 v_mfma_f32_32x32x2_f32 per (slice, offset).  All gathers go through raw buffer
// descriptors (32-bit offsets, a missing pair is out of range and reads as zeros); the header of the next slice (table
// rows of the wave's offsets + row list, 5 dword-DMA instructions) arrives a slice ahead, two X tiles fly behind the one
// computing (counted s_waitcnt vmcnt), and the tile issue is interleaved by hand into the MFMA chain.  The 4 waves of a
// workgroup share a combination and add their accumulators through LDS in wave order; workgroup slabs are added in
// workgroup order by dw2_reduce_kernel: no atomics, bit-reproducible.  Dense 1x1 layers (no table) and a partial last
// output block (Cout % 4 == 0) are covered too.  Measurements and what bounds it: DESIGN.md 4.1.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <type_traits>

#include "common.h"

using namespace wsis;

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int GS = 8;                 // offsets per worker
constexpr int TILE = 32 * 128;        // bytes of a 32-row x 32-channel image
constexpr int HDR_INTS = GS * 32 + 64;                   // nb[8][32] + row[32] (+ 32 written by the upper half wave)
constexpr int HDR = HDR_INTS * 4;
constexpr int DA = 3;                                    // X-tile ring: two gathers in flight behind the one computing
constexpr int WAVE_LDS = 2 * HDR + TILE + DA * TILE + 256;     // headers (double), dY tile, X-tile ring, a row of -1
constexpr int WGW = 4;                                   // waves per workgroup (they share a slab)
constexpr int XSH_DEFAULT = 6;                           // slices per XCD chunk = 64 (2048 rows)

// LDS-DMA through a raw buffer descriptor: 32-bit byte offsets (one multiply per gathered row instead of 64-bit
// pointer arithmetic) and out-of-range offsets read as zero, so a missing pair (index -1, offset 2^32 - pitch) needs no
// select.  The step loop is VALU-issue bound without this: ~2000 cycles of address code against 1024 of MFMA.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* base, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), (short)0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void bdma16(rsrc_t r, uint32_t off, void* lds_dst) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_dst, 16, (int)off, 0, 0, 0);
}
__device__ __forceinline__ void bdma4(rsrc_t r, uint32_t off, void* lds_dst) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_dst, 4, (int)off, 0, 0, 0);
}

// every global read of the main loop is an LDS-DMA with a known instruction count (header 5, dY tile 4, X tile 4), so
// the wait for "the tile this step computes on" is a counted s_waitcnt: `after` = DMA instructions issued behind it
__device__ __forceinline__ void wait_after(int after) {
  switch (after) {
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;      // (8 + a two-instruction header)
    case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
    case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
    case 17: asm volatile("s_waitcnt vmcnt(17)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

// SWAP: the same product with the roles of the two tensors exchanged (the "own rows" form used when the convolution's
// input is a BatchNorm(+ReLU) applied on the fly): the rows of a slice are the convolution's INPUT rows (table =
// the dIn table, rows = inputs), `dY` is the convolution's raw input x [M_out = conv M_in rows, Cout = conv Cin channels]
// -- normalised ONCE per slice as its fragments are read (lane = channel: two per-lane constants, no pair mask: a
// missing pair gathers zeros on the other side) --, `X` is the gathered conv dY [M_in = conv M_out rows, Cin = conv
// Cout channels].  The accumulators then hold dW^T per dIn offset: the slab store transposes them and maps the offset
// back (flip: forward offset = K - 1 - k for submanifold tables).
struct DwBn {
  const float* mean;         // nullptr: the own rows are used as they are
  const float* var;
  const float* gamma;        // may be nullptr (1)
  const float* beta;         // may be nullptr (0)
  float eps;
  int relu;
  int flip;
};

template <bool DIAG, bool XCD, bool SWAP = false>
__global__ __launch_bounds__(WGW * 64, 8 / WGW) void spconv_dw2_kernel(const float* __restrict__ X, const int32_t* __restrict__ nbrS,
                                                            const int32_t* __restrict__ order,
                                                            const float* __restrict__ dY, float* __restrict__ partial, int64_t M_in,
                                                            int64_t M_out, int K, int Cin, int Cout, int NOG, int XSH,
                                                            unsigned long long* dbg, DwBn bn = DwBn{}) {
  unsigned long long t_start = 0, t_loop = 0, t_end_loop = 0;
  unsigned n_steps = 0, n_slices_done = 0;
  unsigned long long d_wait = 0, d_top = 0, d_chain = 0, d_bot = 0, t_prev = 0;
  if (DIAG) t_start = __builtin_readcyclecounter();
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r31 = lane & 31, half = lane >> 5;
  unsigned char* const my = lds + wave * WAVE_LDS;
  int32_t* const hdr0 = reinterpret_cast<int32_t*>(my);
  unsigned char* const Bt = my + 2 * HDR;
  unsigned char* const At = Bt + TILE;
  int32_t* const negrow = reinterpret_cast<int32_t*>(At + DA * TILE);
  negrow[lane] = -1;

  const int nchunk = Cin >> 5, nblk = (Cout + 31) >> 5;      // the last output block may be partial (Cout % 4 == 0)
  int combo = blockIdx.y;
  const int cb = combo % nblk;
  combo /= nblk;
  const int c = combo % nchunk;
  const int og = combo / nchunk;
  // SWAP with a BatchNorm on the own rows: this lane's channel is cb * 32 + r31 for every fragment it reads
  float bn_mu = 0.0f, bn_sc = 1.0f, bn_bt = 0.0f;
  const bool bn_on = SWAP && bn.mean != nullptr;
  if (bn_on) {
    const int ch = cb * 32 + (threadIdx.x & 31);
    if (ch < Cout) {
      bn_mu = bn.mean[ch];
      bn_sc = (bn.gamma ? bn.gamma[ch] : 1.0f) * rsqrtf(bn.var[ch] + bn.eps);
      bn_bt = bn.beta ? bn.beta[ch] : 0.0f;
    }
  }
  // Slice -> worker map, XCD-aware: workgroup id % 8 picks the XCD (gridDim.x is a multiple of 8, so blockIdx.x % 8
  // does); the slices are dealt to the XCDs in chunks of 2^XSH (a spatial brick of the tile order whose gathered X rows
  // are shared by its offsets and neighbouring slices, so they stay in that XCD's L2), and inside an XCD its waves take
  // the XCD's slices round-robin (every wave samples many bricks: the pair density varies along the tile order).
  // m-th slice of XCD x = ((m >> XSH) * 8 + x) << XSH | (m & (2^XSH - 1)).
  const int64_t n_slices = (M_out + 31) >> 5;
  const int xcd = XCD ? (int)(blockIdx.x & 7) : 0;
  const int64_t stride = XCD ? (int64_t)(gridDim.x >> 3) * WGW : (int64_t)gridDim.x * WGW;      // waves per XCD
  const int64_t first_m = XCD ? (int64_t)(blockIdx.x >> 3) * WGW + wave : (int64_t)blockIdx.x * WGW + wave;
  // number of enumerated positions: every position whose slice index is in range is a slice; positions past the
  // last full chunk round may map beyond n_slices and are skipped as empty
  const int64_t n_pos = XCD ? ((((n_slices + (1 << (XSH & 255)) - 1) >> (XSH & 255)) + 7) >> 3) << (XSH & 255) : n_slices;
  // The tile order comes heaviest slice first (rulebook.hip: slice scheduling): both deals run as a snake -- the waves
  // take the positions of every second FULL round of `stride` in reverse, the XCDs every second round of 8 chunks --
  // so that no wave (and no XCD) always gets the heavy end of a round.
  auto slice_of = [&](int64_t pos) -> int64_t {
    const int64_t n = pos / stride;
    int64_t m = pos;
    if ((n & 1) && (n + 1) * stride <= n_pos) m = n * stride + (stride - 1 - (pos - n * stride));
    if (!XCD) return m;
    const int sh = XSH & 255;
    const int64_t q = m >> sh;
    const int xq = (q & 1) ? 7 - xcd : xcd;
    return (((q << 3) + xq) << sh) | (m & ((1 << sh) - 1));
  };
  const int64_t first = first_m;

  f32x16 acc0, acc1, acc2, acc3, acc4, acc5, acc6, acc7;
#pragma unroll
  for (int i = 0; i < 16; ++i)
    acc0[i] = acc1[i] = acc2[i] = acc3[i] = acc4[i] = acc5[i] = acc6[i] = acc7[i] = 0.0f;

  const uint32_t x_pitch = (uint32_t)Cin * 4u, y_pitch = (uint32_t)Cout * 4u;
  const rsrc_t rsX = make_rsrc(reinterpret_cast<const char*>(X) + c * 128, (uint32_t)(M_in * x_pitch) - c * 128);
  const rsrc_t rsY = make_rsrc(reinterpret_cast<const char*>(dY) + cb * 128, (uint32_t)(M_out * y_pitch) - cb * 128);
  const rsrc_t rsN = make_rsrc(nbrS, (uint32_t)((int64_t)K * M_out * 4));
  const rsrc_t rsO = make_rsrc(order, (uint32_t)(M_out * 4));
  const int d_row = lane >> 3;
  const uint32_t d_po = (uint32_t)(lane & 7) * 16u;
  const uint32_t m_last = (uint32_t)(M_out - 1);
  const bool y_piece_ok = (uint32_t)cb * 128u + d_po < y_pitch;

  // header of a slice: table entries of this worker's offsets nb[slot][row] and the slice's output rows, 5 DMA
  // instructions of 4 bytes per lane (no alignment requirement on M_out); rows past M_out read row M_out - 1 and are
  // overwritten with -1 by fix_tail before the header is used, slots past K re-read slot 0 and are masked
  const bool dense = nbrS == nullptr;          // 1x1 layer without a table: row t pairs with itself (K = 1)
  const int NH = dense ? 0 : (XSH & 256) ? 2 : 5;      // DMA instructions of a header
  auto issueH = [&](int64_t s, int32_t* hb) {
    if (dense) {                               // written, not loaded: slot 0 and the row list are the identity
      const int64_t tt = s * 32 + r31;
      const int32_t v = tt < M_out ? (int32_t)tt : -1;
      hb[r31] = v;
      hb[GS * 32 + r31] = v;
      return;
    }
    uint32_t t = (uint32_t)s * 32u + (uint32_t)r31;
    t = t < m_last ? t : m_last;
    if (XSH & 256) {
      // (round 6) the 8 x 32 table block as ONE 16-byte-per-lane DMA: lane = slot * 8 + piece of four rows -- a table line
      // is contiguous in the row; four-byte alignment is enough for the hardware (M_out is arbitrary); rows past M_out
      // read whatever follows and are overwritten by fix_tail
      int k = og + (lane >> 3) * NOG;
      k = k < K ? k : og;
      bdma16(rsN, ((uint32_t)k * (uint32_t)M_out + (uint32_t)s * 32u + (uint32_t)(lane & 7) * 4u) * 4u, hb);
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        int k = og + (2 * q + half) * NOG;
        k = k < K ? k : og;
        bdma4(rsN, ((uint32_t)k * (uint32_t)M_out + t) * 4u, hb + q * 64);
      }
    }
    bdma4(rsO, t * 4u, hb + GS * 32);
  };
  // the last slice only: rows past M_out become missing pairs / missing rows
  auto fix_tail = [&](int32_t* hb, int64_t s) {
    if (s * 32 + 32 <= M_out) return;
    const int64_t t0 = s * 32 + (lane & 7) * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (t0 + e >= M_out) hb[lane * 4 + e] = -1;
    if (lane < 32 && s * 32 + lane >= M_out) hb[GS * 32 + lane] = -1;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };
  // mask of the offset slots with at least one pair in the slice (the header has landed and its tail is fixed)
  auto readmask = [&](const int32_t* hb) -> uint32_t {
    const int4 v = *reinterpret_cast<const int4*>(hb + lane * 4);      // slot lane >> 3, rows (lane & 7) * 4 ..
    const bool any = (og + (lane >> 3) * NOG < K) & ((v.x & v.y & v.z & v.w) >= 0);     // some index non-negative
    unsigned long long b = __ballot(any);
    b |= b >> 4;
    b |= b >> 2;
    b |= b >> 1;                       // bit 8 j = any lane of slot j
    uint32_t m = 0u;
#pragma unroll
    for (int j = 0; j < GS; ++j) m |= (uint32_t)((b >> (8 * j)) & 1ull) << j;
    return m;
  };
  // (the four row ids of a tile are read together, then the four pieces go out: left as one expression per piece hipcc
  // builds read -> wait -> multiply -> DMA four times over, behind a branch on the lane's piece mask)
  const uint32_t y_ok_mask = y_piece_ok ? 0xffffffffu : 0u;
  auto issueB = [&](const int32_t* hb, unsigned char* dst) {
    const int32_t* p = hb + GS * 32 + d_row;
    const int32_t r0 = p[0], r1 = p[8], r2 = p[16], r3 = p[24];
    __builtin_amdgcn_sched_barrier(0);
    // pieces past the row's last column (partial last block) read as zero too
    const uint32_t o0 = (((uint32_t)r0 * y_pitch + d_po) & y_ok_mask) | (0xffffff00u & ~y_ok_mask);
    const uint32_t o1 = (((uint32_t)r1 * y_pitch + d_po) & y_ok_mask) | (0xffffff00u & ~y_ok_mask);
    const uint32_t o2 = (((uint32_t)r2 * y_pitch + d_po) & y_ok_mask) | (0xffffff00u & ~y_ok_mask);
    const uint32_t o3 = (((uint32_t)r3 * y_pitch + d_po) & y_ok_mask) | (0xffffff00u & ~y_ok_mask);
    __builtin_amdgcn_sched_barrier(0);
    bdma16(rsY, o0, dst);
    bdma16(rsY, o1, dst + 1024);
    bdma16(rsY, o2, dst + 2048);
    bdma16(rsY, o3, dst + 3072);
  };
  auto issueA = [&](const int32_t* hb, int j, unsigned char* dst) {
    const int32_t* p = hb + j * 32 + d_row;
    const int32_t r0 = p[0], r1 = p[8], r2 = p[16], r3 = p[24];
    __builtin_amdgcn_sched_barrier(0);
    const uint32_t o0 = (uint32_t)r0 * x_pitch + d_po, o1 = (uint32_t)r1 * x_pitch + d_po;
    const uint32_t o2 = (uint32_t)r2 * x_pitch + d_po, o3 = (uint32_t)r3 * x_pitch + d_po;
    __builtin_amdgcn_sched_barrier(0);
    bdma16(rsX, o0, dst);
    bdma16(rsX, o1, dst + 1024);
    bdma16(rsX, o2, dst + 2048);
    bdma16(rsX, o3, dst + 3072);
  };
  // MFMA operands: lane (channel r31, half) takes rows 2s + half of a 32-row image, s = 0 .. 15
  auto readfrag = [&](const unsigned char* img, float (&f)[16]) {
    const float* p = reinterpret_cast<const float*>(img) + half * 32 + r31;
#pragma unroll
    for (int s = 0; s < 16; ++s) f[s] = p[s * 64];
  };
  auto mfma16 = [&](f32x16& acc, const float (&a)[16], const float (&b)[16]) {
#pragma unroll
    for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
  };

  if (first < n_pos) {
    // ---- first slice with work for this worker, loaded synchronously (positions that map past the last slice read
    // as empty: their rows are masked)
    int64_t g_pos = first;
    int64_t g_slice = slice_of(g_pos);
    uint32_t c_mask = 0u;
    for (;;) {
      issueH(g_slice, hdr0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      fix_tail(hdr0, g_slice);
      c_mask = readmask(hdr0);
      if (c_mask || g_pos + stride >= n_pos) break;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      g_pos += stride;
      g_slice = slice_of(g_pos);
    }
    if (c_mask) {
      int g_hb = 0;                              // header buffer of the generator's slice
      bool g_has_next = g_pos + stride < n_pos;
      if (g_has_next) issueH(slice_of(g_pos + stride), hdr0 + HDR_INTS);
      issueB(hdr0, Bt);
      issueA(hdr0, __builtin_ctz(c_mask), At);
      uint32_t g_rem = c_mask & (c_mask - 1u);   // offsets of the generator's slice not yet issued
      uint32_t mask_next = 0u;
      int gi = 0, ci = 0;                        // slices entered by the generator / by the compute side
      int F = 0, after0 = 0, after1 = 0;         // tiles in flight behind the current one; DMA counts behind tile t, t+1
      int c_rd = 0, a_wr = 1;                    // ring slots: compute reads, generator writes
      bool new_slice = true;
      float bfr[16], afr[16];
      if (DIAG) t_loop = __builtin_readcyclecounter();
      // generator: keeps two X tiles in flight behind the one computing; it enters the next slice only once the
      // compute side is in the slice before it (one dY tile, one spare header).  prepare() does everything of one
      // advance but the X-tile DMAs: returns the header row of the tile to issue and the DMA count of the advance.
      auto prepare = [&](const int32_t*& p, int& n) -> bool {
        int32_t* const hb = hdr0 + g_hb * HDR_INTS;
        if (g_rem) {
          p = hb + __builtin_ctz(g_rem) * 32 + d_row;
          g_rem &= g_rem - 1u;
          n = 4;
          return true;
        }
        if (gi != ci || !g_has_next) return false;
        // dY fragments and header words of the compute slice are in registers before their LDS is reused
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        int32_t* const hn = hdr0 + (g_hb ^ 1) * HDR_INTS;
        g_pos += stride;
        g_slice = slice_of(g_pos);
        fix_tail(hn, g_slice);
        uint32_t m = readmask(hn);
        while (m == 0u && g_pos + stride < n_pos) {            // rare: a slice without pairs for this worker
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          g_pos += stride;
          g_slice = slice_of(g_pos);
          issueH(g_slice, hn);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          after0 = after1 = 0;
          fix_tail(hn, g_slice);
          m = readmask(hn);
        }
        if (m == 0u) {
          g_has_next = false;
          return false;
        }
        g_hb ^= 1;
        ++gi;
        mask_next = m;
        n = 8;
        g_has_next = g_pos + stride < n_pos;
        if (g_has_next) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          issueH(slice_of(g_pos + stride), hb);                       // into the header the generator just left
          n = 8 + NH;
        }
        issueB(hn, Bt);
        p = hn + __builtin_ctz(m) * 32 + d_row;
        g_rem = m & (m - 1u);
        return true;
      };
      auto account = [&](int n, bool real) {
        after0 += n;
        if (F == 1) after1 += n;
        if (real) {
          ++F;
          a_wr = a_wr == DA - 1 ? 0 : a_wr + 1;
        }
      };
      // one (slice, offset slot) step; the accumulator is a compile-time choice at every call site so that each is
      // touched by exactly one MFMA chain.  The X-tile issue of the advance (4 header reads, 4 offset multiplies, 4
      // DMAs) is interleaved by hand into the 16-MFMA chain: left to itself hipcc runs it in front of the chain and the
      // wave spends as long in address code as in MFMAs.
      auto step = [&](f32x16& acc) {
        unsigned long long ta = 0, tb = 0, tc = 0, td = 0;
        if (DIAG) { ta = __builtin_readcyclecounter(); if (t_prev) d_bot += ta - t_prev; }
        wait_after(after0);
        if (DIAG) tb = __builtin_readcyclecounter();
        if (new_slice) {
          readfrag(Bt, bfr);
          if (bn_on) {       // the arithmetic of bn_apply_kernel: fma(x - mean, sc, beta), then ReLU
#pragma unroll
            for (int s = 0; s < 16; ++s) {
              float z = __builtin_fmaf(bfr[s] - bn_mu, bn_sc, bn_bt);
              bfr[s] = bn.relu ? fmaxf(z, 0.0f) : z;
            }
          }
          new_slice = false;
        }
        readfrag(At + c_rd * TILE, afr);
        if (F == 0) {        // after a blocked step: catch up outside the chain
          const int32_t* q;
          int nq;
          if (prepare(q, nq)) {
            unsigned char* dq = At + a_wr * TILE;
#pragma unroll
            for (int i = 0; i < 4; ++i) bdma16(rsX, (uint32_t)q[i * 8] * x_pitch + d_po, dq + i * 1024);
            account(nq, true);
          }
        }
        const int32_t* p = negrow + d_row;     // nothing to issue: four all-out-of-range DMAs (zero fill, no traffic)
        int n = 4;
        const bool real = prepare(p, n);
        unsigned char* const dst = At + a_wr * TILE;
#define WSIS_MFMA(s_) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(afr[s_], bfr[s_], acc, 0, 0, 0)
        if (DIAG) tc = __builtin_readcyclecounter();
        __builtin_amdgcn_sched_barrier(0);
        WSIS_MFMA(0); WSIS_MFMA(1); WSIS_MFMA(2); WSIS_MFMA(3);
        __builtin_amdgcn_sched_barrier(0);
        const int32_t e0 = p[0], e1 = p[8], e2 = p[16], e3 = p[24];
        __builtin_amdgcn_sched_barrier(0);
        WSIS_MFMA(4); WSIS_MFMA(5); WSIS_MFMA(6); WSIS_MFMA(7);
        __builtin_amdgcn_sched_barrier(0);
        const uint32_t o0 = (uint32_t)e0 * x_pitch + d_po, o1 = (uint32_t)e1 * x_pitch + d_po;
        const uint32_t o2 = (uint32_t)e2 * x_pitch + d_po, o3 = (uint32_t)e3 * x_pitch + d_po;
        __builtin_amdgcn_sched_barrier(0);
        WSIS_MFMA(8);
        bdma16(rsX, o0, dst);
        WSIS_MFMA(9);
        bdma16(rsX, o1, dst + 1024);
        WSIS_MFMA(10);
        bdma16(rsX, o2, dst + 2048);
        WSIS_MFMA(11);
        bdma16(rsX, o3, dst + 3072);
        __builtin_amdgcn_sched_barrier(0);
        WSIS_MFMA(12); WSIS_MFMA(13); WSIS_MFMA(14); WSIS_MFMA(15);
        __builtin_amdgcn_sched_barrier(0);
#undef WSIS_MFMA
        if (DIAG) { asm volatile("s_nop 0" :: "v"(acc[0])); td = __builtin_readcyclecounter(); d_wait += tb - ta; d_top += tc - tb; d_chain += td - tc; t_prev = td; }
        account(n, real);
        c_rd = c_rd == DA - 1 ? 0 : c_rd + 1;
        after0 = after1;
        after1 = 0;
        if (F > 0) --F;
        if (DIAG) ++n_steps;
      };
      for (;;) {
        if (c_mask & 1u) step(acc0);
        if (c_mask & 2u) step(acc1);
        if (c_mask & 4u) step(acc2);
        if (c_mask & 8u) step(acc3);
        if (c_mask & 16u) step(acc4);
        if (c_mask & 32u) step(acc5);
        if (c_mask & 64u) step(acc6);
        if (c_mask & 128u) step(acc7);
        if (DIAG) ++n_slices_done;
        if (gi == ci) break;            // the generator found no further slice
        ++ci;
        c_mask = mask_next;
        new_slice = true;
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
  if (DIAG) t_end_loop = __builtin_readcyclecounter();
  __syncthreads();

  // ---- the workgroup's accumulators, four offset slots at a time: LDS, added in wave order, one slab per workgroup.
  // C/D map: col (co) = lane & 31, row (ci) = (reg & 3) + 8 * (reg >> 2) + 4 * half
  float* const red = reinterpret_cast<float*>(lds);          // per wave [4 slots][32 ci][32 co] at stride WAVE_LDS
  float* const slab = partial + (int64_t)blockIdx.x * K * Cin * Cout;
  // SWAP reads the image transposed: rows padded to 33 floats (a stride of 32 would put the 32 lanes of a read on one bank)
  constexpr int RP = SWAP ? 33 : 32, QS = SWAP ? 1056 : 1024;
  auto put = [&](const f32x16& acc, int q) {
    float* mine = red + wave * (WAVE_LDS / 4) + q * QS;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int ci = (reg & 3) + 8 * (reg >> 2) + 4 * half;
      mine[ci * RP + r31] = acc[reg];
    }
  };
  auto flush4 = [&](const f32x16& a0, const f32x16& a1, const f32x16& a2, const f32x16& a3, int jbase) {
    if (og + jbase * NOG >= K) return;               // uniform over the workgroup
    put(a0, 0);
    put(a1, 1);
    put(a2, 2);
    put(a3, 3);
    __syncthreads();
    for (int e = threadIdx.x; e < 4096; e += WGW * 64) {
      const int k = og + (jbase + (e >> 10)) * NOG;
      if (k >= K) break;                             // e ascends through the slots
      if (SWAP) {
        // accumulator row = gathered channel (conv Cout side, chunk c), column = own channel (conv Cin side, block
        // cb): read transposed so that consecutive threads write consecutive conv-Cout elements of dW[kf][ci][co]
        const int a = (e >> 5) & 31, b = e & 31;              // a: own channel, b: gathered channel
        const int et = (e >> 10) * QS + b * RP + a;
        float v = red[et];
#pragma unroll
        for (int w = 1; w < WGW; ++w) v += red[w * (WAVE_LDS / 4) + et];
        const int kf = bn.flip ? K - 1 - k : k;
        if (cb * 32 + a < Cout) slab[((int64_t)kf * Cout + cb * 32 + a) * Cin + c * 32 + b] = v;
        continue;
      }
      float v = red[e];
#pragma unroll
      for (int w = 1; w < WGW; ++w) v += red[w * (WAVE_LDS / 4) + e];
      const int ci = (e >> 5) & 31, co = e & 31;
      if (cb * 32 + co < Cout) slab[((int64_t)k * Cin + c * 32 + ci) * Cout + cb * 32 + co] = v;
    }
    __syncthreads();
  };
  flush4(acc0, acc1, acc2, acc3, 0);
  flush4(acc4, acc5, acc6, acc7, 4);
  if (DIAG && lane == 0) {
    unsigned long long* d = dbg + ((int64_t)(blockIdx.y * gridDim.x + blockIdx.x) * WGW + wave) * 10;
    d[0] = t_start; d[1] = t_loop; d[2] = t_end_loop; d[3] = __builtin_readcyclecounter(); d[4] = n_steps; d[5] = n_slices_done;
    d[6] = d_wait; d[7] = d_top; d[8] = d_chain; d[9] = d_bot;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 6: the gathered operand straight to registers (spconv_dw3_kernel).
// In the product dW[k] = X_gathered^T . dY the gathered tile is the A operand with M = input channel: lane (ci, half) of
// v_mfma_f32_32x32x2_f32 holds X[row][ci] for one row per instruction -- the 32 lanes of a half read 32 CONSECUTIVE floats
// of one gathered row.  A plain `buffer_load_dword` per instruction is therefore fully coalesced (two 128-byte lines, one
// per half) and lands in the fragment layout: no LDS-DMA (four instructions of 100-185 cycles each to issue),
// no X-tile ring in LDS, no 16 transposed `ds_read_b32` per step.  (The forward / dIn products cannot do this: their
// gathered tile is the operand with M = row, a row per lane = four cache lines per quad, measured in round 5.)
// Per step: the 16 row ids of the lane's half come from the slice header in LDS by four broadcast `ds_read_b128`
// (instruction s pairs rows s and 16 + s instead of 2s and 2s + 1 -- any pairing is a valid order of the k-sum as long as
// the dY fragments use the same), one v_mad_u32_u24 per row turns them into byte offsets (a missing pair, -1, lands
// beyond the tensor and reads zeros: dw3_fits checks that for the pitch), 16 loads fill the OTHER of two static register
// sets while the 16 MFMAs of the current step run on this one -- the loads ride the gaps of the dependent MFMA chain.
// Everything else is dw2's: a wave owns up to 8 offsets of one (Cin chunk, Cout block), streams its share of the slices,
// keeps the dY tile of a slice as 16 fragment registers (LDS-DMA once per slice), header of the next slice a slice ahead,
// same slabs, same fixed-order slab sum.  The order of additions inside a 32-row tile differs from dw2 (row pairing), so
// the two kernels agree to rounding, each bit-reproducible.
constexpr int HDR3_INTS = GS * 32 + 64;
constexpr int HDR3 = HDR3_INTS * 4;
constexpr int WAVE_LDS3 = 2 * HDR3 + TILE;      // 6.4 KB per wave: two headers + the dY tile; the epilogue adds ONE accumulator
                                               // per wave at a time through the same bytes (4 KB per wave).  25.6 KB per
                                               // workgroup against dw2's 77 KB: the dIn items that share the CU from the main
                                               // stream (37 KB each at levels 1-2, 8.4 KB at level 0) keep their occupancy

template <bool DIAG>
__global__ __launch_bounds__(WGW * 64, 2) void spconv_dw3_kernel(const float* __restrict__ X, const int32_t* __restrict__ nbrS,
                                                                 const int32_t* __restrict__ order,
                                                                 const float* __restrict__ dY, float* __restrict__ partial,
                                                                 int64_t M_in, int64_t M_out, int K, int Cin, int Cout, int NOG,
                                                                 int flags, unsigned long long* dbg) {
  unsigned long long t_start = 0, t_loop = 0, t_end_loop = 0;
  unsigned n_steps = 0, n_slices_done = 0;
  unsigned long long d_wait = 0, d_top = 0, d_chain = 0, d_bot = 0, t_prev = 0;
  if (DIAG) t_start = __builtin_readcyclecounter();
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int h16 = flags & 1;                 // header table block by one 16-byte DMA
  const bool balanced = (flags & 2) != 0;    // offset groups of a 3 x 3 x 3 product balanced by activity (below)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r31 = lane & 31, half = lane >> 5;
  unsigned char* const my = lds + wave * WAVE_LDS3;
  int32_t* const hdr0 = reinterpret_cast<int32_t*>(my);
  unsigned char* const Bt = my + 2 * HDR3;

  const int nchunk = Cin >> 5, nblk = (Cout + 31) >> 5;
  int combo = blockIdx.y;
  const int cb = combo % nblk;
  combo /= nblk;
  const int c = combo % nchunk;
  const int og = combo / nchunk;
  // offset of slot j of this workgroup's offset group.  3 x 3 x 3: a partition balanced by ACTIVITY -- every slice has its
  // centre offset, ~70 % of the slices a given face offset, ~40 % an edge offset, ~5 % (level 0) a corner offset, and an
  // axis-aligned surface has exactly centre + 4 faces + 4 edges.  Grouped by k mod 4 (rounds 2-6) the group of the centre
  // held six edge offsets as well: 1.29 x the mean number of steps at level 0 (1.24 / 1.17 at levels 1 / 2) and the launch
  // lasts as long as its busiest workgroups.  This partition (centre + 1 face + 2 edges + 3 corners | 6 edges + 1 corner |
  // 2 faces + 3 edges + 1 corner | 3 faces + 1 edge + 3 corners; no group takes more than 3 of the 9 offsets of an axis
  // plane) is within 1.03 x of the mean on the measured and on the class-symmetrised activity of levels 0-3
  // (tools/dw_balance.py, profiles/r06_dw_balance.txt).  Which workgroup owns an offset does not enter the arithmetic: the
  // slices a wave walks and the slab order are those of before, the result is the same bit for bit.  Other kernel sizes
  // (and WSIS_DW_BAL=0 in the EXPERIMENTAL build): slot j = offset og + j * NOG.
  const unsigned long long ktab =
      !(K == 27 && balanced) ? 0ull
      : og == 0 ? 0xff1a13110d0c0602ull      //  2  6 12 13 17 19 26
      : og == 1 ? 0xff17150b07050100ull      //  0  1  5  7 11 21 23
      : og == 2 ? 0xffff19140f0a0403ull      //  3  4 10 15 20 25
                : 0xff181612100e0908ull;     //  8  9 14 16 18 22 24
  auto kof = [&](int j) -> int { return ktab ? (int)((ktab >> (8 * j)) & 0xffull) : og + j * NOG; };
  const int k_first = kof(0);
  const int64_t n_slices = (M_out + 31) >> 5;
  const int64_t stride = (int64_t)gridDim.x * WGW;
  const int64_t first = (int64_t)blockIdx.x * WGW + wave;
  const int64_t n_pos = n_slices;
  // dw2's snake over full rounds of `stride`, by ROUND: this worker's position in round r is first + r * stride, so the
  // round of a position never has to be divided out (a 64-bit division per call -- two per slice advance -- was ~500 of
  // the ~3,000 cycles an advance costs: profiles/r06_dw3_advance_probe.txt)
  auto slice_at = [&](int64_t r) -> int64_t {
    if ((r & 1) && (r + 1) * stride <= n_pos) return r * stride + (stride - 1 - first);
    return first + r * stride;
  };

  f32x16 acc0, acc1, acc2, acc3, acc4, acc5, acc6, acc7;
#pragma unroll
  for (int i = 0; i < 16; ++i)
    acc0[i] = acc1[i] = acc2[i] = acc3[i] = acc4[i] = acc5[i] = acc6[i] = acc7[i] = 0.0f;

  const uint32_t x_pitch = (uint32_t)Cin * 4u, y_pitch = (uint32_t)Cout * 4u;
  const rsrc_t rsX = make_rsrc(reinterpret_cast<const char*>(X) + c * 128, (uint32_t)(M_in * x_pitch) - c * 128);
  const rsrc_t rsY = make_rsrc(reinterpret_cast<const char*>(dY) + cb * 128, (uint32_t)(M_out * y_pitch) - cb * 128);
  const rsrc_t rsN = make_rsrc(nbrS, (uint32_t)((int64_t)K * M_out * 4));
  const rsrc_t rsO = make_rsrc(order, (uint32_t)(M_out * 4));
  const int d_row = lane >> 3;
  const uint32_t d_po = (uint32_t)(lane & 7) * 16u;
  const uint32_t m_last = (uint32_t)(M_out - 1);
  const bool y_piece_ok = (uint32_t)cb * 128u + d_po < y_pitch;
  const uint32_t a_col = (uint32_t)r31 * 4u;

  const bool dense = nbrS == nullptr;
  // header of a slice: nb[slot][row] of this worker's offsets + the slice's own rows.  h16: the 8 x 32 table block is ONE
  // 16-byte-per-lane DMA (lane = slot * 8 + piece of four rows; a table line is contiguous in the row) instead of four
  // 4-byte ones; rows past M_out read whatever follows and are overwritten by fix_tail, slots past K are masked
  auto issueH = [&](int64_t s, int32_t* hb) {
    if (dense) {
      const int64_t tt = s * 32 + r31;
      const int32_t v = tt < M_out ? (int32_t)tt : -1;
      hb[r31] = v;
      hb[GS * 32 + r31] = v;
      return;
    }
    uint32_t t = (uint32_t)s * 32u + (uint32_t)r31;
    t = t < m_last ? t : m_last;
    if (h16) {
      int k = kof(lane >> 3);
      k = k < K ? k : k_first;
      bdma16(rsN, ((uint32_t)k * (uint32_t)M_out + (uint32_t)s * 32u + (uint32_t)(lane & 7) * 4u) * 4u, hb);
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        int k = kof(2 * q + half);
        k = k < K ? k : k_first;
        bdma4(rsN, ((uint32_t)k * (uint32_t)M_out + t) * 4u, hb + q * 64);
      }
    }
    bdma4(rsO, t * 4u, hb + GS * 32);
  };
  auto fix_tail = [&](int32_t* hb, int64_t s) {
    if (s * 32 + 32 <= M_out) return;
    const int64_t t0 = s * 32 + (lane & 7) * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (t0 + e >= M_out) hb[lane * 4 + e] = -1;
    if (lane < 32 && s * 32 + lane >= M_out) hb[GS * 32 + lane] = -1;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };
  auto readmask = [&](const int32_t* hb) -> uint32_t {
    const int4 v = *reinterpret_cast<const int4*>(hb + lane * 4);
    const bool any = (kof(lane >> 3) < K) & ((v.x & v.y & v.z & v.w) >= 0);
    unsigned long long b = __ballot(any);
    b |= b >> 4;
    b |= b >> 2;
    b |= b >> 1;
    // bit 8j of b = slot j has a pair; the eight bits gathered into one byte by a multiplication (bit 8j x bit 7(8 - j)
    // lands on bit 56 + j, all 64 partial products on distinct bits: no carries)
    static_assert(GS == 8, "the gather below is for eight slots");
    return (uint32_t)(((b & 0x0101010101010101ull) * 0x0102040810204080ull) >> 56);
  };
  const uint32_t y_ok_mask = y_piece_ok ? 0xffffffffu : 0u;
  auto issueB = [&](const int32_t* hb, unsigned char* dst) {
    const int32_t* p = hb + GS * 32 + d_row;
    const int32_t r0 = p[0], r1 = p[8], r2 = p[16], r3 = p[24];
    __builtin_amdgcn_sched_barrier(0);
    const uint32_t o0 = (((uint32_t)r0 * y_pitch + d_po) & y_ok_mask) | (0xffffff00u & ~y_ok_mask);
    const uint32_t o1 = (((uint32_t)r1 * y_pitch + d_po) & y_ok_mask) | (0xffffff00u & ~y_ok_mask);
    const uint32_t o2 = (((uint32_t)r2 * y_pitch + d_po) & y_ok_mask) | (0xffffff00u & ~y_ok_mask);
    const uint32_t o3 = (((uint32_t)r3 * y_pitch + d_po) & y_ok_mask) | (0xffffff00u & ~y_ok_mask);
    __builtin_amdgcn_sched_barrier(0);
    bdma16(rsY, o0, dst);
    bdma16(rsY, o1, dst + 1024);
    bdma16(rsY, o2, dst + 2048);
    bdma16(rsY, o3, dst + 3072);
  };
  // the dY fragments of a slice: instruction s takes rows s (lower half wave) and 16 + s (upper)
  auto readfragB = [&](float (&f)[16]) {
    const float* p = reinterpret_cast<const float*>(Bt) + half * 16 * 32 + r31;
#pragma unroll
    for (int s = 0; s < 16; ++s) f[s] = p[s * 32];
  };
  // row ids of the 16 gathered rows of this lane's half for offset slot j of header hb (nullptr: nothing to gather -- -1:
  // sixteen loads beyond the tensor: zeros, no traffic)
  auto rowids = [&](const int32_t* hb, int j, int4 (&q)[4]) {
    if (!hb) {
      q[0] = q[1] = q[2] = q[3] = make_int4(-1, -1, -1, -1);
      return;
    }
    const int4* p = reinterpret_cast<const int4*>(hb + j * 32 + half * 16);
    q[0] = p[0];
    q[1] = p[1];
    q[2] = p[2];
    q[3] = p[3];
  };
  auto idof = [&](const int4 (&q)[4], int s) -> int32_t {
    const int4& v = q[s >> 2];
    return (s & 3) == 0 ? v.x : (s & 3) == 1 ? v.y : (s & 3) == 2 ? v.z : v.w;
  };
  auto loadA = [&](int32_t id) -> float {      // (v_mad_u32_u24: one VALU instruction per gathered row)
    const uint32_t off = __umul24((uint32_t)id, x_pitch) + a_col;
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsX, (int)off, 0, 0));
  };

  // ONE register set for the gathered operand: a[s] is read by MFMA s of this step and, right behind it, becomes the
  // destination of the load for MFMA s of the NEXT step -- a prefetch distance of exactly one step (16 MFMAs, ~1,040
  // cycles) with 16 registers.  The waits in front of the MFMAs are the compiler's (plain loads: it counts them).
  float a[16], bfr[16];
#pragma unroll
  for (int s = 0; s < 16; ++s) a[s] = bfr[s] = 0.0f;

  if (first < n_pos) {
    int64_t g_pos = first, g_round = 0;
    int64_t g_slice = slice_at(g_round);
    uint32_t c_mask = 0u;
    for (;;) {      // first slice with work for this worker, loaded synchronously
      issueH(g_slice, hdr0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      fix_tail(hdr0, g_slice);
      c_mask = readmask(hdr0);
      if (c_mask || g_pos + stride >= n_pos) break;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      g_pos += stride;
      g_slice = slice_at(++g_round);
    }
    if (c_mask) {
      int hb = 0;                                        // header buffer of the slice whose tiles are being issued
      bool has_next = g_pos + stride < n_pos;            // a header for the position after it is in flight / has landed
      if (has_next) issueH(slice_at(g_round + 1), hdr0 + HDR3_INTS);
      issueB(hdr0, Bt);
      uint32_t g_rem = c_mask & (c_mask - 1u);           // slots of that slice not yet handed out
      uint32_t mask_next = 0u;
      bool more = false;                                 // another slice follows the one being computed
      {
        int4 q[4];
        rowids(hdr0, __builtin_ctz(c_mask), q);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int s = 0; s < 16; ++s) a[s] = loadA(idof(q, s));
      }
      bool new_slice = true;
      if (DIAG) t_loop = __builtin_readcyclecounter();

      // one (slice, offset slot) step: find the next step's tile (a slice advance where the current slice is used up:
      // mask of the next header, its dY tile, the header after it), then 16 MFMAs with the 16 loads of the next step
      // between them
      auto step = [&](f32x16& acc) {
        unsigned long long ta = 0, tb = 0, tc = 0, td = 0;
        if (DIAG) { ta = __builtin_readcyclecounter(); if (t_prev) d_bot += ta - t_prev; }
        if (new_slice) {
          // everything but the 16 loads of this step's tile has landed: the dY tile (issued at the advance, a step ago)
          asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
          readfragB(bfr);
          new_slice = false;
          if (DIAG) ++n_slices_done;
        }
        if (DIAG) tb = __builtin_readcyclecounter();
        const int32_t* nh = nullptr;
        int nj = 0;
        if (g_rem) {
          nh = hdr0 + hb * HDR3_INTS;
          nj = __builtin_ctz(g_rem);
          g_rem &= g_rem - 1u;
        } else if (has_next) {
          // the next header was issued a slice ago: older than the 16 loads in flight
          asm volatile("s_waitcnt vmcnt(16)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");      // (dY fragments in registers before Bt is reused)
          int32_t* const hn = hdr0 + (hb ^ 1) * HDR3_INTS;
          g_pos += stride;
          g_slice = slice_at(++g_round);
          fix_tail(hn, g_slice);
          uint32_t m = readmask(hn);
          while (m == 0u && g_pos + stride < n_pos) {             // rare: a slice without pairs for this worker
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            g_pos += stride;
            g_slice = slice_at(++g_round);
            issueH(g_slice, hn);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            fix_tail(hn, g_slice);
            m = readmask(hn);
          }
          if (m) {
            has_next = g_pos + stride < n_pos;
            if (has_next) {
              asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
              issueH(slice_at(g_round + 1), hdr0 + hb * HDR3_INTS);      // into the header the finished slice leaves
            }
            issueB(hn, Bt);
            hb ^= 1;
            nh = hn;
            nj = __builtin_ctz(m);
            g_rem = m & (m - 1u);
            mask_next = m;
            more = true;
          } else {
            has_next = false;
          }
        }
        int4 q[4];
        rowids(nh, nj, q);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (DIAG) tc = __builtin_readcyclecounter();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 16; ++s) {
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], bfr[s], acc, 0, 0, 0);
          a[s] = loadA(idof(q, s));
          __builtin_amdgcn_sched_barrier(0);
        }
        if (DIAG) { td = __builtin_readcyclecounter(); d_wait += tb - ta; d_top += tc - tb; d_chain += td - tc; t_prev = td; ++n_steps; }
      };
      for (;;) {
        more = false;
        if (c_mask & 1u) step(acc0);
        if (c_mask & 2u) step(acc1);
        if (c_mask & 4u) step(acc2);
        if (c_mask & 8u) step(acc3);
        if (c_mask & 16u) step(acc4);
        if (c_mask & 32u) step(acc5);
        if (c_mask & 64u) step(acc6);
        if (c_mask & 128u) step(acc7);
        if (!more) break;
        c_mask = mask_next;
        new_slice = true;
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
  if (DIAG) t_end_loop = __builtin_readcyclecounter();
  __syncthreads();

  // ---- epilogue: one offset slot at a time through LDS (4 KB per wave), added in wave order, one slab per workgroup
  float* const red = reinterpret_cast<float*>(lds);
  float* const slab = partial + (int64_t)blockIdx.x * K * Cin * Cout;
  auto flush1 = [&](const f32x16& acc, int j) {
    const int k = kof(j);
    if (k >= K) return;                              // uniform over the workgroup
    float* mine = red + wave * 1024;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int ci = (reg & 3) + 8 * (reg >> 2) + 4 * half;
      mine[ci * 32 + r31] = acc[reg];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 1024; e += WGW * 64) {
      float v = red[e];
#pragma unroll
      for (int w = 1; w < WGW; ++w) v += red[w * 1024 + e];
      const int ci = e >> 5, co = e & 31;
      if (cb * 32 + co < Cout) slab[((int64_t)k * Cin + c * 32 + ci) * Cout + cb * 32 + co] = v;
    }
    __syncthreads();
  };
  flush1(acc0, 0);
  flush1(acc1, 1);
  flush1(acc2, 2);
  flush1(acc3, 3);
  flush1(acc4, 4);
  flush1(acc5, 5);
  flush1(acc6, 6);
  flush1(acc7, 7);
  if (DIAG && lane == 0) {
    unsigned long long* d = dbg + ((int64_t)(blockIdx.y * gridDim.x + blockIdx.x) * WGW + wave) * 10;
    d[0] = t_start; d[1] = t_loop; d[2] = t_end_loop; d[3] = __builtin_readcyclecounter(); d[4] = n_steps; d[5] = n_slices_done;
    d[6] = d_wait; d[7] = d_top; d[8] = d_chain; d[9] = d_bot;
  }
}

// dW = sum over workgroup slabs in a fixed order.  A workgroup of 256 threads covers QW = 256 / SL element quads; thread
// (quad el, slab lane sl of SL) adds slabs sl, sl + SL, ... in ascending order -- every load of a thread in flight at once
// (P / SL <= 8 for the plans of dw2_P_plan) -- and the SL lane sums are then added in lane order through LDS.  SL follows
// P alone (dw2_reduce_lanes), so the order of additions of an element is a function of the slab count: bit-reproducible,
// the same whether a product is finished by its own launch (dw2_reduce_kernel) or by the batched launch of a whole
// backward pass (dw2_reduce_batch_kernel).  Round 5's form (8 slab lanes, 32 quads per workgroup, a walk of P / 8 dependent
// trips of 4) left a level-0 launch of four scenes at 216 workgroups of 16 sequential loads: 30 us for 14 MB.
template <int SL>
__device__ __forceinline__ void dw2_reduce_body(const float4* __restrict__ partial, float4* __restrict__ dW, int64_t total4,
                                                int P, int64_t block, float4* red) {
  constexpr int QW = 256 / SL;
  const int el = threadIdx.x % QW, sl = threadIdx.x / QW;
  const int64_t t = block * QW + el;
  float4 s = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  if (t < total4) {
    const float4* src = partial + t;
    int p = sl;
    for (; p + 3 * SL < P; p += 4 * SL) {            // four slabs of this lane per trip, all loads issued before the adds
      const float4 v0 = src[(int64_t)p * total4], v1 = src[(int64_t)(p + SL) * total4];
      const float4 v2 = src[(int64_t)(p + 2 * SL) * total4], v3 = src[(int64_t)(p + 3 * SL) * total4];
      s.x += v0.x; s.y += v0.y; s.z += v0.z; s.w += v0.w;
      s.x += v1.x; s.y += v1.y; s.z += v1.z; s.w += v1.w;
      s.x += v2.x; s.y += v2.y; s.z += v2.z; s.w += v2.w;
      s.x += v3.x; s.y += v3.y; s.z += v3.z; s.w += v3.w;
    }
    for (; p < P; p += SL) {
      const float4 v = src[(int64_t)p * total4];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  if (SL > 1) {
    red[sl * (QW + 1) + el] = s;                      // [SL][QW + 1] of the workgroup's DW2_RED_F4 float4s
    __syncthreads();
    if (sl == 0 && t < total4) {
#pragma unroll
      for (int l = 1; l < SL; ++l) {
        const float4 v = red[l * (QW + 1) + el];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
      dW[t] = s;
    }
    __syncthreads();                                  // (the batched form walks several blocks per workgroup)
  } else if (t < total4) {
    dW[t] = s;
  }
}
__host__ __device__ inline int dw2_reduce_lanes(int P) { return P >= 64 ? 32 : P >= 16 ? 16 : P >= 8 ? 8 : P >= 2 ? 2 : 1; }
__host__ __device__ inline int64_t dw2_reduce_blocks(int64_t total4, int P) {
  const int qw = 256 / dw2_reduce_lanes(P);
  return (total4 + qw - 1) / qw;
}
constexpr int DW2_RED_F4 = 32 * 9;      // SL (QW + 1) at its largest (SL = 32)
__device__ __forceinline__ void dw2_reduce_any(const float4* partial, float4* dW, int64_t total4, int P, int64_t block,
                                               float4* red) {
  switch (dw2_reduce_lanes(P)) {
    case 32: dw2_reduce_body<32>(partial, dW, total4, P, block, red); break;
    case 16: dw2_reduce_body<16>(partial, dW, total4, P, block, red); break;
    case 8: dw2_reduce_body<8>(partial, dW, total4, P, block, red); break;
    case 2: dw2_reduce_body<2>(partial, dW, total4, P, block, red); break;
    default: dw2_reduce_body<1>(partial, dW, total4, P, block, red); break;
  }
}
__global__ __launch_bounds__(256) void dw2_reduce_kernel(const float4* __restrict__ partial, float4* __restrict__ dW,
                                                         int64_t total4, int P) {
  __shared__ float4 red[DW2_RED_F4];
  dw2_reduce_any(partial, dW, total4, P, blockIdx.x, red);
}

// every deferred slab sum of (a part of) a backward pass in ONE launch: workgroup b finishes block b - first[i] of
// product i -- the body, and with it the order of additions, of dw2_reduce_kernel
constexpr int DW2_BATCH = 64;
struct DwRedBatch {
  const float4* partial[DW2_BATCH];
  float4* dW[DW2_BATCH];
  int64_t total4[DW2_BATCH];
  int32_t P[DW2_BATCH];
  int32_t first[DW2_BATCH + 1];
  int32_t n;
};
__global__ __launch_bounds__(256) void dw2_reduce_batch_kernel(const DwRedBatch b) {
  __shared__ float4 red[DW2_RED_F4];
  int i = 0;
  while (i + 1 < b.n && (int)blockIdx.x >= b.first[i + 1]) ++i;
  dw2_reduce_any(b.partial[i], b.dW[i], b.total4[i], b.P[i], (int64_t)((int)blockIdx.x - b.first[i]), red);
}

#if WSIS_EXPERIMENTAL
// round 5's slab sum (8 slab lanes, 32 quads per workgroup), kept for the in-process A/B (WSIS_DW2_RED=0)
__global__ __launch_bounds__(256) void dw2_reduce_r5_kernel(const float4* __restrict__ partial, float4* __restrict__ dW,
                                                         int64_t total4, int P) {
  __shared__ float4 red[8][32];
  const int el = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int64_t t = (int64_t)blockIdx.x * 32 + el;
  float4 s = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  if (t < total4) {
#pragma unroll 4
    for (int p = sl; p < P; p += 8) {
      const float4 v = partial[(int64_t)p * total4 + t];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  red[sl][el] = s;
  __syncthreads();
  if (sl == 0 && t < total4) {
#pragma unroll
    for (int l = 1; l < 8; ++l) {
      const float4 v = red[l][el];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    dW[t] = s;
  }
}

#endif

int dw2_env(const char* name, int dflt) {      // (launch-plan knobs: live in the EXPERIMENTAL build)
  const char* e = tune_env(name);
  return e ? atoi(e) : dflt;
}

// the fixed-order slab sum that finishes a product (EXPERIMENTAL build: WSIS_DW2_RED=0 selects round 5's kernel)
void dw2_launch_reduce(const float* partial, float* d_dW, int64_t total4, int P, hipStream_t st) {
#if WSIS_EXPERIMENTAL
  if (dw2_env("WSIS_DW2_RED", 1) == 0) {
    hipLaunchKernelGGL(dw2_reduce_r5_kernel, dim3((unsigned)((total4 + 31) / 32)), dim3(256), 0, st,
                       reinterpret_cast<const float4*>(partial), reinterpret_cast<float4*>(d_dW), total4, P);
    return;
  }
#endif
  hipLaunchKernelGGL(dw2_reduce_kernel, dim3((unsigned)dw2_reduce_blocks(total4, P)), dim3(256), 0, st,
                     reinterpret_cast<const float4*>(partial), reinterpret_cast<float4*>(d_dW), total4, P);
}

// workgroups per combination (= slabs): ~target waves over the launch, never more than there are slices
// Launch-plan hint (wsis_hint_batch_rows): rows of the finest level of the batch being trained.  One scene per step the
// main stream's chain of dIn products is the step's critical path and the weight gradients beside it should stay small
// (1,024 waves: one 4-wave workgroup per CU; 2,048 cost the step +0.6 ... +1.5 %); from two scenes per step on the GPU is
// busy throughout and what counts is how fast the gradients get done (2,048 waves at every level: -0.4 / -0.6 / -1.9 %
// at 2 / 3 / 4 scenes per step; tools/ab_step.py).  Never a matter of correctness: the workspace covers both plans.
std::atomic<int64_t> g_batch_rows{0};
constexpr int64_t WIDE_ROWS = 250000;

int dw2_P_plan(int64_t M_out, int K, int Cin, int Cout, bool wide) {
  // (read per call) up to 8,192 slices one 4-wave workgroup per CU, above that -- or with the batch hint -- two
  const int64_t n_slices = (M_out + 31) / 32;
  const int64_t lo = dw2_env("WSIS_DW2_LO", 0), hi = dw2_env("WSIS_DW2_HI", 8192);
  const int target = dw2_env("WSIS_DW2_WAVES", (wide || n_slices >= hi || n_slices < lo) ? 2048 : 1024);
  const int NOG = (K + GS - 1) / GS;
  const int64_t combos = (int64_t)NOG * (Cin / 32) * ((Cout + 31) / 32);
  int64_t P = (target / WGW + combos - 1) / combos;
  // where rounding up overshoots the workgroup target (36 combinations at 96 channels: 8 x 36 = 288 workgroups for 256
  // CUs -- 32 CUs take two, and the launch lasts as long as those), round down instead (7 x 36 = 252: one workgroup per
  // CU): level 2 46.3 -> 42.8 us, level 4 23.5 -> 21.0 alone; with the balanced offset groups the one-scene step
  // 7.458 -> 7.438 ms in-process (profiles/r06_dw_bal2.txt).  WSIS_DW2_PFLOOR=0 (EXPERIMENTAL build): round up
  if (dw2_env("WSIS_DW2_PFLOOR", 1) && K != 1 && P * combos > target / WGW && P > 1) --P;
  // dense products with many output blocks (K = 1: the shared filter weights of the GNN, 32 x 2,080 over 7 x S rows):
  // the wave target leaves a combination 4 workgroups, i.e. ~30 slices per wave, each a slice advance -- and the launch runs
  // in the GNN phase of the step, alone on the GPU.  Up to 32 workgroups per combination while a wave keeps >= 4 slices
  // (in-process: the step 7.471 -> 7.451 ms at one scene, 20.148 -> 20.044 at four; profiles/r06_ab_densewg.txt).
  if (K == 1) P = std::max(P, std::min<int64_t>(n_slices / (4 * WGW), dw2_env("WSIS_DW2_DENSE_WG", 32)));
  const int64_t cap = (n_slices + WGW - 1) / WGW;
  if (P > cap) P = cap;
  if (P >= 8) P &= ~(int64_t)7;     // multiple of 8: blockIdx.x % 8 is the XCD for every blockIdx.y
  if (P < 1) P = 1;
  return (int)P;
}
int dw2_P(int64_t M_out, int K, int Cin, int Cout) {
  return dw2_P_plan(M_out, K, Cin, Cout, g_batch_rows.load(std::memory_order_relaxed) >= WIDE_ROWS);
}

}  // namespace

namespace wsis {

bool dw2_supported(int K, int Cin, int Cout) {
  static int on = -1;
  if (on < 0) on = dw2_env("WSIS_DW2", 1);
  return on && K >= 1 && K <= 32 && Cin >= 32 && Cin % 32 == 0 && Cout >= 4 && Cout % 4 == 0;     // (16-byte dY pieces)
}

bool dw2_fits(int64_t M_in, int64_t M_out, int K, int Cin, int Cout) {
  const int64_t lim = (int64_t)1 << 31;
  return M_in * Cin * 4 < lim && M_out * Cout * 4 < lim && (int64_t)K * M_out * 4 < lim && M_in >= 1 && M_out >= 1;
}

int64_t dw2_workspace_bytes(int64_t M_out, int K, int Cin, int Cout) {      // (either plan: the hint may change between sizing and launch)
  const int P = std::max(dw2_P_plan(M_out, K, Cin, Cout, false), dw2_P_plan(M_out, K, Cin, Cout, true));
  return (int64_t)P * K * Cin * Cout * (int64_t)sizeof(float) + 256;
}

void dw2_set_batch_rows(int64_t rows) { g_batch_rows.store(rows, std::memory_order_relaxed); }
int64_t dw2_batch_rows() { return g_batch_rows.load(std::memory_order_relaxed); }

static thread_local DwRedRec* t_defer = nullptr;
void dw2_set_defer(DwRedRec* slot) { t_defer = slot; }

int dw2_reduce_batch(const DwRedRec* recs, int n, hipStream_t st) {
  int i = 0;
  while (i < n) {
    DwRedBatch b;
    b.n = 0;
    int64_t blocks = 0;
    for (; i < n && b.n < DW2_BATCH; ++i) {
      const DwRedRec& r = recs[i];
      if (!r.partial) continue;
      const int64_t nb = dw2_reduce_blocks(r.total4, r.P);
      if (blocks + nb > ((int64_t)1 << 30)) break;
      b.partial[b.n] = reinterpret_cast<const float4*>(r.partial);
      b.dW[b.n] = reinterpret_cast<float4*>(r.dW);
      b.total4[b.n] = r.total4;
      b.P[b.n] = r.P;
      b.first[b.n] = (int32_t)blocks;
      blocks += nb;
      ++b.n;
    }
    if (b.n == 0) continue;
    b.first[b.n] = (int32_t)blocks;
    for (int j = b.n; j < DW2_BATCH; ++j) {      // (defined values in the unused tail of the argument block)
      b.partial[j] = nullptr;
      b.dW[j] = nullptr;
      b.total4[j] = 0;
      b.P[j] = 0;
      b.first[j + 1] = (int32_t)blocks;
    }
    hipLaunchKernelGGL(dw2_reduce_batch_kernel, dim3((unsigned)blocks), dim3(256), 0, st, b);
    WSIS_LAUNCH_CHECK();
  }
  return WSIS_OK;
}

// the own-rows form (kernel SWAP): conv input x [M_in, Cin] normalised on the fly, gathered conv dY [M_out, Cout] through
// the dIn table (rows = conv inputs); writes dW [K, Cin, Cout] exactly like dw2_launch
int dw2_launch_swapped(const float* d_X, const float* d_mean, const float* d_var, const float* d_gamma, const float* d_beta,
                       float eps, int relu, const int32_t* d_nbr_b, const int32_t* d_order_b, int flip, const float* d_dY,
                       float* d_dW, int64_t M_in, int64_t M_out, int K, int Cin, int Cout, void* d_ws, hipStream_t st) {
  // kernel roles: gathered = conv dY (rows M_out, channels Cout), own = conv x (rows M_in, channels Cin)
  const int NOG = (K + GS - 1) / GS;
  const int P = dw2_P(M_in, K, Cout, Cin);
  float* partial = static_cast<float*>(d_ws);
  const dim3 grid((unsigned)P, (unsigned)(NOG * (Cout / 32) * ((Cin + 31) / 32)), 1);
  const size_t ldsb = (size_t)WAVE_LDS * WGW;
  static bool attr_set = false;
  if (!attr_set) {
    WSIS_HIP_CHECK(hipFuncSetAttribute((const void*)spconv_dw2_kernel<false, false, true>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
    attr_set = true;
  }
  DwBn bn;
  bn.mean = d_mean;
  bn.var = d_var;
  bn.gamma = d_gamma;
  bn.beta = d_beta;
  bn.eps = eps;
  bn.relu = relu;
  bn.flip = flip;
  ProfScope prof(1, st);
  hipLaunchKernelGGL((spconv_dw2_kernel<false, false, true>), grid, dim3(WGW * 64), ldsb, st, d_dY, d_nbr_b, d_order_b, d_X,
                     partial, M_out, M_in, K, Cout, Cin, NOG, XSH_DEFAULT, (unsigned long long*)nullptr, bn);
  prof.stop();
  WSIS_LAUNCH_CHECK();
  const int64_t total4 = (int64_t)K * Cin * Cout / 4;
  if (t_defer) {      // the executor finishes this product with the other slab sums of its pass (dw2_reduce_batch)
    t_defer->partial = partial;
    t_defer->dW = d_dW;
    t_defer->total4 = total4;
    t_defer->P = P;
    return WSIS_OK;
  }
  dw2_launch_reduce(partial, d_dW, total4, P, st);
  prof.tail();
  WSIS_LAUNCH_CHECK();
  return WSIS_OK;
}

// the register-gather kernel applies when row ids fit 24 bits (v_mad_u32_u24) and a missing pair's offset
// (0xFFFFFF x pitch mod 2^32) lies beyond the gathered tensor, so that it reads zeros
bool dw3_fits(int64_t M_in, int Cin) {
  const uint64_t p = (uint64_t)Cin * 4u;
  const uint32_t w = (uint32_t)(0xFFFFFFull * p);
  return M_in < (1 << 24) - 1 && (uint64_t)w >= (uint64_t)M_in * p && w <= 0xFFFFFF00u;
}
// WSIS_DW3 (read per call): 1 (default) = spconv_dw3_kernel where it fits, 0 = spconv_dw2_kernel everywhere.  Alone on the
// GPU the two kernels are at parity (profiles/r06_dw3.txt); IN THE STEP the register-gather kernel wins because of what it
// does not occupy: 26 KB of LDS per workgroup instead of 77 KB -- the dIn items that share its CU from the main stream
// (37 KB each at levels 1-2, 8.4 KB at level 0) keep their occupancy: 7.77 -> 7.66 ms at one scene, 11.75 -> 11.67 at two,
// 20.37 -> 20.06 at four (profiles/r06_ab_dw3_lds.txt; with a 66 KB epilogue the same kernel measured 20.40 vs 20.35)
int dw3_mode() {
  const char* e = getenv("WSIS_DW3");
  return e ? atoi(e) : 1;
}
// header table block by ONE 16-byte DMA (both kernels): WSIS_DW_H16 = 1 always (default), 2 only where every table line is
// 16-byte aligned, 0 never (EXPERIMENTAL build: live)
bool dw3_h16(int64_t M_out) {
  const int m = dw2_env("WSIS_DW_H16", 1);
  return m == 1 || (m == 2 && (M_out & 3) == 0);
}
int dw3_flags(int64_t M_out) { return (dw3_h16(M_out) ? 1 : 0) | (dw2_env("WSIS_DW_BAL", 1) ? 2 : 0); }

int dw2_launch(const float* d_X, const int32_t* d_nbr, const int32_t* d_order, const float* d_dY, float* d_dW,
               int64_t M_in, int64_t M_out, int K, int Cin, int Cout, void* d_ws, hipStream_t st) {
  const int NOG = (K + GS - 1) / GS;
  const int P = dw2_P(M_out, K, Cin, Cout);
  float* partial = static_cast<float*>(d_ws);
  const dim3 grid((unsigned)P, (unsigned)(NOG * (Cin / 32) * ((Cout + 31) / 32)), 1);
  if (dw3_mode() && dw3_fits(M_in, Cin)) {
    const size_t ldsb3 = (size_t)WAVE_LDS3 * WGW;
    static bool attr3_set = false;
    if (!attr3_set) {
      WSIS_HIP_CHECK(hipFuncSetAttribute((const void*)spconv_dw3_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)ldsb3));
      attr3_set = true;
    }
    ProfScope prof(1, st);
    hipLaunchKernelGGL((spconv_dw3_kernel<false>), grid, dim3(WGW * 64), ldsb3, st, d_X, d_nbr, d_order, d_dY, partial, M_in,
                       M_out, K, Cin, Cout, NOG, dw3_flags(M_out), (unsigned long long*)nullptr);
    prof.stop();
    WSIS_LAUNCH_CHECK();
    const int64_t total4 = (int64_t)K * Cin * Cout / 4;
    if (t_defer) {
      t_defer->partial = partial;
      t_defer->dW = d_dW;
      t_defer->total4 = total4;
      t_defer->P = P;
      return WSIS_OK;
    }
    dw2_launch_reduce(partial, d_dW, total4, P, st);
    prof.tail();
    WSIS_LAUNCH_CHECK();
    return WSIS_OK;
  }
  const size_t ldsb = (size_t)WAVE_LDS * WGW;
  static bool attr_set = false;
  if (!attr_set) {
    WSIS_HIP_CHECK(hipFuncSetAttribute((const void*)spconv_dw2_kernel<false, true>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
    WSIS_HIP_CHECK(hipFuncSetAttribute((const void*)spconv_dw2_kernel<false, false>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
    attr_set = true;
  }
  // (read per call: in-process A/Bs flip them) XCD-aware dealing of the slices: rows >= WSIS_DW2_XCD (0 = never)
  const int64_t xcd_rows = dw2_env("WSIS_DW2_XCD", 0);     // measured at one scene: no gain at level 0, slower below (chunk imbalance)
  const bool xcd_on = xcd_rows > 0 && M_out >= xcd_rows;
  // (bit 8 of the kernel's XSH argument: the header's table block by one 16-byte DMA instead of four 4-byte ones)
  const int xsh = dw2_env("WSIS_DW2_XSH", XSH_DEFAULT) | (dw3_h16(M_out) ? 256 : 0);
  ProfScope prof(1, st);
  if (xcd_on && P % 8 == 0)
    hipLaunchKernelGGL((spconv_dw2_kernel<false, true>), grid, dim3(WGW * 64), ldsb, st, d_X, d_nbr, d_order,
                       d_dY, partial, M_in, M_out, K, Cin, Cout, NOG, xsh, (unsigned long long*)nullptr);
  else
    hipLaunchKernelGGL((spconv_dw2_kernel<false, false>), grid, dim3(WGW * 64), ldsb, st, d_X, d_nbr, d_order, d_dY,
                       partial, M_in, M_out, K, Cin, Cout, NOG, xsh, (unsigned long long*)nullptr);
  prof.stop();
  WSIS_LAUNCH_CHECK();
  const int64_t total4 = (int64_t)K * Cin * Cout / 4;
  if (t_defer) {      // the executor finishes this product with the other slab sums of its pass (dw2_reduce_batch)
    t_defer->partial = partial;
    t_defer->dW = d_dW;
    t_defer->total4 = total4;
    t_defer->P = P;
    return WSIS_OK;
  }
  dw2_launch_reduce(partial, d_dW, total4, P, st);
  prof.tail();
  WSIS_LAUNCH_CHECK();
  return WSIS_OK;
}

}  // namespace wsis

// diagnostic build of the kernel: per-wave cycle stamps (start, loop begin, loop end, end) and step / slice counts,
// 6 x u64 per wave in launch order; *n_waves receives the wave count.  tools/dw2_stamps.py reads it.
extern "C" int wsis_debug_dw2_diag(const void* d_X, const void* d_nbr, const void* d_order, const void* d_dY,
                                   int64_t M_in, int64_t M_out, int K, int Cin, int Cout, void* d_ws, void* d_dbg, int64_t dbg_bytes,
                                   int64_t* n_waves, void* stream) {
  WSIS_REQUIRE(wsis::dw2_supported(K, Cin, Cout) && wsis::dw2_fits(M_in, M_out, K, Cin, Cout) &&
                   ((d_nbr && d_order) || (!d_nbr && K == 1)),
               "shape not supported by the dw2 kernel");
  const int NOG = (K + GS - 1) / GS;
  const int P = dw2_P(M_out, K, Cin, Cout);
  const dim3 grid((unsigned)P, (unsigned)(NOG * (Cin / 32) * ((Cout + 31) / 32)), 1);
  *n_waves = (int64_t)grid.x * grid.y * WGW;
  WSIS_REQUIRE(dbg_bytes >= *n_waves * 80, "stamp buffer too small");
  if (wsis::dw3_mode() && wsis::dw3_fits(M_in, Cin)) {      // the register-gather kernel's stamps (same record layout)
    const size_t ldsb3 = (size_t)WAVE_LDS3 * WGW;
    WSIS_HIP_CHECK(hipFuncSetAttribute((const void*)spconv_dw3_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)ldsb3));
    hipLaunchKernelGGL((spconv_dw3_kernel<true>), grid, dim3(WGW * 64), ldsb3, wsis::as_stream(stream),
                       static_cast<const float*>(d_X), static_cast<const int32_t*>(d_nbr),
                       static_cast<const int32_t*>(d_order), static_cast<const float*>(d_dY), static_cast<float*>(d_ws),
                       M_in, M_out, K, Cin, Cout, NOG, wsis::dw3_flags(M_out), static_cast<unsigned long long*>(d_dbg));
    WSIS_LAUNCH_CHECK();
    return WSIS_OK;
  }
  WSIS_REQUIRE(P % 8 == 0, "diagnostic build is the XCD-aware variant");
  const size_t ldsb = (size_t)WAVE_LDS * WGW;
  WSIS_HIP_CHECK(hipFuncSetAttribute((const void*)spconv_dw2_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)ldsb));
  hipLaunchKernelGGL((spconv_dw2_kernel<true, true>), grid, dim3(WGW * 64), ldsb, wsis::as_stream(stream),
                     static_cast<const float*>(d_X), static_cast<const int32_t*>(d_nbr),
                     static_cast<const int32_t*>(d_order), static_cast<const float*>(d_dY), static_cast<float*>(d_ws),
                     M_in, M_out, K, Cin, Cout, NOG, dw2_env("WSIS_DW2_XSH", XSH_DEFAULT) | (wsis::dw3_h16(M_out) ? 256 : 0),
                     static_cast<unsigned long long*>(d_dbg));
  WSIS_LAUNCH_CHECK();
  return WSIS_OK;
}
