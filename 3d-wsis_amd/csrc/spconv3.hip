// Sparse convolution, role-split ring form (round 5; SURVEY 8a a7-a11, the forward and dIn products of
// sparse_unet3d.py:127-143,254-298 at the levels with many work items).
//
//   out[r,:] = sum_k X[nbr[k][r],:] @ W[k]      weights as B^T: WT[k][cout][cin]   (the product of spconv2.hip)
//
// What spconv_fwd2_kernel loses (tools/conv2_stamps.py, round 5): a step (16 MFMAs, 1,024 cycles) waits for a gather
// that was issued ONE step earlier while a gather takes 1,900 (idle) - 3,500 (loaded) cycles to land, so a wave makes a
// step per ~3,700 cycles whatever shares its SIMD; every work item pays a 2.5-4.7 us prologue (table, then first rows) and
// a 1-5 us epilogue with all waves of a round in the same phase; the second round of a launch runs under-occupied.
//
// Here a workgroup is resident (one per CU) and its 12 waves have ROLES:
//   * 4 CONSUMER waves (one per SIMD): ds_read_b128 fragments from a private ring of R_D gathered tiles, weights straight
//     to registers three steps ahead (counted vmcnt), 16 v_mfma_f32_32x32x2_f32 per step.  No other memory traffic, no
//     barrier: at the end of a work item the accumulator goes to an LDS partial buffer and the next item starts at once.
//   * 4 LOADER waves, one per consumer: per work item the gather-table lines of the consumer's kernel offsets and the
//     tile-order line arrive by LDS-DMA (dword pieces), one item ahead; the active-offset mask is their by-product and is
//     published to the consumer; every step's 32 x 128 B tile is gathered by 4 LDS-DMA pieces into the consumer's ring
//     (in-order vmcnt, a 1-bit tag FIFO in a scalar register pair) and published through a monotonic counter in LDS once
//     landed.  A loader runs as far ahead as ring and table buffers allow, across work items: the prologue latencies of
//     an item overlap the MFMAs of the items before.  (A wave issues an instruction per 4-5 cycles: ONE loader for two
//     consumers, as first built, took ~4,700 cycles per step.)
//   * 4 FINISHER waves: add the partial accumulators of the NT consumers of a team in consumer order, bias / residual,
//     store, BatchNorm slice partials (forward statistics or the backward reduction) -- the epilogue of work item n runs
//     beside the steps of item n + 1.
// A team of NT consumers (1, 2 or 4) shares a work item by offset index (k mod NT), as the waves of spconv_fwd2_kernel
// do: the value of an output element is bit-identical to spconv_fwd2_kernel<1, NT, ...> (k-ordered chain per consumer,
// consumers added in order), whatever workgroup computes it.  Exact fp32, no atomics.  Every spin is bounded: a wait
// that gives up sets the abort word of the workgroup (everybody leaves) and the launch's error word.
#include "spconv2_body.h"

namespace {

// every LDS access of this kernel goes through address-space-3 pointers: through generic pointers hipcc emits flat_
// loads (counted in vmcnt AND lgkmcnt) and drains the LDS-DMA pieces in flight in front of each of them
#define LDS_AS __attribute__((address_space(3)))
typedef LDS_AS unsigned char* lptr;

constexpr int R_NC = 4;                   // consumer waves
constexpr int R_NL = 4;                   // loader waves (one per consumer)
constexpr int R_NF = 4;                   // finisher waves
constexpr int R_WAVES = R_NC + R_NL + R_NF;
constexpr int R_D = 6;                    // ring slots (4 KB tiles) per consumer
constexpr int R_LB = 3;                   // weight fragments in flight ahead of the MFMAs (register sets: R_LB + 1)
constexpr int R_TL = 28;                  // table lines (128 B) per (consumer, parity): <= 27 offsets + the tile-order line
constexpr int R_KMAX = 27;
constexpr int R_HR = 6;                   // published work-item headers (mask, row ids) per consumer: the weight generator of
                                          // a consumer runs up to 4 work items ahead of its MFMAs (one-step items), and a
                                          // header slot is free again only when the finisher is done with the item
constexpr int R_LAQ = 15;                 // quads (4 LDS-DMA pieces) in flight per loader: 60 <= the 6-bit vmcnt
constexpr unsigned R_SPIN = 1u << 22;     // polls before a wait gives up

struct RingCtl {
  volatile unsigned prod[R_NC];           // steps landed in the ring (loader)
  volatile unsigned cons[R_NC];           // steps whose fragments are in registers (consumer)
  volatile unsigned hcnt[R_NC];           // work-item headers published (loader)
  volatile unsigned pdone[R_NC];          // partial accumulators written (consumer)
  volatile unsigned finp[R_NC][R_NF];     // finp[team][x]: finished work items i of the team with i % NS == x (one finisher
                                          // per (team, x): NS = finishers per team)
  volatile unsigned iss[R_NC];            // work items fully issued (loader): the table buffer of item iss - 1 is free
  volatile unsigned abort_;
  unsigned pad_[3];
  volatile unsigned hmask[R_NC][R_HR];    // active offsets of the consumer in its i-th work item
  volatile int32_t rowid[R_NC][R_HR][32]; // output rows of that work item's slice (-1: past the end)
};
constexpr int R_CTL_BYTES = 3328;
static_assert(sizeof(RingCtl) <= R_CTL_BYTES, "control block");
constexpr int R_TAB_BYTES = R_TL * 128;                          // per (consumer, parity)
constexpr int R_OFF_TAB = R_CTL_BYTES;
constexpr int R_OFF_PART = R_OFF_TAB + R_NC * 2 * R_TAB_BYTES;   // [consumer][parity][4 KB]
constexpr int R_OFF_RING = R_OFF_PART + R_NC * 2 * 4096;         // [consumer][R_D][4 KB]
constexpr int R_LDS_BYTES = R_OFF_RING + R_NC * R_D * 4096;
static_assert(R_LDS_BYTES <= 160 * 1024, "LDS budget");

struct RingArgs {
  const float* X;
  const int32_t* nbrS;        // packed gather table [K][M_out] (may be null: dense 1x1, rows gather themselves)
  const int32_t* order;       // tile order (may be null: identity)
  const float* WT;
  const float* bias;
  const float* residual;
  float* out;
  float* stats;
  unsigned* err;              // error word of the launch (may be null)
  unsigned long long* dbg;    // DIAG build: [workgroup][wave][8] cycle counters
  BnEpi epi;
  int64_t M_out;
  int K, Cin, Cout, flip;
  uint32_t x_bytes;
  int total, gx;              // work items, slices
};

// LDS accesses of the LOADER waves: in front of every LDS read it can see, hipcc drains the LDS-DMA pieces the wave has
// in flight (s_waitcnt vmcnt(0): the DMA destinations might alias) -- the loader then runs one gather at a time (first
// builds: <= 3 VMEM operations outstanding, 3,400 cycles per loop).  Through inline asm the compiler sees no LDS access.
__device__ __forceinline__ uint32_t lds_addr(lptr p) { return (uint32_t)(uintptr_t)p; }
__device__ __forceinline__ uint32_t lds_ld(lptr p) {      // a wave-uniform word (control counters)
  uint32_t v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(lds_addr(p)) : "memory");
  return __builtin_amdgcn_readfirstlane(v);
}
__device__ __forceinline__ uint32_t lds_ld_lane(lptr p) {      // a word per lane
  uint32_t v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(lds_addr(p)) : "memory");
  return v;
}
template <int STRIDE>      // four words at p, p + STRIDE, ...: one LDS latency
__device__ __forceinline__ void lds_ld4(lptr p, uint32_t (&v)[4]) {
  asm volatile("ds_read_b32 %0, %4\n\tds_read_b32 %1, %4 offset:%5\n\tds_read_b32 %2, %4 offset:%6\n\t"
               "ds_read_b32 %3, %4 offset:%7\n\ts_waitcnt lgkmcnt(0)"
               : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3])
               : "v"(lds_addr(p)), "n"(STRIDE), "n"(2 * STRIDE), "n"(3 * STRIDE)
               : "memory");
}
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4 lds_ld128_lane(lptr p) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(lds_addr(p)) : "memory");
  return v;
}
__device__ __forceinline__ void lds_st128(lptr p, u32x4 v) {
  asm volatile("ds_write_b128 %0, %1" ::"v"(lds_addr(p)), "v"(v) : "memory");
}
__device__ __forceinline__ void lds_st(lptr p, uint32_t v) {
  asm volatile("ds_write_b32 %0, %1" ::"v"(lds_addr(p)), "v"(v) : "memory");
}
__device__ __forceinline__ void lds_drain() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
#define CTL_PTR(field) ((lptr)ctl + offsetof(RingCtl, field))

// a bounded wait; on give-up: abort word (every wait of the workgroup returns at once from then on) + error word
template <typename F>
__device__ __forceinline__ void spin_until(LDS_AS RingCtl* ctl, unsigned* err, unsigned code, F ready) {
  unsigned n = 0;
  while (!ready()) {
    __builtin_amdgcn_s_sleep(1);
    if (ctl->abort_) return;
    if (++n > R_SPIN) {
      ctl->abort_ = 1u;
      if (err) atomicOr(err, code);
      return;
    }
  }
}

// DIAG: cycles spent inside a wait, accumulated per wave (tools/ring_stamps.py)
struct Stamp {
  unsigned long long v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};
template <bool DIAG>
__device__ __forceinline__ unsigned long long now() {
  return DIAG ? __builtin_amdgcn_s_memtime() : 0ull;
}
template <bool DIAG>
__device__ __forceinline__ void stamp_out(const RingArgs& a, const Stamp& st, int wave, int lane) {
  if (DIAG && a.dbg && lane == 0) {
    unsigned long long* d = a.dbg + ((int64_t)blockIdx.x * R_WAVES + wave) * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) d[j] = st.v[j];
  }
}

__device__ __forceinline__ uint32_t own_pattern(int NT, int r, int K) {
  const uint32_t all = K >= 32 ? 0xffffffffu : ((1u << K) - 1u);
  const uint32_t pat = NT == 1 ? 0xffffffffu : NT == 2 ? 0x55555555u : 0x11111111u;
  return (pat << r) & all;
}

// work items of a team: e = wg + (q + i NTEAM) G, i = 0, 1, ...; (bx, by) advance without a division per item
struct ItemPos {
  int e, bx, by, sx, sy, gx;
  __device__ __forceinline__ void init(int e0, int stride, int gx_) {
    gx = gx_;
    e = e0;
    by = e0 / gx_;
    bx = e0 - by * gx_;
    sy = stride / gx_;
    sx = stride - sy * gx_;
  }
  __device__ __forceinline__ void next(int stride) {
    e += stride;
    bx += sx;
    by += sy;
    if (bx >= gx) {
      bx -= gx;
      ++by;
    }
  }
};

// ------------------------------------------------------------------------------------------------------------ loader
// One loader per consumer: the steps of the consumer's work items, nothing else -- the table fetch and the header of an
// item are the helper's (below).  ~60 scalar instructions per step at ~8 cycles each is what a step costs this wave; with
// the table / header work on the same wave it could not stay ahead of a consumer that takes ~1,400 cycles per step.
template <int NT, bool DIAG>
__device__ __forceinline__ void ring_loader(const RingArgs& a, lptr lds, LDS_AS RingCtl* ctl, const int c,
                                            const int lane, const int wg, const int G) {
  constexpr int NTEAM = R_NC / NT;
  const int q = c / NT;
  const int stride = NTEAM * G;
  if (wg + q * G >= a.total) return;
  const int d_row = lane >> 3, d_piece = lane & 7;
  const int nchunk = a.Cin >> 5;
  const rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.X), (short)0, (int)a.x_bytes, 0x00020000);
  uint32_t a_po[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) a_po[j] = (uint32_t)((d_piece ^ swz(j * 8 + d_row)) << 4);
  const uint32_t pat = own_pattern(NT, c % NT, a.K);
  const lptr tab = lds + R_OFF_TAB + c * 2 * R_TAB_BYTES;
  const lptr ring = lds + R_OFF_RING + c * R_D * 4096;
  Stamp st;
  const unsigned long long t_begin = now<DIAG>();
  int nfifo = 0;                        // quads (steps) in flight: every one is a step, published when it has landed
  unsigned issued = 0, published = 0, cons_c = 0, hcnt_c = 0;
  int slot = 0;
  unsigned idle = 0;

  // The loader never blocks in s_waitcnt while something else could be done: a wave that waits for its oldest gather to
  // land cannot issue the next ones.  It reads its own outstanding-VMEM count from HW_REG_IB_STS (vm_cnt: bits 3:0 and
  // 23:22) and retires what has landed.  `fresh`: a piece was issued since the last look at the counter -- the value read
  // by s_getreg may not yet include the pieces issued just before it (s_waitcnt interlocks with the issue path, s_getreg
  // need not): then one quad of margin is kept, and the exact comparison is made on the next pass
  bool fresh = false;
  auto retire_landed = [&]() -> bool {
    if (nfifo == 0) return false;
    const uint32_t v = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 7);
    const int out = (int)((v & 15u) | (((v >> 22) & 3u) << 4)) + (fresh ? 4 : 0);
    fresh = false;
    if (DIAG && (unsigned long long)out > st.v[6]) st.v[6] = (unsigned long long)out;
    const int left = (out + 3) >> 2;                 // quads that may still be in flight
    if (left >= nfifo) return false;
    published += (unsigned)(nfifo - left);
    nfifo = left;
    lds_st(CTL_PTR(prod) + c * 4, published);
    return true;
  };
  auto give_up = [&]() -> bool {       // a pass without progress; true: leave
    __builtin_amdgcn_s_sleep(1);
    if (lds_ld(CTL_PTR(abort_))) return true;
    if (++idle > R_SPIN) {
      lds_st(CTL_PTR(abort_), 1u);
      if (a.err) atomicOr(a.err, 0x100u << c);
      return true;
    }
    return false;
  };

  ItemPos pos;
  pos.init(wg + q * G, stride, a.gx);
  bool dead = false;
  for (int i = 0; pos.e < a.total && !dead; ++i, pos.next(stride)) {
    // header of the item (helper): its mask; the table lines of buffer i & 1 hold byte offsets
    while ((unsigned)i >= hcnt_c) {
      hcnt_c = lds_ld(CTL_PTR(hcnt) + c * 4);
      if ((unsigned)i < hcnt_c) break;
      if (retire_landed()) {
        idle = 0;
      } else if (give_up()) {
        dead = true;
        break;
      }
    }
    if (dead) break;
    idle = 0;
    uint32_t rem = lds_ld(CTL_PTR(hmask) + (c * R_HR + i % R_HR) * 4);
    const lptr tb = tab + (i & 1) * R_TAB_BYTES + d_row * 4;
    do {
      uint32_t voff[4];
      int nch = nchunk;
      if (rem) {
        const int k = __builtin_ctz(rem);
        rem &= rem - 1u;
        const int o = a.nbrS ? __builtin_popcount(pat & ((1u << k) - 1u)) : 0;
        lds_ld4<32>(tb + o * 128, voff);
#pragma unroll
        for (int j = 0; j < 4; ++j) voff[j] += a_po[j];
      } else {
        nch = 1;                       // an item without an active offset is ONE step of zeros
#pragma unroll
        for (int j = 0; j < 4; ++j) voff[j] = NO_ROW;
      }
      for (int ch = 0; ch < nch && !dead; ++ch) {
        if (DIAG) ++st.v[1];
        (void)retire_landed();
        while (issued - cons_c >= (unsigned)R_D || nfifo >= R_LAQ) {       // ring full (or the vmcnt field)
          cons_c = lds_ld(CTL_PTR(cons) + c * 4);
          if (issued - cons_c < (unsigned)R_D && nfifo < R_LAQ) break;
          if (DIAG) ++st.v[7];
          if (retire_landed()) {
            idle = 0;
          } else if (give_up()) {
            dead = true;
            break;
          }
        }
        if (dead) break;
        idle = 0;
        const lptr dst = ring + slot * 4096;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (LDS_AS void*)(dst + j * 1024), 16, (int)voff[j], ch * 128, 0, 0);
        fresh = true;
        ++nfifo;
        ++issued;
        slot = slot + 1 == R_D ? 0 : slot + 1;
      }
    } while (rem && !dead);
    lds_st(CTL_PTR(iss) + c * 4, (unsigned)i + 1u);      // the table buffer of the item is free
  }
  while (nfifo > 0 && !dead) {
    if (retire_landed()) {
      idle = 0;
    } else if (give_up()) {
      break;
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  if (DIAG) {
    st.v[0] = now<DIAG>() - t_begin;
    st.v[5] = issued;
    stamp_out<DIAG>(a, st, R_NC + c, lane);
  }
}

// ---------------------------------------------------------------------------------------------------------- consumer
template <int NT, bool DIAG>
__device__ __forceinline__ void ring_consumer(const RingArgs& a, lptr lds, LDS_AS RingCtl* ctl, const int c,
                                              const int lane, const int wg, const int G) {
  constexpr int NTEAM = R_NC / NT;
  constexpr int NS = R_NF / NTEAM;
  const int r31 = lane & 31, half = lane >> 5;
  const int q = c / NT;
  const int stride = NTEAM * G;
  const int nchunk = a.Cin >> 5;
  const int e0 = wg + q * G;
  if (e0 >= a.total) return;
  const uint32_t b_voff = (uint32_t)(r31 * a.Cin + half * 16) * 4u;
  const lptr ring = lds + R_OFF_RING + c * R_D * 4096;
  const lptr frag0 = ring + r31 * 128;                 // + slot * 4096 + swizzled piece
  uint32_t fo[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) fo[p] = (uint32_t)(((half * 4 + p) ^ swz(r31)) << 4);
  const int64_t w_kstride = (int64_t)a.Cout * a.Cin * 4;      // bytes between the weight slices of two offsets
  const int64_t w_bstride = (int64_t)32 * a.Cin * 4;          // ... of two output blocks

  Stamp st;
  const unsigned long long t_begin = now<DIAG>();
  unsigned hcnt_c = 0, prod_c = 0;
  auto get_mask = [&](int i) -> uint32_t {
    if ((unsigned)i >= hcnt_c) {
      const unsigned long long w0 = now<DIAG>();
      spin_until(ctl, a.err, 1u << c, [&]() {
        hcnt_c = __builtin_amdgcn_readfirstlane(ctl->hcnt[c]);
        return (unsigned)i < hcnt_c;
      });
      if (DIAG) st.v[2] += now<DIAG>() - w0;
    }
    return __builtin_amdgcn_readfirstlane(ctl->hmask[c][i % R_HR]);      // (uniform: keeps the step generators scalar)
  };

  // ---- weight-fragment generator: runs R_LB steps ahead of the MFMAs, across work items
  struct BGen {
    int i, ch;
    uint32_t rem;
    bool valid, null;
    ItemPos pos;
    const char* wb;          // WT + by * w_bstride
    const char* wk;          // ... + kk * w_kstride: chunk ch of the step is at wk + 128 ch
  } gb;
  auto bgen_k = [&](BGen& g, int k) {
    const int kk = (a.flip & 1) ? a.K - 1 - k : k;
    g.wk = g.wb + kk * w_kstride;
  };
  auto bgen_item = [&](BGen& g) {
    if (g.pos.e >= a.total) {
      g.valid = false;
      g.wk = reinterpret_cast<const char*>(a.WT);      // past the end: any readable tile (never used)
      g.ch = 0;
      return;
    }
    const uint32_t m = get_mask(g.i);
    g.null = m == 0u;
    g.rem = m & (m - 1u);
    g.ch = 0;
    g.wb = reinterpret_cast<const char*>(a.WT) + g.pos.by * w_bstride;
    bgen_k(g, m ? __builtin_ctz(m) : 0);
  };
  auto bgen_next = [&](BGen& g) {
    if (!g.valid) return;
    if (!g.null && ++g.ch < nchunk) return;
    g.ch = 0;
    if (!g.null && g.rem) {
      bgen_k(g, __builtin_ctz(g.rem));
      g.rem &= g.rem - 1u;
      return;
    }
    ++g.i;
    g.pos.next(stride);
    bgen_item(g);
  };
  auto addrB = [&](const BGen& g) -> const char* {
    const uint64_t bp = reinterpret_cast<uint64_t>(g.wk) + (uint64_t)(g.ch * 128);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)bp);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(bp >> 32));
    return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
  };
  auto loadBat = [&](const char* base, f32x4 (&b)[4]) {
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=v"(b[0]) : "v"(b_voff), "s"(base) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:16" : "=v"(b[1]) : "v"(b_voff), "s"(base) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:32" : "=v"(b[2]) : "v"(b_voff), "s"(base) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:48" : "=v"(b[3]) : "v"(b_voff), "s"(base) : "memory");
  };
  auto loadB = [&](const BGen& g, f32x4 (&b)[4]) { loadBat(addrB(g), b); };
  auto tie = [&](f32x4 (&b)[4]) { asm volatile("" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3])::"memory"); };
  auto readfrag = [&](int slot, f32x4 (&af)[4]) {
    const lptr arow = frag0 + slot * 4096;
#pragma unroll
    for (int p = 0; p < 4; ++p) af[p] = *(const LDS_AS f32x4*)(arow + fo[p]);
  };
  auto wait_prod = [&](unsigned n) {      // step n of this consumer has landed
    if (prod_c <= n) {
      const unsigned long long w0 = now<DIAG>();
      spin_until(ctl, a.err, 0x10u << c, [&]() {
        prod_c = __builtin_amdgcn_readfirstlane(ctl->prod[c]);
        return prod_c > n;
      });
      if (DIAG) {
        st.v[1] += now<DIAG>() - w0;
        ++st.v[6];
      }
    }
  };

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  f32x4 bq[R_LB + 1][4], af[2][4];

  gb.i = 0;
  gb.pos.init(e0, stride, a.gx);
  gb.valid = true;
  bgen_item(gb);
  // prime: weights of the first R_LB steps, fragments of the first step
  loadB(gb, bq[0]);
  bgen_next(gb);
  loadB(gb, bq[1]);
  bgen_next(gb);
  loadB(gb, bq[2]);
  bgen_next(gb);
  wait_prod(0u);
  readfrag(0, af[0]);

  unsigned n = 0;          // steps done
  int slot_next = 1;       // ring slot of step n + 1
  const bool dbg_nomfma = DIAG && (a.flip & 2);      // DIAG experiment: the consumers skip their MFMAs
  auto mfma = [&](const f32x4 (&av)[4], const f32x4 (&bv)[4], int s0, int s1) {
    if (dbg_nomfma) return;
#pragma unroll
    for (int s = s0; s < s1; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s >> 2][s & 3], bv[s >> 2][s & 3], acc, 0, 0, 0);
  };
  // one step with static register sets: weights bq[S], the set bq[(S + R_LB) & 3] is refilled for step n + R_LB
  // (free since step n - 1), fragments af[S & 1] now, af[(S + 1) & 1] for step n + 1.  The bookkeeping sits between
  // groups of four MFMAs: a wave issues in order, and what follows a group issues while its last MFMAs execute
  ItemPos pos;
  pos.init(e0, stride, a.gx);
  unsigned fin_c[NS];
#pragma unroll
  for (int x = 0; x < NS; ++x) fin_c[x] = 0u;
  int i = 0, t = 0, T = 0;
  bool more_items = false, done = false;
  auto item_begin = [&]() {
    const uint32_t mask = get_mask(i);
    T = mask ? __builtin_popcount(mask) * nchunk : 1;
    t = 0;
    more_items = pos.e + stride < a.total;
  };
  // end of a work item: accumulator -> partial buffer (parity i & 1), free once item i - 2 is finished
  auto item_end = [&]() {
    if (i >= 2) {
      const int x = (i - 2) % NS;
      const unsigned need = (unsigned)((i - 2) / NS) + 1u;
#pragma unroll
      for (int xx = 0; xx < NS; ++xx) {
        if (xx == x && fin_c[xx] < need) {
          const unsigned long long w0 = now<DIAG>();
          spin_until(ctl, a.err, 0x1000u << c, [&]() {
            fin_c[xx] = __builtin_amdgcn_readfirstlane(ctl->finp[q][xx]);
            return fin_c[xx] >= need;
          });
        }
      }
    }
    const lptr pb = lds + R_OFF_PART + (c * 2 + (i & 1)) * 4096;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      f32x4 v;
      v[0] = acc[4 * p + 0];
      v[1] = acc[4 * p + 1];
      v[2] = acc[4 * p + 2];
      v[3] = acc[4 * p + 3];
      *(volatile LDS_AS f32x4*)(pb + p * 1024 + lane * 16) = v;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    ctl->pdone[c] = (unsigned)i + 1u;
    pos.next(stride);
    ++i;
    if (pos.e < a.total)
      item_begin();
    else
      done = true;
  };
  // one step with static register sets: weights bq[S], the set bq[(S + R_LB) & 3] is refilled for step n + R_LB
  // (free since step n - 1), fragments af[S & 1] now, af[(S + 1) & 1] for step n + 1.  The bookkeeping sits between
  // groups of four MFMAs: a wave issues in order, and what follows a group issues while its last MFMAs execute.
  // The steps of the consumer's whole stream run through ONE loop of four static copies (a switch over the copies made
  // hipcc keep two versions of every register set: 288 registers)
  // The bookkeeping of a step is cut into pieces that sit between single MFMAs: a wave issues in order, and ~10 scalar
  // instructions fit into the 64 cycles of an MFMA (left to itself hipcc issues the 16 MFMAs back to back and the
  // bookkeeping behind them: 1,500-1,700 cycles per step instead of ~1,050)
#define RING_SB() __builtin_amdgcn_sched_barrier(0)
  auto step = [&](auto SI) {
    constexpr int S = decltype(SI)::value;
    const bool has_next = t + 1 < T || more_items;
    // weights of step n landed: at most the two younger sets may still fly; the fragment reads of this step are back
    const unsigned long long w_top = now<DIAG>();
    asm volatile("s_waitcnt vmcnt(8)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    if (DIAG) st.v[7] += now<DIAG>() - w_top;
    tie(bq[S]);
    RING_SB();
    mfma(af[S & 1], bq[S], 0, 1);
    RING_SB();
    ctl->cons[c] = n + 1u;                 // the ring slot of step n is free
    RING_SB();
    mfma(af[S & 1], bq[S], 1, 2);
    RING_SB();
    if (has_next) wait_prod(n + 1u);
    RING_SB();
    mfma(af[S & 1], bq[S], 2, 3);
    RING_SB();
    if (has_next) {
      readfrag(slot_next, af[(S + 1) & 1]);
      slot_next = slot_next + 1 == R_D ? 0 : slot_next + 1;
    }
    RING_SB();
    mfma(af[S & 1], bq[S], 3, 5);
    RING_SB();
    const char* const wbase = addrB(gb);
    RING_SB();
    mfma(af[S & 1], bq[S], 5, 7);
    RING_SB();
    loadBat(wbase, bq[(S + R_LB) & 3]);
    RING_SB();
    mfma(af[S & 1], bq[S], 7, 9);
    RING_SB();
    bgen_next(gb);
    RING_SB();
    const unsigned long long w_m = now<DIAG>();
    mfma(af[S & 1], bq[S], 9, 16);
    RING_SB();
    if (DIAG) {      // (s_memtime returns once the wave's earlier instructions have ISSUED: + the last MFMA's issue slot)
      asm volatile("s_nop 0" ::: "memory");
      st.v[3] += now<DIAG>() - w_m;
    }
    ++n;
    if (++t == T) item_end();
  };
  item_begin();
  for (;;) {
    step(std::integral_constant<int, 0>{});
    if (done) break;
    step(std::integral_constant<int, 1>{});
    if (done) break;
    step(std::integral_constant<int, 2>{});
    if (done) break;
    step(std::integral_constant<int, 3>{});
    if (done) break;
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // the weight loads past the end
  if (DIAG) {
    st.v[0] = now<DIAG>() - t_begin;
    st.v[4] = n;
    st.v[5] = (unsigned long long)i;
    stamp_out<DIAG>(a, st, c, lane);
  }
}

// ------------------------------------------------------------------------------------------------------------ helper
// Helper wave h has two duties that never wait for one another:
//   * headers of consumer h's work items, one or two items ahead of its loader: the gather-table lines of the consumer's
//     kernel offsets and the tile-order line by LDS-DMA (16-byte pieces: 8 lines each), active-offset mask, the lines
//     rewritten as byte offsets for the loader, output rows of the slice;
//   * finisher of its share of the team's work items: the partial accumulators of the NT consumers added in consumer
//     order, bias / residual, store, BatchNorm slice partials.
template <int NT, bool DIAG>
__device__ __forceinline__ void ring_helper(const RingArgs& a, lptr lds, LDS_AS RingCtl* ctl, const int fw,
                                            const int lane, const int wg, const int G) {
  constexpr int NTEAM = R_NC / NT;
  constexpr int NS = R_NF / NTEAM;
  const int r31 = lane & 31, half = lane >> 5;
  const int stride = NTEAM * G;
  Stamp st;
  const unsigned long long t_begin = now<DIAG>();
  // ---- header side: consumer c = fw
  const int c = fw;
  const int qc = c / NT;
  const uint32_t a_pitch = (uint32_t)a.Cin * 4u;
  const uint32_t pat = own_pattern(NT, c % NT, a.K);
  const int kr = c % NT;
  const int nk = a.nbrS ? __builtin_popcount(pat) : 0;        // lines 0 .. nk - 1: the consumer's offsets (ascending)
  const int nl = nk + (a.order ? 1 : 0);                      // line nk: the tile order (when there is one)
  const lptr tab = lds + R_OFF_TAB + c * 2 * R_TAB_BYTES;
  ItemPos hpos;
  hpos.init(wg + qc * G, stride, a.gx);
  int th = 0;
  bool h_live = hpos.e < a.total;
  unsigned pre_c = 0, iss_c = 0;
  // ---- finisher side: team q, work items i = x + m NS
  const int q = fw % NTEAM, x = fw / NTEAM;
  const int c0 = q * NT;
  ItemPos fpos;
  fpos.init(wg + (q + x * NTEAM) * G, stride * NS, a.gx);
  int fi = x, fm = 0;
  bool f_live = fpos.e < a.total;
  const float* const bias = a.bias;
  const float* const residual = a.residual;
  const BnEpi epi = a.epi;
  float* const stats = a.stats;
  const int Cout = a.Cout;
  unsigned idle = 0;
  bool fetched = false;             // the table pieces of header th are in flight / landed
  int passes_since_fetch = 0;

  auto fetch_table = [&]() {
    const lptr tb = tab + (th & 1) * R_TAB_BYTES;
    const int64_t t0 = (int64_t)hpos.bx * 32;
    const bool full = t0 + 32 <= a.M_out;
    // -- fetch.  Slices of 32 existing rows: pieces of 16 B per lane -- lane l takes entries 4 (l & 7) .. + 3 of line
    // 8 m + (l >> 3): eight lines per piece (sources 4-byte aligned only: the hardware takes unaligned 16-byte loads);
    // the last slice of a tensor: dword pieces, two lines each, rows past the end clamped (masked below)
    if (nl > 0) {
      if (full) {
        const int64_t tl = t0 + 4 * (lane & 7);
        for (int m = 0; m < (nl + 7) / 8; ++m) {
          const int o = m * 8 + (lane >> 3);
          const int32_t* src = (o < nk ? a.nbrS + (int64_t)(kr + o * NT) * a.M_out : a.order) + tl;
          if (o < nl)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (LDS_AS void*)(tb + m * 1024),
                                             16, 0, 0);
        }
      } else {
        int64_t t = t0 + r31;
        if (t >= a.M_out) t = a.M_out - 1;
        for (int m = 0; m < (nl + 1) / 2; ++m) {
          const int o = 2 * m + half;
          const int oo = o < nl ? o : nl - 1;
          const int32_t* src = (oo < nk ? a.nbrS + (int64_t)(kr + oo * NT) * a.M_out : a.order) + t;
          if (o < nl)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (LDS_AS void*)(tb + m * 256), 4,
                                             0, 0);
        }
      }
    }
  };
  auto make_header = [&]() {
    const lptr tb = tab + (th & 1) * R_TAB_BYTES;
    const int64_t t0 = (int64_t)hpos.bx * 32;
    const bool full = t0 + 32 <= a.M_out;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the table pieces (fetched a finish_item ago: landed long since)
    // -- header
    const bool row_ok = t0 + r31 < a.M_out;
    int32_t my_row = a.order ? (int32_t)lds_ld_lane(tb + nk * 128 + r31 * 4) : (int32_t)(t0 + r31);
    if (!row_ok) my_row = -1;
    uint32_t mask = 0u;
    if (a.nbrS && full) {
      for (int m = 0; m < (nk + 7) / 8; ++m) {
        const lptr pa = tb + m * 1024 + lane * 16;
        const u32x4 v = lds_ld128_lane(pa);
        u32x4 w;
        bool any4 = false;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bool ok = (int32_t)v[e] >= 0;
          any4 = any4 || ok;
          w[e] = ok ? v[e] * a_pitch : NO_ROW;
        }
        const bool mine = m * 8 + (lane >> 3) < nk;
        unsigned long long b = __ballot(any4 && mine);
        if (mine) lds_st128(pa, w);
        // byte j of b = the lanes of line 8 m + j: any bit set -> offset kr + (8 m + j) NT is active
        b |= b >> 4;
        b |= b >> 2;
        b |= b >> 1;
        uint32_t sp = (uint32_t)(((b & 0x0101010101010101ull) * 0x0102040810204080ull) >> 56);      // bit j -> bit j NT
        if (NT == 2) {
          sp = (sp | (sp << 4)) & 0x0f0fu;
          sp = (sp | (sp << 2)) & 0x3333u;
          sp = (sp | (sp << 1)) & 0x5555u;
        } else if (NT == 4) {
          sp = (sp | (sp << 12)) & 0x000f000fu;
          sp = (sp | (sp << 6)) & 0x03030303u;
          sp = (sp | (sp << 3)) & 0x11111111u;
        }
        mask |= sp << (kr + m * 8 * NT);
      }
    } else if (a.nbrS) {
      for (int m = 0; m < (nk + 1) / 2; ++m) {
        const uint32_t v = lds_ld_lane(tb + m * 256 + lane * 4);
        const int o = 2 * m + half;
        const bool ok = (int32_t)v >= 0 && row_ok && o < nk;
        const unsigned long long b = __ballot(ok);
        if ((uint32_t)b) mask |= 1u << (kr + 2 * m * NT);
        if ((uint32_t)(b >> 32)) mask |= 1u << (kr + (2 * m + 1) * NT);
        if (o < nk) lds_st(tb + m * 256 + lane * 4, ok ? v * a_pitch : NO_ROW);
      }
    } else if (pat & 1u) {
      // dense 1x1: offset 0 gathers the slice's own rows
      if (lane < 32) lds_st(tb + lane * 4, my_row >= 0 ? (uint32_t)my_row * a_pitch : NO_ROW);
      mask = __ballot(my_row >= 0) ? 1u : 0u;
    }
    mask = __builtin_amdgcn_readfirstlane(mask);      // (hipcc does not see that the ballots are wave-uniform)
    const int hs = th % R_HR;
    if (lane < 32) lds_st(CTL_PTR(rowid) + ((c * R_HR + hs) * 32 + lane) * 4, (uint32_t)my_row);
    lds_st(CTL_PTR(hmask) + (c * R_HR + hs) * 4, mask);
    lds_drain();
    lds_st(CTL_PTR(hcnt) + c * 4, (unsigned)th + 1u);
  };

  auto finish_item = [&]() {
    const int i = fi;
    const int bx = fpos.bx, by = fpos.by;
    const int cch = by * 32 + r31;               // output channel of this lane
    int32_t rows[16];      // rows (reg & 3) + 8 (reg >> 2) + 4 half: four consecutive entries per group
    {
      const lptr rp = CTL_PTR(rowid) + (c0 * R_HR + i % R_HR) * 128 + half * 16;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        uint32_t v[4];
        lds_ld4<4>(rp + g * 32, v);
#pragma unroll
        for (int e = 0; e < 4; ++e) rows[4 * g + e] = (int32_t)v[e];
      }
    }
    const float bv = bias ? bias[cch] : 0.0f;
    float rv[16], xv[16];
    if (residual) {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) rv[reg] = residual[(int64_t)(rows[reg] >= 0 ? rows[reg] : 0) * Cout + cch];
    }
    const bool bn_mode = stats && epi.x;
    BnCoef kc = {0.0f, 0.0f, 0.0f, 0.0f};
    if (bn_mode) {
      kc = bn_coef(epi, cch);
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) xv[reg] = epi.x[(int64_t)(rows[reg] >= 0 ? rows[reg] : 0) * Cout + cch];
    }
    // partial accumulators of the team, added in consumer order
    float val[16];
#pragma unroll
    for (int w = 0; w < NT; ++w) {
      const lptr pb = lds + R_OFF_PART + ((c0 + w) * 2 + (i & 1)) * 4096;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        // (a plain load: hipcc turned the elements of an asm ds_read_b128 result into four copies of the first one here)
        const f32x4 v = *(const volatile LDS_AS f32x4*)(pb + p * 1024 + lane * 16);
#pragma unroll
        for (int xx = 0; xx < 4; ++xx) val[4 * p + xx] = w == 0 ? v[xx] : val[4 * p + xx] + v[xx];
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    lds_st(CTL_PTR(finp) + (q * R_NF + x) * 4, (unsigned)fm + 1u);      // partial buffers and header slot are free
    if (residual) {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) val[reg] = (val[reg] + bv) + rv[reg];
    } else {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) val[reg] = val[reg] + bv;
    }
#pragma unroll
    for (int reg = 0; reg < 16; ++reg)
      if (rows[reg] >= 0) a.out[(int64_t)rows[reg] * Cout + cch] = val[reg];
    if (bn_mode) {                     // BatchNorm-backward partials of the slice (BnEpi)
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
        stats[((int64_t)bx * 2 + 0) * Cout + cch] = sa;
        stats[((int64_t)bx * 2 + 1) * Cout + cch] = sb;
      }
    } else if (stats) {                // forward statistics: (sum, sum of squared deviations from the slice mean)
      float sa = 0.0f, sb = 0.0f;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) sa += rows[reg] >= 0 ? val[reg] : 0.0f;
      sa += __shfl_xor(sa, 32, 64);
      const int64_t left = a.M_out - (int64_t)bx * 32;
      const float mean_s = sa / (float)(left < 32 ? left : 32);
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const float d = rows[reg] >= 0 ? val[reg] - mean_s : 0.0f;
        sb += d * d;
      }
      sb += __shfl_xor(sb, 32, 64);
      if (half == 0) {
        st_sc1(stats + ((int64_t)bx * 2 + 0) * Cout + cch, sa);
        st_sc1(stats + ((int64_t)bx * 2 + 1) * Cout + cch, sb);
      }
    }
  };

  while (h_live || f_live) {
    bool progress = false;
    // (1) the next header: a header slot is free (th < finished prefix + R_HR) and so is the table buffer (the loader is
    // done with item th - 2)
    if (h_live) {
      if ((unsigned)th >= pre_c + R_HR) {
        unsigned pre = 0xffffffffu;
#pragma unroll
        for (int xx = 0; xx < NS; ++xx) {
          const unsigned v = lds_ld(CTL_PTR(finp) + (qc * R_NF + xx) * 4) * NS + xx;
          pre = v < pre ? v : pre;
        }
        pre_c = pre;
      }
      if ((unsigned)th > iss_c + 1u) iss_c = lds_ld(CTL_PTR(iss) + c * 4);
      if (!fetched && (unsigned)th < pre_c + R_HR && (unsigned)th <= iss_c + 1u) {
        fetch_table();               // the pieces fly while this wave finishes a work item (below)
        fetched = true;
        progress = true;
      } else if (fetched && (passes_since_fetch > 0 || !f_live)) {
        const unsigned long long w0 = now<DIAG>();
        make_header();
        if (DIAG) st.v[1] += now<DIAG>() - w0;
        fetched = false;
        passes_since_fetch = 0;
        ++th;
        hpos.next(stride);
        h_live = hpos.e < a.total;
        progress = true;
      }
      if (fetched) ++passes_since_fetch;
    }
    // (2) the next work item to finish: its header is there (always, before its steps) and every consumer of the team
    // has written its partial accumulator
    if (f_live) {
      bool ready = lds_ld(CTL_PTR(hcnt) + c0 * 4) > (unsigned)fi;
#pragma unroll
      for (int w = 0; w < NT; ++w) ready = ready && lds_ld(CTL_PTR(pdone) + (c0 + w) * 4) > (unsigned)fi;
      if (ready) {
        const unsigned long long w0 = now<DIAG>();
        finish_item();
        if (DIAG) {
          st.v[2] += now<DIAG>() - w0;
          ++st.v[3];
        }
        fi += NS;
        ++fm;
        fpos.next(stride * NS);
        f_live = fpos.e < a.total;
        progress = true;
      }
    }
    if (progress) {
      idle = 0;
      continue;
    }
    __builtin_amdgcn_s_sleep(1);
    if (lds_ld(CTL_PTR(abort_))) break;
    if (++idle > R_SPIN) {
      lds_st(CTL_PTR(abort_), 1u);
      if (a.err) atomicOr(a.err, 0x10000u << fw);
      break;
    }
  }
  if (DIAG) {
    st.v[0] = now<DIAG>() - t_begin;
    stamp_out<DIAG>(a, st, R_NC + R_NL + fw, lane);
  }
}

template <int NT, bool DIAG>
__global__ __launch_bounds__(64 * R_WAVES) void spconv_ring_kernel(const RingArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const lptr L = (lptr)lds;
  LDS_AS RingCtl* const ctl = (LDS_AS RingCtl*)L;
  const int tid = (int)threadIdx.x;
  for (int t = tid; t < R_CTL_BYTES / 4; t += 64 * R_WAVES) ((volatile LDS_AS unsigned*)L)[t] = 0u;
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int wg = (int)blockIdx.x, G = (int)gridDim.x;
#ifndef RING_ROLES
#define RING_ROLES 7
#endif
  if (wave < R_NC) {
    if (RING_ROLES & 1) ring_consumer<NT, DIAG>(a, L, ctl, wave, lane, wg, G);
  } else if (wave < R_NC + R_NL) {
    if (RING_ROLES & 2) ring_loader<NT, DIAG>(a, L, ctl, wave - R_NC, lane, wg, G);
  } else {
    if (RING_ROLES & 4) ring_helper<NT, DIAG>(a, L, ctl, wave - R_NC - R_NL, lane, wg, G);
  }
}

}  // namespace

namespace wsis {

// 0: not applicable (the caller keeps spconv_fwd2_kernel), else the team size NT the launch would take
int spconv_ring_plan(int64_t M_out, int K, int Cin, int Cout) {
  // (read per call: tools/ab_step.py switches them inside one process)
  const int mode = env_int("WSIS_RING", 0), nt_force = env_int("WSIS_RING_NT", 0);
  const int min_items = env_int("WSIS_RING_MIN_ITEMS", 400);
  if (!mode) return 0;
  if (K < 1 || K > R_KMAX || Cin % 32 || Cout % 32 || Cin < 32 || Cout < 32) return 0;
  const int64_t items = ceil_div(M_out, 32) * (Cout / 32);
  if (items < min_items || items >= ((int64_t)1 << 30)) return 0;
  if (nt_force == 1 || nt_force == 2 || nt_force == 4) return nt_force;
  // consumers of the chip: 4 per CU.  One consumer per work item while every consumer gets several items; otherwise
  // the offsets of an item are split over 2 or 4 consumers (K = 1 has nothing to split)
  if (K == 1 || items >= 3 * 1024) return 1;
  if (K * (Cin / 32) >= 8 && items < 1536) return 4;
  return K >= 2 ? 2 : 1;
}

int spconv_ring_launch(int nt, const float* d_X, const int32_t* d_nbr, const int32_t* d_order, const float* d_WT, int flip,
                       const float* d_bias, const float* d_residual, float* d_out, float* d_stats, const void* epi_p,
                       unsigned* d_err, int64_t M_in, int64_t M_out, int K, int Cin, int Cout, hipEvent_t ka, hipEvent_t kb, hipStream_t st,
                       unsigned long long* d_dbg) {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    WSIS_HIP_CHECK(hipGetDevice(&dev));
    WSIS_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  }
  RingArgs a;
  a.X = d_X;
  a.nbrS = d_nbr;
  a.order = d_order;
  a.WT = d_WT;
  a.bias = d_bias;
  a.residual = d_residual;
  a.out = d_out;
  a.stats = d_stats;
  a.err = d_err;
  a.dbg = d_dbg;
  a.epi = *static_cast<const BnEpi*>(epi_p);
  a.M_out = M_out;
  a.K = K;
  a.Cin = Cin;
  a.Cout = Cout;
  a.flip = flip;
  a.x_bytes = (uint32_t)(M_in * Cin * 4);
  a.gx = (int)ceil_div(M_out, 32);
  a.total = a.gx * (Cout / 32);
  const int nteam = R_NC / nt;
  int64_t g = ceil_div(a.total, nteam);
  if (g > cus) g = cus;
  if (g < 1) g = 1;
#define WSIS_RING_GO(NTV, DG)                                                                                        \
  do {                                                                                                               \
    static bool attr = false;                                                                                        \
    if (!attr) {                                                                                                     \
      WSIS_HIP_CHECK(hipFuncSetAttribute((const void*)spconv_ring_kernel<NTV, DG>,                                   \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, R_LDS_BYTES));                  \
      attr = true;                                                                                                   \
    }                                                                                                                \
    hipExtLaunchKernelGGL((spconv_ring_kernel<NTV, DG>), dim3((unsigned)g), dim3(64 * R_WAVES), (uint32_t)R_LDS_BYTES, st, \
                          ka, kb, 0u, a);                                                                            \
  } while (0)
  if (d_dbg) {
    if (nt == 1)
      WSIS_RING_GO(1, true);
    else if (nt == 2)
      WSIS_RING_GO(2, true);
    else
      WSIS_RING_GO(4, true);
  } else if (nt == 1)
    WSIS_RING_GO(1, false);
  else if (nt == 2)
    WSIS_RING_GO(2, false);
  else
    WSIS_RING_GO(4, false);
#undef WSIS_RING_GO
  return WSIS_OK;
}

}  // namespace wsis

extern "C" {
// diagnostic (not part of the ABI header): the ring kernel's DIAG build; dbg[grid * 12 waves * 8] cycle counters
// (tools/ring_stamps.py); returns the grid size through *grid_out
int wsis_debug_ring_diag(int32_t nt, const float* d_X, const int32_t* d_nbr, const int32_t* d_order, const float* d_WT,
                         float* d_out, int64_t M_in, int64_t M_out, int32_t K, int32_t Cin, int32_t Cout,
                         unsigned long long* d_dbg, void* stream) {
  BnEpi epi{};
  const int dflags = (nt >> 8) << 1;      // experiment bits (see the kernel)
  nt &= 0xff;
  return wsis::spconv_ring_launch(nt, d_X, d_nbr, d_order, d_WT, dflags, nullptr, nullptr, d_out, nullptr, &epi, nullptr, M_in,
                                  M_out, K, Cin, Cout, nullptr, nullptr, wsis::as_stream(stream), d_dbg);
}
}
