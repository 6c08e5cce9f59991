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
#include "spconv2_body.h"

namespace {

// Workgroup index -> work item.  The grid is ONE-dimensional: the dispatcher hands workgroup i to CU i % n_cu while
// workgroups fit (tools/conv2_stamps.py), and the launch lasts as long as its most loaded CU.  The slices of the tile
// order come heaviest first (rulebook.hip: slice scheduling), the output blocks and offset slabs of a slice weigh the
// same: item j of the weight order = (slice j / (gy gz), block, slab), and the workgroups take the items in a snake over
// bands of `band` (= n_cu) items -- band 0 ascending, band 1 descending, ... -- so every CU gets one item of each weight
// band and the light end of one band meets the heavy end of the next (C2 level 2, 96 -> 96: the most loaded CU carried
// 51 steps against a mean of 26 with the old (slice, block) grid; 33 with this deal).  Which workgroup computes an item
// has no influence on its result.
struct ItemMap {
  int bx, by, bz;
};
__host__ __device__ __forceinline__ ItemMap item_of(int i, int gx, int gy, int gz, int band) {
  const int total = gx * gy * gz;
  ItemMap m;
  const int per = gy * gz;
  if (band < 0) {
    // XCD-chunked deal: workgroup i runs on XCD i % 8 (static round-robin), each XCD with its own 4 MB L2.  XCD x takes
    // runs of C = -band consecutive items of the weight order -- runs x, x + 8, x + 16, ... -- so slices that are
    // neighbours in space (the order is Morton inside a weight class) gather through the same L2; inside a run the
    // direction alternates (the CUs of an XCD take its items round-robin)
    const int C = -band, x = i & 7, q = i >> 3;
    const int nb = total / (8 * C);
    int j = i;
    if (q < nb * C) {
      const int sb = q / C, w = q - sb * C;
      j = (sb * 8 + x) * C + ((sb & 1) ? C - 1 - w : w);
    }
    m.bx = j / per;
    const int r = j - m.bx * per;
    m.by = r / gz;
    m.bz = r - m.by * gz;
    return m;
  }
  const int b = i / band, c = i - b * band;
  int j = i;
  const int F = total / band, R = total - F * band;      // full rounds of the CUs, workgroups of the partial last one
  if (F >= 1 && F <= 3 && R > 0 && band < 0x7fffffff) {
    // a few items per CU, all resident at once (levels 2-4 of a scene): the R CUs that hold one item more get the
    // LIGHTEST (F + 1) R items, the other band - R CUs the heaviest F (band - R), a snake inside each group.  C2 level 2
    // (618 items, weights 81 .. 27 steps): most loaded CU 180 steps with the plain snake -- its partial third band lands
    // on the CUs that already hold the heaviest items -- 135 with this deal, 120 = the greedy bound, mean 106
    const int nB = band - R;
    if (c >= R) {
      const int bb = c - R;
      j = b * nB + ((b & 1) ? nB - 1 - bb : bb);
    } else {
      j = F * nB + b * R + ((b & 1) ? R - 1 - c : c);
    }
  } else if (b & 1) {
    const int left = total - b * band;
    j = b * band + ((left < band ? left : band) - 1 - c);
  }
  m.bx = j / per;
  const int r = j - m.bx * per;
  m.by = r / gz;
  m.bz = r - m.by * gz;
  return m;
}

template <int NB, int NW, int DA, bool BD, bool DIAG = false, bool FB = false, bool TR = false>
__global__ __launch_bounds__(64 * NW) void spconv_fwd2_kernel(
    const float* __restrict__ X, const int32_t* __restrict__ nbrS, const int32_t* __restrict__ order,
    const float* __restrict__ WT, const float* __restrict__ bias, const float* __restrict__ residual,
    float* __restrict__ out, float* __restrict__ partial, int64_t M_out, int K, int Cin, int Cout, int flip_deal,
    uint32_t x_bytes, float* __restrict__ stats, BnEpi epi, BnIn bin, StatFin fin, int gz, int band,
    unsigned long long* __restrict__ dbg = nullptr) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  WgSync sync;
  const LateVals late{bias, residual, out, partial, stats, epi};
  const int gx = (int)((M_out + SL - 1) / SL), gy = Cout / (32 * NB);
  const ItemMap it = item_of((int)blockIdx.x, gx, gy, gz, band);
  fwd2_body<NB, NW, DA, BD, DIAG, FB, false, TR>(X, nbrS, order, WT, late, M_out, K, Cin, Cout, flip_deal, x_bytes, bin, &fin,
                                                dbg, it.bx, it.by, it.bz, gy, gz, lds, (int)threadIdx.x, sync);
}


#if WSIS_EXPERIMENTAL
// ---- table entries in registers (TR) and, per ITEM, the walk: the LDS-DMA ring for the items of the full rounds, the
// register-gather walk (spconv2_body.h: rg_walk) for the waves that end up alone on their SIMD -- items of at least
// `rg_steps` active offsets (flip_deal bits 8..15) and every workgroup dispatched after the first `late_from` (= the
// launch's resident capacity; packed above gz, which is 1 here).  Bit-identical.  Measured (WSIS_FWD2_RGH=12): the long
// items end at 31-38 instead of 43-50 us and a level-0 launch alone takes 50.5-52 instead of 52.6-54.5 us -- and the step,
// where the weight gradients share the texture path, gets SLOWER: 7.763 against 7.732 ms, four scenes 21.40 against 20.86.
template <bool DIAG>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void spconv_fwd2h_kernel(
    const float* __restrict__ X, const int32_t* __restrict__ nbrS, const int32_t* __restrict__ order,
    const float* __restrict__ WT, const float* __restrict__ bias, const float* __restrict__ residual,
    float* __restrict__ out, float* __restrict__ partial, int64_t M_out, int K, int Cin, int Cout, int flip_deal,
    uint32_t x_bytes, float* __restrict__ stats, BnEpi epi, BnIn bin, StatFin fin, int gz_late, int band,
    unsigned long long* __restrict__ dbg = nullptr) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  WgSync sync;
  const LateVals late{bias, residual, out, partial, stats, epi};
  const int gx = (int)((M_out + SL - 1) / SL), gy = Cout / 32;
  const int late_from = gz_late >> 8;
  const ItemMap it = item_of((int)blockIdx.x, gx, gy, 1, band);
  const int fd = flip_deal | ((late_from > 0 && (int)blockIdx.x >= late_from) ? 8 : 0);
  fwd2_body<1, 1, 2, true, DIAG, false, false, true, false, true>(X, nbrS, order, WT, late, M_out, K, Cin, Cout, fd, x_bytes, bin,
                                                                   &fin, dbg, it.bx, it.by, it.bz, gy, 1, lds, (int)threadIdx.x, sync);
}

// ---- one-wave work items with both operands straight to registers (spconv2_body.h: RG).  Four waves per SIMD: the
// register file is what bounds the waves per CU here (5.4 KB of LDS per item), so the kernel is held to 128 registers.
// Bit-identical, measured SLOWER on level 0 (61.4 against 55.2 us): the long items and the last round run a step per
// 1,700-2,500 cycles instead of 2,700-3,600, but a row-per-lane load touches 4x the cache lines per instruction of the
// coalesced LDS-DMA piece and the full rounds go from 5,300 to 7,800 cycles per step.  Kept as WSIS_FWD2_RG=1.
template <bool DIAG>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void spconv_fwd2rg_kernel(
    const float* __restrict__ X, const int32_t* __restrict__ nbrS, const int32_t* __restrict__ order,
    const float* __restrict__ WT, const float* __restrict__ bias, const float* __restrict__ residual,
    float* __restrict__ out, float* __restrict__ partial, int64_t M_out, int K, int Cin, int Cout, int flip_deal,
    uint32_t x_bytes, float* __restrict__ stats, BnEpi epi, BnIn bin, StatFin fin, int gz, int band,
    unsigned long long* __restrict__ dbg = nullptr) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  WgSync sync;
  const LateVals late{bias, residual, out, partial, stats, epi};
  const int gx = (int)((M_out + SL - 1) / SL), gy = Cout / 32;
  const ItemMap it = item_of((int)blockIdx.x, gx, gy, gz, band);
  fwd2_body<1, 1, 2, true, DIAG, false, false, false, true>(X, nbrS, order, WT, late, M_out, K, Cin, Cout, flip_deal, x_bytes, bin,
                                                            &fin, dbg, it.bx, it.by, it.bz, gy, gz, lds, (int)threadIdx.x, sync);
}

// ---- persistent form for launches of more than one round of work items (levels 0-1 of a scene: 4,803 one-wave items
// over 3,072 resident waves are 1.56 rounds whose second round runs on 7 of the 12 wave slots of a CU).  gridDim.x
// workgroups -- as many as are resident at once -- stay on the machine: each starts with work item blockIdx.x and then
// draws further ones from a ticket counter; the tile order puts the heavy slices first, so the light tail fills the
// gaps the heavy ones leave and every workgroup finishes at about the same time.  The ticket of the NEXT item is drawn
// before the current one is computed (its round trip hides behind the item), the counters are sharded eight ways
// (workgroup % 8; 128 bytes apart in the caller's sync slot) and reset by the last draw of the launch.  A work item is
// computed by the same code whoever draws it: results bit-identical to spconv_fwd2_kernel.
template <int NW>
__global__ __launch_bounds__(64 * NW) void spconv_fwd2p_kernel(
    const float* __restrict__ X, const int32_t* __restrict__ nbrS, const int32_t* __restrict__ order,
    const float* __restrict__ WT, const float* __restrict__ bias, const float* __restrict__ residual,
    float* __restrict__ out, int64_t M_out, int K, int Cin, int Cout, int flip_deal, uint32_t x_bytes,
    float* __restrict__ stats, BnEpi epi, unsigned* __restrict__ q_ctr, int total, int gx) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  WgSync sync;
  const int W = (int)gridDim.x, shard = (int)blockIdx.x & 7;
  const int gy = Cout / 32;
  // items W + shard + 8 t, t = 0 .. n_sh - 1, are handed out by this shard's counter; every workgroup of the shard draws
  // until its ticket is past them: n_sh + (workgroups of the shard) draws in all, the last one resets the counter
  const unsigned n_sh = total > W + shard ? (unsigned)((total - W - shard + 7) / 8) : 0u;
  const unsigned pullers = (unsigned)((W - shard + 7) / 8);
  unsigned* const ctr = q_ctr + shard * 32;
  volatile unsigned* const s_next = reinterpret_cast<volatile unsigned*>(lds + HDR_BYTES + NW * Layout<1, 2, true>::WAVE_BYTES);
  int item = (int)blockIdx.x;
  while (item < total) {
    // (nothing of one work item is kept in registers for the next: see deep.hip)
    asm volatile("" ::: "memory");
    int tid = (int)threadIdx.x;
    asm volatile("" : "+v"(tid));
    unsigned t = 0u;
    if (tid == 0) t = atomicAdd(ctr, 1u);
    const int bx = item % gx, by = item / gx;
    const LateVals late{bias, residual, out, nullptr, stats, epi};      // (built per item: behind the clobber above it
    const BnIn bin{};                                                   // would have to live in private memory)
    fwd2_body<1, NW, 2, true, false, false, false>(X, nbrS, order, WT, late, M_out, K, Cin, Cout, flip_deal, x_bytes, bin,
                                                  nullptr, nullptr, bx, by, 0, gy, 1, lds, tid, sync);
    if (NW == 1) {
      t = __builtin_amdgcn_readfirstlane(t);
    } else {
      if (tid == 0) *s_next = t;
      sync();          // also: the next work item re-writes the header and the rings
      t = *s_next;
    }
    if (tid == 0 && t == n_sh + pullers - 1u) __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    item = t < n_sh ? W + shard + 8 * (int)t : total;
  }
}

#endif      // WSIS_EXPERIMENTAL

// out = sum_z partial[z] (+ bias, + residual)
__global__ void spconv2_reduce_kernel(const float4* __restrict__ partial, const float4* __restrict__ bias,
                                      const float4* __restrict__ residual, float4* __restrict__ out, int64_t total4,
                                      int cout4, int zs) {
  reduce_body<false>(partial, bias, residual, out, total4, cout4, zs, blockIdx.x * (int64_t)blockDim.x + threadIdx.x,
                     (int64_t)gridDim.x * blockDim.x);
}

// the same sum for levels whose consumer is a BatchNorm (slice partials of the statistics / of the backward reduction)
__global__ __launch_bounds__(256) void spconv2_reduce_stats_kernel(const float4* __restrict__ partial,
                                                                  const float4* __restrict__ bias,
                                                                  const float4* __restrict__ residual,
                                                                  float4* __restrict__ out, int64_t M_out, int cout4,
                                                                  int zs, float* __restrict__ stats, BnEpi epi) {
  __shared__ float sred[256 * 4];
  WgSync sync;
  reduce_stats_body<false>(partial, bias, residual, out, M_out, cout4, zs, stats, epi, (int)blockIdx.x, (int)blockIdx.y,
                           (int)gridDim.y, (int)threadIdx.x, sred, sync);
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

#if WSIS_EXPERIMENTAL
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

#endif

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
#if WSIS_EXPERIMENTAL
    WSIS_REQUIRE(d_ws && ws_bytes >= wsis_spconv_fwd_f_workspace_bytes(M_out, K, Cin, Cout), "workspace too small");
#else
    return fail(WSIS_ERR_ARG, "in-launch statistics finish: EXPERIMENTAL build only");
#endif
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
  WSIS_REQUIRE(ceil_div(M_out, SL) * (Cout / 32 / p.NB) * p.ZS < ((int64_t)1 << 31), "too many work items");
  const dim3 grid((unsigned)(ceil_div(M_out, SL) * (Cout / 32 / p.NB) * p.ZS), 1u, 1u);
  static int n_cu = 0;                 // snake period of the item deal (item_of): the CUs of the device
  if (!n_cu) {
    int dev = 0, v = 0;
    WSIS_HIP_CHECK(hipGetDevice(&dev));
    WSIS_HIP_CHECK(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev));
    n_cu = v > 0 ? v : 256;
  }
  // launches of many one-wave items (level 0 of a scene: several rounds of the CUs): runs of 64 items per XCD, so the
  // gathers of neighbouring slices share an L2 (level 0, 32 -> 32: 56.3 -> 52.7 us, 64 -> 32: 96.0 -> 86.3; with fewer
  // items per CU the balance of the snake matters more: levels 1-2 measured 0 ... +15 % slower, they keep the snake)
  const int64_t n_items = ceil_div(M_out, SL) * (Cout / (32 * p.NB)) * p.ZS;
  const int xcd_run = tune_int("WSIS_FWD2_XCD", (p.NW == 1 && n_items >= 8 * 64 * 8) ? 64 : 0);
  const int deal_band = xcd_run > 0 ? -xcd_run : tune_int("WSIS_FWD2_SNAKE", 1) ? n_cu : 0x7fffffff;      // (read per call; 0: items in weight order)
  // WSIS_FWD2_DEAL=1 (read per call; default 0): active offsets dealt round-robin to the waves of a work item instead
  // of ownership by offset index.  Measured neutral on the C2 step (level 1: 1046 -> 1035-1050 us per step, level 2:
  // 855 -> 835) -- the waves of a work item are not what its lifetime waits for -- and it gives up the tile-order
  // independence of the results, so it stays off.
  const char* deal_env = tune_env("WSIS_FWD2_DEAL");
  // (wave priorities by step count -- s_setprio 1..3 for waves with many steps, the launch lasts as long as its longest
  // wave -- measured neutral on every level: not kept)
  const int flip_deal = (flip ? 1 : 0) | (((deal_env ? atoi(deal_env) : 0) != 0 && p.ZS == 1) ? 2 : 0);
  ProfScope prof(0, st, /*exact_events=*/true);
#if WSIS_EXPERIMENTAL
  // role-split ring form (spconv3.hip): the levels with many work items
  if (!bn_in && n_targets == 0 && !fused) {
    const int nt = spconv_ring_plan(M_out, K, Cin, Cout);
    if (nt) {
      const int rc = spconv_ring_launch(nt, d_X, d_nbr, d_order, d_WT, flip ? 1 : 0, d_bias, d_residual, d_out, d_stats, &epi,
                                        d_sync ? &static_cast<SyncSlot*>(d_sync)->err : nullptr, M_in, M_out, K, Cin, Cout, prof.ka(), prof.kb(), st);
      if (rc != WSIS_OK) return rc;
      prof.stop();
      WSIS_LAUNCH_CHECK();
      return WSIS_OK;
    }
  }
  // persistent form (spconv_fwd2p_kernel): launches of more than one round of resident work items
  {
    const char* pe = getenv("WSIS_FWD2P");          // (read per call)
    const int fwd2p = pe ? atoi(pe) : 0;
    static int cus = 0;
    if (!cus) {
      int dev = 0;
      WSIS_HIP_CHECK(hipGetDevice(&dev));
      WSIS_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    }
    const int64_t items = ceil_div(M_out, SL) * (Cout / 32);
    // measured (tools/conv2p_bench.py): one scene per step (level 0: 1.56 rounds, level 1: 1.6) the hardware's own
    // dispatch of one-shot workgroups is as good a queue and has no per-item ticket -- 59.8 -> 63.3 us at level 0,
    // 46.0 -> 52.1 at level 1; four scenes per step (6.4 rounds of one-wave items) 202 -> 180 us for the 3x3x3 layers of
    // level 0, while the short 2x2x2 items (42 -> 51) and the 4-wave items (139 -> 152) stay slower.  Inside the step (the
    // weight gradients beside the dIn products) even that gain is gone: 21.92 (off) / 21.97 (one-wave items of >= 16 offsets
    // from 4 rounds on: WSIS_FWD2P=1) / 22.41 ms (wherever it applies: =2) per 4-scene step, in-process A/B -> default OFF
    const bool p_shape = fwd2p >= 2 ? (p.NW == 1 || p.NW == 2 || p.NW == 4) : (p.NW == 1 && K >= 16);
    if (fwd2p && !bn_in && n_targets == 0 && d_sync && p.NB == 1 && p.ZS == 1 && p.DA == 2 && p.BD && p_shape &&
        items < ((int64_t)1 << 30)) {
      const size_t ldsb = (size_t)HDR_BYTES + (size_t)Layout<1, 2, true>::WAVE_BYTES * p.NW + 16;
      int per_cu = (int)((size_t)160 * 1024 / ldsb);
      const int wave_cap = 12 / p.NW > 0 ? 12 / p.NW : 1;      // 3 waves per SIMD: the occupancy the one-shot kernel runs at
      if (per_cu > wave_cap) per_cu = wave_cap;
      const int64_t resident = (int64_t)cus * per_cu;
      const char* me = tune_env("WSIS_FWD2P_MIN");      // rounds (x 100) from which the persistent form is taken
      const int64_t min_pct = me ? atoi(me) : (fwd2p >= 2 ? 110 : 400);
      if (items * 100 >= resident * min_pct) {
        unsigned* q = static_cast<SyncSlot*>(d_sync)->fin;
        const int gx = (int)ceil_div(M_out, SL);
#define WSIS_F2P(nw)                                                                                                   \
  do {                                                                                                                 \
    static bool attr = false;                                                                                          \
    if (!attr) {                                                                                                       \
      WSIS_HIP_CHECK(hipFuncSetAttribute((const void*)spconv_fwd2p_kernel<nw>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                         (int)ldsb));                                                                  \
      attr = true;                                                                                                     \
    }                                                                                                                  \
    hipExtLaunchKernelGGL((spconv_fwd2p_kernel<nw>), dim3((unsigned)resident), dim3(64 * nw), (uint32_t)ldsb, st,      \
                          prof.ka(), prof.kb(), 0u, d_X, d_nbr, d_order, d_WT, d_bias, d_residual, d_out, M_out, K, Cin, \
                          Cout, flip_deal, x_bytes, d_stats, epi, q, (int)items, gx);                                  \
  } while (0)
        if (p.NW == 1)
          WSIS_F2P(1);
        else if (p.NW == 2)
          WSIS_F2P(2);
        else
          WSIS_F2P(4);
#undef WSIS_F2P
        prof.stop();
        WSIS_LAUNCH_CHECK();
        return WSIS_OK;
      }
    }
  }
#endif      // WSIS_EXPERIMENTAL
#define WSIS_F2Y(nb, nw, da, bd, fb, tr)                                                                         \
  do {                                                                                                           \
    const size_t ldsb = (size_t)((tr) ? HDR_TR_BYTES : HDR_BYTES) + (size_t)Layout<nb, da, bd>::WAVE_BYTES * nw + \
                        (fb ? (size_t)Cin * 12 : 0);                                                             \
    static size_t attr_set = 0;                                                                                  \
    if (attr_set < ldsb) {                                                                                       \
      WSIS_HIP_CHECK(hipFuncSetAttribute((const void*)spconv_fwd2_kernel<nb, nw, da, bd, false, fb, tr>,         \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));               \
      attr_set = ldsb;                                                                                           \
    }                                                                                                            \
    hipExtLaunchKernelGGL((spconv_fwd2_kernel<nb, nw, da, bd, false, fb, tr>), grid, dim3(64 * nw), (uint32_t)ldsb, st, \
                          prof.ka(), prof.kb(), 0u, d_X, d_nbr, d_order, d_WT, d_bias, d_residual, d_out, partial,   \
                          M_out, K, Cin, Cout, flip_deal, x_bytes, d_stats, epi, bin, fin, (int)p.ZS, deal_band,    \
                          (unsigned long long*)nullptr);                                                            \
  } while (0)
#define WSIS_F2X(nb, nw, da, bd, fb) WSIS_F2Y(nb, nw, da, bd, fb, false)
#if WSIS_EXPERIMENTAL
#define WSIS_F2RG() /* one-wave items, both operands to registers: LDS = gather table + epilogue scratch */           \
  do {                                                                                                           \
    const size_t ldsb = (size_t)HDR_BYTES + (fin.chunk ? (size_t)Layout<1, 2, true>::WAVE_BYTES : (size_t)RG_SCRATCH_BYTES); \
    hipExtLaunchKernelGGL((spconv_fwd2rg_kernel<false>), grid, dim3(64), (uint32_t)ldsb, st, \
                          prof.ka(), prof.kb(), 0u, d_X, d_nbr, d_order, d_WT, d_bias, d_residual, d_out, partial,   \
                          M_out, K, Cin, Cout, flip_deal, x_bytes, d_stats, epi, bin, fin, (int)p.ZS, deal_band,    \
                          (unsigned long long*)nullptr);                                                            \
  } while (0)
#endif
#define WSIS_F2(nb, nw, da)             \
  if (p.BD)                             \
    WSIS_F2X(nb, nw, da, true, false);  \
  else                                  \
    WSIS_F2X(nb, nw, da, false, false)
#if WSIS_EXPERIMENTAL
#define WSIS_F2B(nw) /* NB = 1, DA = 2, weights to registers: with or without the fused input BatchNorm */ \
  if (bn_in)                                                                                               \
    WSIS_F2X(1, nw, 2, true, true);                                                                        \
  else                                                                                                     \
    WSIS_F2X(1, nw, 2, true, false)
#else
#define WSIS_F2B(nw) WSIS_F2X(1, nw, 2, true, false)
  WSIS_REQUIRE(!bn_in, "fused input BatchNorm: EXPERIMENTAL build only");
#endif
  // one-wave items with the table entries in registers (8.25 instead of 12.3 KB of LDS per item): needs a gather table
  // (the 1x1x1 convolutions have none) and row indices the 24-bit multiply holds
#if WSIS_EXPERIMENTAL
  // per-item walk (spconv_fwd2h_kernel): items of >= rgh_steps active offsets and the workgroups behind the first full
  // round of 16 per CU; only where a launch HAS a second round or long items to speak of (many one-wave items); the late
  // rule only where the launch's last round is a partial SECOND round: with many rounds the workgroups behind the first
  // are the bulk, and a bulk that walks by register gather saturates the texture path
  const int rgh_steps = n_items >= 8 * 64 * 8 ? tune_int("WSIS_FWD2_RGH", 0) & 255 : 0;
  const int rgh_late = (tune_int("WSIS_FWD2_RGH_LATE", 1) && n_items > 16 * n_cu && n_items <= 2 * 16 * n_cu) ? 16 * n_cu : 0;
#endif
  const bool tr_ok = d_nbr != nullptr && !bn_in && M_in < ((int64_t)1 << 24) && tune_int("WSIS_FWD2_TR", 1) != 0;
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
    case 112:
#if WSIS_EXPERIMENTAL
      if (p.BD && !bn_in && tune_int("WSIS_FWD2_RG", 0) != 0)
        WSIS_F2RG();
      else if (p.BD && tr_ok && tune_int("WSIS_FWD2_TR_DA", 2) == 3)
        WSIS_F2Y(1, 1, 3, true, false, true);      // the table's 4 KB as a third ring slot: 12 waves per CU, as before
      else
#endif
#if WSIS_EXPERIMENTAL
      if (p.BD && tr_ok && p.ZS == 1 && rgh_steps > 0) {
        const size_t ldsb = (size_t)HDR_TR_BYTES + (size_t)Layout<1, 2, true>::WAVE_BYTES;
        hipExtLaunchKernelGGL((spconv_fwd2h_kernel<false>), grid, dim3(64), (uint32_t)ldsb, st, prof.ka(), prof.kb(), 0u, d_X,
                              d_nbr, d_order, d_WT, d_bias, d_residual, d_out, partial, M_out, K, Cin, Cout,
                              flip_deal | (rgh_steps << 8), x_bytes, d_stats, epi, bin, fin, 1 | (rgh_late << 8), deal_band,
                              (unsigned long long*)nullptr);
      } else
#endif
      if (p.BD && tr_ok)
        WSIS_F2Y(1, 1, 2, true, false, true);
      else
        WSIS_F2(1, 1, 2);
      break;
    case 123: WSIS_F2(1, 2, 3); break;
    case 122: WSIS_F2(1, 2, 2); break;
    case 142: WSIS_F2(1, 4, 2); break;
    case 143: WSIS_F2(1, 4, 3); break;
    case 144: WSIS_F2(1, 4, 4); break;
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
#undef WSIS_F2Y
#undef WSIS_F2RG
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

// diagnostic (not part of the ABI header; host only, no GPU call): the work item of workgroup i of a launch of
// gx slices x gy blocks x gz slabs dealt over bands of `band` -- the map of item_of, for tests/test_item_deal.py
int wsis_debug_item_of(int32_t i, int32_t gx, int32_t gy, int32_t gz, int32_t band, int32_t* out3) {
  if (!out3 || gx < 1 || gy < 1 || gz < 1 || band == 0 || i < 0 || (int64_t)i >= (int64_t)gx * gy * gz) return -1;
  const ItemMap m = item_of(i, gx, gy, gz, band);
  out3[0] = m.bx;
  out3[1] = m.by;
  out3[2] = m.bz;
  return 0;
}

// diagnostic (not part of the ABI header): the NW = 1 kernel with per-workgroup stamps, dbg[ceil(M/32) * Cout/32 * 8];
// variant 0: weights direct to registers, ring depth 2; 1: both operands through LDS rings, depth 3
int wsis_debug_spconv2_diag(const float* d_X, const int32_t* d_nbr, const int32_t* d_order, const float* d_WT,
                            float* d_out, int64_t M_out, int32_t K, int32_t Cin, int32_t Cout, int32_t variant,
                            unsigned long long* d_dbg, void* d_sync, void* stream) {
  WSIS_REQUIRE(wsis_spconv_fwd_t_supported(K, Cin, Cout) && d_dbg, "bad args");
  const dim3 grid((unsigned)(ceil_div(M_out, SL) * (Cout / 32)), 1, 1);
  hipStream_t st = as_stream(stream);
  const int diag_xcd = tune_int("WSIS_FWD2_XCD", (variant & 0xff) < 2 && grid.x >= 8 * 64 * 8 ? 64 : 0);
  const int diag_band = diag_xcd > 0 ? -diag_xcd : tune_int("WSIS_FWD2_SNAKE", 1) ? 256 : 0x7fffffff;
  const int dflags = (variant >> 8) << 4;      // experiment bits (see the kernel)
  variant &= 0xff;
#if WSIS_EXPERIMENTAL
  if (variant == 0 && tune_int("WSIS_FWD2_RG", 0) != 0) {
    const size_t ldsb = (size_t)HDR_BYTES + (size_t)RG_SCRATCH_BYTES;
    hipLaunchKernelGGL((spconv_fwd2rg_kernel<true>), grid, dim3(64), ldsb, st, d_X, d_nbr, d_order,
                       d_WT, (const float*)nullptr, (const float*)nullptr, d_out, (float*)nullptr, M_out, K, Cin, Cout, dflags,
                       (uint32_t)(M_out * Cin * 4), (float*)nullptr, BnEpi{}, BnIn{}, StatFin{}, 1, diag_band, d_dbg);
  } else if (variant == 0 && d_nbr && tune_int("WSIS_FWD2_TR", 1) != 0 && tune_int("WSIS_FWD2_TR_DA", 2) == 3) {
    const size_t ldsb = (size_t)HDR_TR_BYTES + (size_t)Layout<1, 3, true>::WAVE_BYTES;
    hipLaunchKernelGGL((spconv_fwd2_kernel<1, 1, 3, true, true, false, true>), grid, dim3(64), ldsb, st, d_X, d_nbr, d_order,
                       d_WT, (const float*)nullptr, (const float*)nullptr, d_out, (float*)nullptr, M_out, K, Cin, Cout, dflags,
                       (uint32_t)(M_out * Cin * 4), (float*)nullptr, BnEpi{}, BnIn{}, StatFin{}, 1, diag_band, d_dbg);
  } else
#endif
#if WSIS_EXPERIMENTAL
  if (variant == 0 && d_nbr && tune_int("WSIS_FWD2_TR", 1) != 0 && grid.x >= 8 * 64 * 8 && (tune_int("WSIS_FWD2_RGH", 0) & 255) > 0) {
    const size_t ldsb = (size_t)HDR_TR_BYTES + (size_t)Layout<1, 2, true>::WAVE_BYTES;      // the production form of big launches
    hipLaunchKernelGGL((spconv_fwd2h_kernel<true>), grid, dim3(64), ldsb, st, d_X, d_nbr, d_order,
                       d_WT, (const float*)nullptr, (const float*)nullptr, d_out, (float*)nullptr, M_out, K, Cin, Cout,
                       dflags | ((tune_int("WSIS_FWD2_RGH", 0) & 255) << 8), (uint32_t)(M_out * Cin * 4), (float*)nullptr, BnEpi{},
                       BnIn{}, StatFin{}, 1 | ((tune_int("WSIS_FWD2_RGH_LATE", 1) ? 16 * 256 : 0) << 8), diag_band, d_dbg);
  } else
#endif
  if (variant == 0 && d_nbr && tune_int("WSIS_FWD2_TR", 1) != 0) {      // the production one-wave form
    const size_t ldsb = (size_t)HDR_TR_BYTES + (size_t)Layout<1, 2, true>::WAVE_BYTES;
    hipLaunchKernelGGL((spconv_fwd2_kernel<1, 1, 2, true, true, false, true>), grid, dim3(64), ldsb, st, d_X, d_nbr, d_order,
                       d_WT, (const float*)nullptr, (const float*)nullptr, d_out, (float*)nullptr, M_out, K, Cin, Cout, dflags,
                       (uint32_t)(M_out * Cin * 4), (float*)nullptr, BnEpi{}, BnIn{}, StatFin{}, 1, diag_band, d_dbg);
  } else if (variant == 0) {
    const size_t ldsb = (size_t)HDR_BYTES + (size_t)Layout<1, 2, true>::WAVE_BYTES;
    hipLaunchKernelGGL((spconv_fwd2_kernel<1, 1, 2, true, true>), grid, dim3(64), ldsb, st, d_X, d_nbr, d_order, d_WT,
                       (const float*)nullptr, (const float*)nullptr, d_out, (float*)nullptr, M_out, K, Cin, Cout, dflags,
                       (uint32_t)(M_out * Cin * 4), (float*)nullptr, BnEpi{}, BnIn{}, StatFin{}, 1, diag_band, d_dbg);
  } else if (variant >= 2) {      // the production small-level form: 4 waves per work item, `variant - 1` offset slabs
    const dim3 g4((unsigned)(ceil_div(M_out, SL) * (Cout / 32) * (variant - 1)), 1, 1);
    const size_t ldsb = (size_t)HDR_BYTES + (size_t)Layout<1, 2, true>::WAVE_BYTES * 4;
    hipLaunchKernelGGL((spconv_fwd2_kernel<1, 4, 2, true, true>), g4, dim3(256), ldsb, st, d_X, d_nbr, d_order, d_WT,
                       (const float*)nullptr, (const float*)nullptr, d_out, d_out, M_out, K, Cin, Cout, dflags,
                       (uint32_t)(M_out * Cin * 4), (float*)nullptr, BnEpi{}, BnIn{}, StatFin{}, variant - 1, diag_band, d_dbg);
  } else {
    const size_t ldsb = (size_t)HDR_BYTES + (size_t)Layout<1, 3, false>::WAVE_BYTES;
    hipLaunchKernelGGL((spconv_fwd2_kernel<1, 1, 3, false, true>), grid, dim3(64), ldsb, st, d_X, d_nbr, d_order, d_WT,
                       (const float*)nullptr, (const float*)nullptr, d_out, (float*)nullptr, M_out, K, Cin, Cout, 0,
                       (uint32_t)(M_out * Cin * 4), (float*)nullptr, BnEpi{}, BnIn{}, StatFin{}, 1, diag_band, d_dbg);
  }
  WSIS_LAUNCH_CHECK();
  return WSIS_OK;
}

}  // extern "C"
