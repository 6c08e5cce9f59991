// Shared helpers for libwsis_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdlib>

#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/wsis_hip.h"

namespace wsis {
// Tuning knobs (launch plans, scheduling periods, grid sizes ...) are LIVE in the EXPERIMENTAL build only -- the in-process
// A/B tools sweep them there -- and compile to their measured defaults in the default library: what bench.py runs reads
// a dozen switches (selectors the tests need), not ninety.
inline const char* tune_env(const char* name) {
#if WSIS_EXPERIMENTAL
  return getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}
inline int tune_int(const char* name, int dflt) {
  const char* e = tune_env(name);
  return e ? atoi(e) : dflt;
}

std::string& err_slot();
int fail(int code, const char* fmt, ...);

#define WSIS_HIP_CHECK(expr)                                                          \
  do {                                                                                \
    hipError_t e_ = (expr);                                                           \
    if (e_ != hipSuccess)                                                             \
      return ::wsis::fail(WSIS_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                          __FILE__, __LINE__);                                        \
  } while (0)

#define WSIS_LAUNCH_CHECK() WSIS_HIP_CHECK(hipGetLastError())

#define WSIS_REQUIRE(cond, msg)                                     \
  do {                                                              \
    if (!(cond)) return ::wsis::fail(WSIS_ERR_ARG, "%s: %s", __func__, msg); \
  } while (0)

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// grid size for a grid-stride memory-bound kernel: cap at 8 blocks per CU (guide: Guideline 11)
inline int grid_for(int64_t work_items, int block) {
  int64_t g = ceil_div(work_items, block);
  if (g < 1) g = 1;
  if (g > 256 * 8) g = 256 * 8;
  return (int)g;
}

// ---- live kernel timing for bench.py's roofline: HIP events recorded on the launch stream directly around the
// dominant kernels (not around the host wrapper), enabled by wsis_prof_enable().
struct ProfRec {
  hipEvent_t a, b, c;      // a -> b: the main kernel; b -> c: the fixed-order slab sum that finishes it (c == nullptr: none)
  // a product that ran as phases of the resident deep-level kernel (deep.hip): no events; its time is the difference of
  // two in-kernel s_memrealtime stamps (100 MHz) -- from the end of the previous phase to the end of the (last) phase of
  // the product, grid barrier included: stamps[s1] - stamps[s0] (main: [s0, sm])
  const unsigned long long* d_stamps = nullptr;
  int s0 = 0, sm = 0, s1 = 0;
};
inline bool g_prof_on = false;
inline std::vector<ProfRec> g_prof[3];   // 0 = spconv_fwd_kernel, 1 = spconv_dw_kernel, 2 = BatchNorm ops of wsis_run_ops (all launches of an op)
inline std::vector<void*> g_prof_bufs;    // stamp buffers of profiled resident launches (freed when profiling restarts)

struct ProfScope {
  int which;
  hipStream_t st;
  ProfRec r{};
  int stage = 0;           // 1: a recorded (or, exact: created), 2: b recorded
  bool exact = false;      // a / b are the start / stop events of ONE launch (hipExtLaunchKernelGGL): its own duration, as
                           // rocprofv3 reads it -- events recorded around a launch also measure the two marker packets
  ProfScope(int w, hipStream_t s, bool exact_events = false) : which(w), st(s) {
    if (g_prof_on && hipEventCreate(&r.a) == hipSuccess && hipEventCreate(&r.b) == hipSuccess) {
      exact = exact_events && prof_exact();
      stage = exact ? 1 : (hipEventRecord(r.a, st) == hipSuccess ? 1 : 0);
    }
  }
  static bool prof_exact() {
    static int on = -1;
    if (on < 0) {
      const char* e = getenv("WSIS_PROF_EXACT");
      on = e ? atoi(e) : 1;
    }
    return on != 0;
  }
  void bracket() {          // this launch does not take start / stop events: record around it after all
    if (exact && stage == 1) {
      exact = false;
      stage = hipEventRecord(r.a, st) == hipSuccess ? 1 : 0;
    }
  }
  hipEvent_t ka() const { return (exact && stage == 1) ? r.a : nullptr; }      // for hipExtLaunchKernelGGL
  hipEvent_t kb() const { return (exact && stage == 1) ? r.b : nullptr; }
  void stop() {             // behind the main kernel
    if (stage == 1) {
      if (!exact) (void)hipEventRecord(r.b, st);
      stage = 2;
    }
  }
  void tail() {             // behind the launch(es) that finish the product (slab sums): counted with it
    if (stage == 2 && hipEventCreate(&r.c) == hipSuccess) (void)hipEventRecord(r.c, st);
  }
  ~ProfScope() {
    if (stage == 2) g_prof[which].push_back(r);
  }
};


#ifdef __HIPCC__
// sc1 (write-through) stores / loads of the words that another workgroup of the SAME launch reads (statistics partials,
// chunk rows): no release fence, no acquire -- cdna_hip_programming.md Guideline 16, counter form:
// {sc1 stores -> s_waitcnt vmcnt(0) in every storing wave -> barrier -> one agent-scope atomic add}; the workgroup whose
// add came back last reads with sc1 loads
__device__ __forceinline__ void st_sc1(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_sc1(const float* p) {
  return __hip_atomic_load(const_cast<float*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_sc1(const double* p) {
  return __hip_atomic_load(const_cast<double*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void wait_stores_left() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#endif

// ---- cross-workgroup sync words of the one-launch reductions (no global mutable state).  The words live in
// CALLER memory: a slot is 4 KiB, zero-filled once by the caller; every kernel that uses a slot leaves it zero
// again (the last arrival of a ticket resets it, the last workgroup to leave a flag wait resets flag and counter), so
// consecutive launches on one stream can share a slot; launches that may overlap (different streams) need different
// slots.  A NULL slot selects the multi-launch form of the operator.
struct SyncSlot {
  unsigned ticket[16];     // one arrival counter per channel group
  unsigned done, flag, left, err;
  unsigned ctr[44];        // slice-queue counters of spconv_fwd3_kernel: one per (output block, offset slab)
  unsigned fin[960];       // chunk / final tickets of the in-launch statistics finish of spconv_fwd2_kernel
};
static_assert(sizeof(SyncSlot) == 4096, "sync slot layout");
constexpr int kSyncSlots = 64;
constexpr int kSyncSlotsTotal = kSyncSlots + 1;      // + the barrier words of the resident deep-level kernel (deep.hip)
inline SyncSlot* sync_slot(void* d_sync, int i) {
  return d_sync ? static_cast<SyncSlot*>(d_sync) + (i % kSyncSlots) : nullptr;
}
inline void* deep_sync_slot(void* d_sync) { return d_sync ? static_cast<SyncSlot*>(d_sync) + kSyncSlots : nullptr; }

// mean / biased variance (+ running statistics) of one channel from the fp64 totals of the centred slice partials:
// S = sum of slice sums, Q = sum of the slices' centred sums of squares, W = sum S_i^2 / n_i (Chan's combination)
__device__ __forceinline__ void bn_finish_centred(double S, double Q, double W, int64_t M, int c, float* mean, float* var,
                                                  float* running_mean, float* running_var, float momentum) {
  const double n = (double)M;
  const double mu = S / n;
  double v = (Q + (W - n * mu * mu)) / n;
  if (v < 0.0) v = 0.0;
  mean[c] = (float)mu;
  var[c] = (float)v;
  if (running_mean) {
    const double unb = n > 1 ? v * n / (n - 1) : v;
    running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mu);
    running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unb);
  }
}

// number of chunks / partial rows per chunk of the two-level statistics finish (bn.hip and the in-launch finish of
// spconv2.hip must agree: identical order of additions)
constexpr int kBnFinChunks = 64;
inline int bn_fin_chunks(int64_t n_part) {
  int64_t g = n_part / 64;           // at least 64 partials per chunk
  if (g < 1) g = 1;
  if (g > kBnFinChunks) g = kBnFinChunks;
  return (int)g;
}

// 6-channel input convolution on the matrix cores (csrc/spconv_in.hip), dispatched from wsis_spconv_fwd
bool spconv_in_supported(int K, int Cin, int Cout);
int spconv_in_launch(const float* d_X, const int32_t* d_nbr, const int32_t* d_order, const float* d_W, const float* d_bias,
                     const float* d_residual, float* d_out, int64_t M_out, hipEvent_t ka, hipEvent_t kb, hipStream_t st);

int64_t spconv_in_dw_workspace_bytes(int64_t M_out);
int spconv_in_dw_launch(const float* d_X, const int32_t* d_nbr, const int32_t* d_order, const float* d_dY, float* d_dW,
                        int64_t M_out, void* d_ws, hipEvent_t ka, hipEvent_t kb, hipStream_t st);

// role-split ring convolution (csrc/spconv3.hip), dispatched from wsis_spconv_fwd_t*: plan = 0 (not taken) or the team
// size; epi = the BnEpi of the launch
int spconv_ring_plan(int64_t M_out, int K, int Cin, int Cout);
int spconv_ring_launch(int nt, const float* d_X, const int32_t* d_nbr, const int32_t* d_order, const float* d_WT, int flip,
                       const float* d_bias, const float* d_residual, float* d_out, float* d_stats, const void* epi,
                       unsigned* d_err, int64_t M_in, int64_t M_out, int K, int Cin, int Cout, hipEvent_t ka, hipEvent_t kb, hipStream_t st,
                       unsigned long long* d_dbg = nullptr);

// wave-autonomous weight-gradient kernel (csrc/spconv_dw2.hip), dispatched from wsis_spconv_dw
bool dw2_supported(int K, int Cin, int Cout);
bool dw2_fits(int64_t M_in, int64_t M_out, int K, int Cin, int Cout);   // 32-bit buffer offsets
int64_t dw2_workspace_bytes(int64_t M_out, int K, int Cin, int Cout);
void dw2_set_batch_rows(int64_t rows);      // launch-plan hint of the weight-gradient launches (wsis_hint_batch_rows)
int64_t dw2_batch_rows();                   // the hint as last set (0: none)
// Deferred slab sums (the op-list executor, round 6): with a record slot set for the calling thread, dw2_launch /
// dw2_launch_swapped issue the main kernel only and describe the fixed-order slab sum that finishes the product in the
// slot; the executor finishes all products of (a part of) a backward pass with ONE dw2_reduce_batch launch -- same
// arithmetic and order per product as the per-launch sum (bit-identical), without 55 launches of a few microseconds
// each.  The slabs of every deferred product need their own workspace until that launch has run.
struct DwRedRec {
  const float* partial = nullptr;      // nullptr: the product finished itself (shape outside the dw2 kernel, empty level)
  float* dW = nullptr;
  int64_t total4 = 0;
  int32_t P = 0;
};
void dw2_set_defer(DwRedRec* slot);    // thread-local; nullptr switches deferral off
int dw2_reduce_batch(const DwRedRec* recs, int n, hipStream_t st);
int dw2_launch(const float* d_X, const int32_t* d_nbr, const int32_t* d_order, const float* d_dY, float* d_dW,
               int64_t M_in, int64_t M_out, int K, int Cin, int Cout, void* d_ws, hipStream_t st);
int dw2_launch_swapped(const float* d_X, const float* d_mean, const float* d_var, const float* d_gamma, const float* d_beta,
                       float eps, int relu, const int32_t* d_nbr_b, const int32_t* d_order_b, int flip, const float* d_dY,
                       float* d_dW, int64_t M_in, int64_t M_out, int K, int Cin, int Cout, void* d_ws, hipStream_t st);

// row / column of element t of a row-major [rows, C] tensor without a 64-bit division per element: a runtime int64
// division is ~100 instructions on this ISA -- in segment_bwd_kernel it, not HBM, set the time (21 us for 25 MB).
// Powers of two shift, everything below 2^32 elements divides in 32 bits.
#ifdef __HIPCC__
struct RowCol {
  int64_t row;
  int col;
};
__device__ __forceinline__ RowCol row_col(int64_t t, int C, int64_t total) {
  RowCol rc;
  if ((C & (C - 1)) == 0) {
    const int sh = __ffs(C) - 1;
    rc.row = t >> sh;
    rc.col = (int)(t & (C - 1));
  } else if (total < ((int64_t)1 << 32)) {
    const uint32_t q = (uint32_t)t / (uint32_t)C;
    rc.row = q;
    rc.col = (int)((uint32_t)t - q * (uint32_t)C);
  } else {
    rc.row = t / C;
    rc.col = (int)(t - rc.row * C);
  }
  return rc;
}
#endif

// ---- device-side hash (linear-index keys) --------------------------------------------------
constexpr int64_t kEmptyKey = -1;

__device__ __forceinline__ uint64_t mix64(uint64_t x) {
  x ^= x >> 33;
  x *= 0xff51afd7ed558ccdULL;
  x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ULL;
  x ^= x >> 33;
  return x;
}

__device__ __forceinline__ int32_t hash_lookup(const int64_t* __restrict__ keys,
                                               const int32_t* __restrict__ vals, uint64_t mask,
                                               int64_t key) {
  uint64_t h = mix64((uint64_t)key) & mask;
  for (;;) {
    int64_t k = keys[h];
    if (k == key) return vals[h];
    if (k == kEmptyKey) return -1;
    h = (h + 1) & mask;
  }
}

}  // namespace wsis
