// Op-list executor (runtime of the sparse UNet hot path): the host side records one forward or backward pass of
// the UNet as a flat array of wsis_op records -- plain pointers and sizes -- and this file issues all of their
// kernels from ONE C call on one stream.  The reference runs the same sequence as ~100 (forward) / ~350
// (backward) separate Python-dispatched autograd nodes (modules/model/sparse_unet3d.py:103-350); issuing them
// natively removes the per-node interpreter / dispatcher / allocator cost that made the step host-bound.
// Every op maps 1:1 onto the single-op entry points of this library (same kernels, same summation orders, so the
// results are bit-identical to calling them one by one); CAT / SPLIT / ADD are the elementwise glue of the UNet
// (torch.cat of the skip connection, its backward, gradient accumulation at the residual fan-out).
#include <optional>
#include <condition_variable>
#include <cstdlib>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include <unistd.h>

#include <algorithm>

#include "common.h"
#if WSIS_EXPERIMENTAL
#include "deep.h"
#endif

using namespace wsis;

namespace {

constexpr int64_t ALIGN = 256;
inline int64_t up(int64_t v) { return (v + ALIGN - 1) / ALIGN * ALIGN; }

// out[r] = [a[r] | b[r]]  (float4 granules when both widths are multiples of 4)
template <int W>
__global__ void cat_rows_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                                int64_t M, int Ca, int Cb) {
  const int Ca_w = Ca / W, C_w = (Ca + Cb) / W;
  const int64_t total = M * C_w;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const RowCol rc = row_col(t, C_w, total);
    const int64_t r = rc.row;
    const int c = rc.col;
    const float* src = c < Ca_w ? a + (r * Ca_w + c) * W : b + (r * (C_w - Ca_w) + (c - Ca_w)) * W;
    if (W == 4)
      reinterpret_cast<float4*>(out)[t] = *reinterpret_cast<const float4*>(src);
    else
      out[t] = *src;
  }
}

template <int W>
__global__ void split_rows_kernel(const float* __restrict__ in, float* __restrict__ a, float* __restrict__ b, int64_t M,
                                  int Ca, int Cb) {
  const int Ca_w = Ca / W, C_w = (Ca + Cb) / W;
  const int64_t total = M * C_w;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const RowCol rc = row_col(t, C_w, total);
    const int64_t r = rc.row;
    const int c = rc.col;
    float* dst = c < Ca_w ? a + (r * Ca_w + c) * W : b + (r * (C_w - Ca_w) + (c - Ca_w)) * W;
    if (W == 4)
      *reinterpret_cast<float4*>(dst) = reinterpret_cast<const float4*>(in)[t];
    else
      *dst = in[t];
  }
}

__global__ void add_inplace_kernel(float* __restrict__ dst, const float* __restrict__ src, int64_t n) {
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x)
    dst[t] += src[t];
}

inline bool vec4_ok(const wsis_op& op, const void* p0, const void* p1, const void* p2) {
  return (op.Cin % 4 == 0) && (op.Cout % 4 == 0) &&
         ((reinterpret_cast<uintptr_t>(p0) | reinterpret_cast<uintptr_t>(p1) | reinterpret_cast<uintptr_t>(p2)) & 15) == 0;
}

// every dIn convolution of a backward pass needs W[k]^T (flipped for submanifold tables): all of them are produced
// by ONE launch up front instead of one small launch per layer.  One workgroup moves a 32x32 tile of one [Cin, Cout]
// slice through LDS: reads run along Cout, writes along Cin, both coalesced (an element-per-thread version with
// stride-Cout reads took 103 us for the 49 layers of the UNet, 11 M weights).
constexpr int WT_MAX = 64;
constexpr int WT_TILE = 32;
struct WtBatch {
  const float* src[WT_MAX];
  float* dst[WT_MAX];
  int K[WT_MAX], Cin[WT_MAX], Cout[WT_MAX], flip[WT_MAX];
  int start[WT_MAX + 1];   // first tile of each layer
  int n;
};

__global__ __launch_bounds__(256) void weight_transpose_batch_kernel(WtBatch b) {
  __shared__ float tile[WT_TILE][WT_TILE + 1];
  const int t = blockIdx.x;
  int lo = 0, hi = b.n - 1;           // largest i with start[i] <= t (uniform over the workgroup)
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (b.start[mid] <= t)
      lo = mid;
    else
      hi = mid - 1;
  }
  const int Cin = b.Cin[lo], Cout = b.Cout[lo], K = b.K[lo];
  const int ti = (Cin + WT_TILE - 1) / WT_TILE, to = (Cout + WT_TILE - 1) / WT_TILE;
  int u = t - b.start[lo];
  const int bo = u % to;
  u /= to;
  const int bi = u % ti;
  const int kt = u / ti;               // slice of the OUTPUT [K, Cout, Cin]
  const int ks = b.flip[lo] ? (K - 1 - kt) : kt;
  const float* __restrict__ src = b.src[lo] + (int64_t)ks * Cin * Cout;
  float* __restrict__ dst = b.dst[lo] + (int64_t)kt * Cin * Cout;
  const int x = threadIdx.x & 31, y0 = threadIdx.x >> 5;
#pragma unroll
  for (int y = y0; y < WT_TILE; y += 8) {
    const int ci = bi * WT_TILE + y, co = bo * WT_TILE + x;
    if (ci < Cin && co < Cout) tile[y][x] = src[(int64_t)ci * Cout + co];
  }
  __syncthreads();
#pragma unroll
  for (int y = y0; y < WT_TILE; y += 8) {
    const int co = bo * WT_TILE + y, ci = bi * WT_TILE + x;
    if (ci < Cin && co < Cout) dst[(int64_t)co * Cin + ci] = tile[x][y];
  }
}

// ---- weight-gradient side stream: dW of a layer is a leaf of the backward graph (nothing downstream reads it
// before the optimizer), while dIn feeds the next layer.  All dW launches of a pass go to one library-owned side
// stream (fork: an event after the producer of dY; join: one event at the end of the pass), so they overlap the
// dIn / BatchNorm-backward chain -- at the deep levels neither kernel fills 256 CUs on its own.  dW has its own
// workspace region.  WSIS_DW_STREAM=0 keeps everything on the caller's stream.
struct SideStream {
  bool pending_join = false;      // a part of a pass forked weight gradients and left the join to a later part
  hipStream_t stream = nullptr;
  hipEvent_t join = nullptr;
  hipEvent_t mark_main = nullptr, mark_side = nullptr;   // milestone of wsis_run_ops_marked
  hipEvent_t wt_fork = nullptr, wt_done = nullptr;       // the pass's weight transposes on the side stream (run_ops_impl)
  std::vector<hipEvent_t> fork;
  size_t next = 0;
};

SideStream* side_stream_for(hipStream_t main) {
  static std::mutex mu;
  static std::map<std::pair<int, hipStream_t>, SideStream> pool;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  SideStream& s = pool[std::make_pair(dev, main)];
  if (!s.stream) {
    // (created at the device's lowest queue priority the side stream starves: 9.7 -> 24 ms per step)
    if (hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) != hipSuccess) return nullptr;
    if (hipEventCreateWithFlags(&s.join, hipEventDisableTiming) != hipSuccess) return nullptr;
    if (hipEventCreateWithFlags(&s.mark_main, hipEventDisableTiming) != hipSuccess) return nullptr;
    if (hipEventCreateWithFlags(&s.mark_side, hipEventDisableTiming) != hipSuccess) return nullptr;
    if (hipEventCreateWithFlags(&s.wt_fork, hipEventDisableTiming) != hipSuccess) return nullptr;
    if (hipEventCreateWithFlags(&s.wt_done, hipEventDisableTiming) != hipSuccess) return nullptr;
  }
  return &s;
}

hipEvent_t next_fork_event(SideStream* s) {
  if (s->next == s->fork.size()) {
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
    s->fork.push_back(e);
  }
  return s->fork[s->next++];
}

// ---- issuing the weight-gradient launches from a second host thread.  A backward pass is ~250 launches on the
// caller's stream plus, per convolution, an event wait and two launches on the side stream; the issuing thread is the
// longer side of a training step, and the side-stream calls do not depend on anything the caller's
// thread does next.  The caller records the fork event and hands (op, event) to this worker, which makes the side
// stream wait for the event and launches the dW product; the caller waits for the worker to drain before it records a
// milestone or the join event.  One worker per process (lazily started); WSIS_DW_THREAD=0: everything from the caller.
struct DwTask {
  std::thread::id owner;      // the host thread that pushed it (set by DwWorker::push)
  wsis_op op;
  hipEvent_t ev;
  int dev;
  hipStream_t side;
  char* ws;
  int64_t ws_bytes;
  DwRedRec* rec = nullptr;          // deferred slab sum: the product's record slot (nullptr: the product finishes itself)
  const DwRedRec* flush = nullptr;  // != nullptr: no product -- ONE launch finishes the deferred products flush[0 .. flush_n)
  int flush_n = 0;
};
static int issue_dw_raw(const wsis_op& op, char* dw_ws, int64_t dw_bytes, void* dw_stream) {
  if (op.flags & WSIS_OPF_BN_IN)     // the forward input was relu(bn(in[0])) applied on the fly: own-rows form
    return wsis_spconv_dw_bn((const float*)op.in[0], (const float*)op.in[7], (const float*)op.in[8], (const float*)op.in[9],
                             (const float*)op.in[10], op.eps, (op.flags & WSIS_OPF_RELU) ? 1 : 0, (const int32_t*)op.in[5],
                             (const int32_t*)op.in[6], (op.flags & WSIS_OPF_FLIP) ? 1 : 0, (const float*)op.in[2],
                             (float*)op.out[1], op.M_in, op.M_out, op.K, op.Cin, op.Cout, dw_ws, dw_bytes, dw_stream);
  return wsis_spconv_dw((const float*)op.in[0], (const int32_t*)op.in[3], (const int32_t*)op.in[4], (const float*)op.in[2],
                        (float*)op.out[1], op.M_in, op.M_out, op.K, op.Cin, op.Cout, dw_ws, dw_bytes, dw_stream);
}
// rec != nullptr: the product's slab sum is deferred (the dw2 kernel fills the slot; any other path finishes by itself
// and leaves rec->partial == nullptr)
int issue_dw(const wsis_op& op, char* dw_ws, int64_t dw_bytes, void* dw_stream, DwRedRec* rec = nullptr) {
  if (rec) *rec = DwRedRec{};
  dw2_set_defer(rec);
  const int rc = issue_dw_raw(op, dw_ws, dw_bytes, dw_stream);
  dw2_set_defer(nullptr);
  return rc;
}
class DwWorker {
 public:
  void push(DwTask t) {
    std::lock_guard<std::mutex> lock(mu_);
    const long pid = (long)getpid();
    if (started_ && pid_ != pid) {      // a forked child inherits the flag but not the thread: start over
      started_ = false;
      q_.clear();
      st_.clear();
    }
    if (!started_) {
      started_ = true;
      pid_ = pid;
      std::thread(&DwWorker::run, this).detach();
    }
    t.owner = std::this_thread::get_id();
    q_.push_back(t);
    ++st_[t.owner].pending;
    cv_.notify_one();
  }
  // waits until every task THIS thread pushed has been issued; returns the first error among them (and its message):
  // two host threads driving passes on different streams neither wait for nor see each other's tasks
  int drain(std::string* msg) {
    const std::thread::id me = std::this_thread::get_id();
    std::unique_lock<std::mutex> lock(mu_);
    done_.wait(lock, [&] { return st_[me].pending == 0; });
    State& s = st_[me];
    const int rc = s.err;
    if (rc != WSIS_OK && msg) *msg = s.msg;
    s.err = WSIS_OK;
    s.msg.clear();
    return rc;
  }

 private:
  struct State {
    int pending = 0, err = WSIS_OK;
    std::string msg;
  };
  void run() {
    for (;;) {
      DwTask t;
      {
        std::unique_lock<std::mutex> lock(mu_);
        cv_.wait(lock, [this] { return !q_.empty(); });
        t = q_.front();
        q_.pop_front();
      }
      int rc = WSIS_OK;
      std::string msg;
      hipError_t he = hipSetDevice(t.dev);
      if (he == hipSuccess && t.ev) he = hipStreamWaitEvent(t.side, t.ev, 0);
      if (he != hipSuccess) {
        rc = WSIS_ERR_HIP;
        msg = std::string("dW worker: ") + hipGetErrorString(he);
      } else {
        rc = t.flush ? dw2_reduce_batch(t.flush, t.flush_n, t.side) : issue_dw(t.op, t.ws, t.ws_bytes, t.side, t.rec);
        if (rc != WSIS_OK) msg = wsis_last_error();
      }
      std::lock_guard<std::mutex> lock(mu_);
      State& s = st_[t.owner];
      if (rc != WSIS_OK && s.err == WSIS_OK) {
        s.err = rc;
        s.msg = msg;
      }
      if (--s.pending == 0) done_.notify_all();
    }
  }
  std::mutex mu_;
  std::condition_variable cv_, done_;
  std::deque<DwTask> q_;
  std::map<std::thread::id, State> st_;
  bool started_ = false;
  long pid_ = 0;
};
DwWorker& dw_worker() {
  static DwWorker* w = new DwWorker();      // never destroyed: its thread may outlive static destruction
  return *w;
}
bool dw_thread_enabled() {
  const char* e = tune_env("WSIS_DW_THREAD");
  return e ? atoi(e) != 0 : true;
}

// deferred slab sums (one dw2_reduce_batch launch per part of a pass instead of one small launch per product); read per
// pass.  Off while the profiler brackets every product with events: a product's duration then includes its own sum.
// Measured (tools/r06_ab_reduce.sh, profiles/r06_ab_reduce_c*.txt): bit-identical and SLOWER in the step -- 7.75 -> 7.90 ms
// at one scene, 20.36 -> 20.42 at four: the per-product sums hide under the dIn products as they go, the batched launch
// reads every slab of the pass from HBM (0.44 GB at one scene) at the end of the pass, in front of the join.  Opt-in.
bool dw_batch_reduce_enabled() {
  const char* e = getenv("WSIS_DW_BATCH_REDUCE");
  return (e ? atoi(e) != 0 : false) && !g_prof_on;
}

// The weight gradient of the LAST op of a pass that has no dIn product of its own (the input convolution: its dY is the
// final result of the main stream's chain) goes on the CALLER's stream: the main stream has nothing left to run, so the
// product overlaps the previous layer's weight gradient still running on the side stream instead of queueing behind it
// -- the tail of a backward pass that nothing hides (one scene: ~55 us of a 7.8 ms step).  Its slabs use the op workspace
// (free: the op issues nothing else).  WSIS_DW_TAIL_MAIN=0 (read per pass): everything on the side stream.
bool dw_tail_main_enabled() {
  const char* e = getenv("WSIS_DW_TAIL_MAIN");
  return e ? atoi(e) != 0 : true;
}
inline bool dw_tail_on_main(const wsis_op* ops, int n, int i) {
  return i == n - 1 && ops[i].kind == WSIS_OP_CONV_BWD && ops[i].out[1] && !ops[i].out[0];
}

bool dw_stream_enabled() {   // read per pass: bench.py switches it off for its event-instrumented roofline steps
  const char* e = getenv("WSIS_DW_STREAM");
  return e ? atoi(e) != 0 : true;
}

inline int64_t wt_bytes_of(const wsis_op& op) {
  return up((int64_t)op.K * op.Cin * op.Cout * (int64_t)sizeof(float));
}

// wave-autonomous kernel (csrc/spconv2.hip) for a product with Cin input and Cout output channels; WSIS_FWD2=0 keeps
// every convolution on spconv_fwd_kernel
bool fwd2_enabled() {
  const char* e = getenv("WSIS_FWD2");
  return e ? atoi(e) != 0 : true;
}
inline bool use_fwd2(const wsis_op& op, bool on) {
  // 32-bit gather offsets: the gathered tensor (input of the forward pass, dY of the dIn pass) stays below 2 GiB
  const int64_t gathered = op.kind == WSIS_OP_CONV ? op.M_in * op.Cin : op.M_out * op.Cout;
  return on && wsis_spconv_fwd_t_supported(op.K, op.Cin, op.Cout) != 0 && gathered * 4 < ((int64_t)1 << 31);
}
// does op need a transposed copy of its weights?  forward on the new kernel (B^T layout); dIn on the old one
inline bool needs_wt(const wsis_op& op, bool on) {
  if (op.kind == WSIS_OP_CONV) return use_fwd2(op, on);
  if (op.kind == WSIS_OP_CONV_BWD) return op.out[0] != nullptr && !use_fwd2(op, on);
  return false;
}

// slab workspace of an op's weight-gradient product (the own-rows form when its input BatchNorm is applied on the fly)
inline int64_t dw_ws_of(const wsis_op& op) {
  if (op.flags & WSIS_OPF_BN_IN) return up(wsis_spconv_dw_bn_workspace_bytes(op.M_in, op.K, op.Cin, op.Cout));
  return up(wsis_spconv_dw_workspace_bytes(op.M_out, op.K, op.Cin, op.Cout));
}

int64_t op_ws_bytes(const wsis_op& op, bool on) {
  switch (op.kind) {
    case WSIS_OP_CONV:
#if WSIS_EXPERIMENTAL
      if (op.flags & (WSIS_OPF_BN_IN | WSIS_OPF_STAT_FIN))
        return up(std::max(wsis_spconv_fwd_f_workspace_bytes(op.M_out, op.K, op.Cin, op.Cout),
                           wsis_spconv_fwd_t_workspace_bytes(op.M_out, op.K, op.Cin, op.Cout)));
#else
      if (op.flags & (WSIS_OPF_BN_IN | WSIS_OPF_STAT_FIN)) return -1;      // EXPERIMENTAL build only
#endif
      if (use_fwd2(op, on)) return up(wsis_spconv_fwd_t_workspace_bytes(op.M_out, op.K, op.Cin, op.Cout));
      return up(wsis_spconv_fwd_workspace_bytes(op.M_out, op.K, op.Cin, op.Cout));
    case WSIS_OP_BN_RELU:
      if (!(op.flags & WSIS_OPF_TRAINING)) return 0;
      return up(std::max(wsis_bn_workspace_bytes(op.M_in, op.Cin),
                         wsis_bn_stats_finalize_workspace_bytes((op.M_in + 31) / 32, op.Cin)));
    case WSIS_OP_BN_RELU_BWD:
      return up(std::max(wsis_bn_workspace_bytes(op.M_in, op.Cin),
                         wsis_bn_stats_finalize_workspace_bytes((op.M_in + 31) / 32, op.Cin)));
    case WSIS_OP_CONV_BWD:     // the transposed weights and the dW slabs live in their own regions, not here
      if (!op.out[0]) return 0;
      if (use_fwd2(op, on)) return up(wsis_spconv_fwd_t_workspace_bytes(op.M_in, op.K, op.Cout, op.Cin));
      return up(wsis_spconv_fwd_workspace_bytes(op.M_in, op.K, op.Cout, op.Cin));
    default:
      return 0;
  }
}


#if WSIS_EXPERIMENTAL      // (make EXPERIMENTAL=1: the retired designs of DESIGN.md section 8)
// ---- resident deep-level kernel (deep.hip): which ops of a pass can run as phases of ONE launch ------------------------
// rows of the tensor an op WRITES (a dIn product writes M_in rows); 0: the op writes nothing on the caller's stream
inline int64_t deep_rows(const wsis_op& op) {
  switch (op.kind) {
    case WSIS_OP_CONV: return op.M_out;
    case WSIS_OP_CONV_BWD: return op.out[0] ? op.M_in : 0;
    default: return op.M_in;
  }
}
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// the BatchNorm backward op that takes the slice partials of dIn op i (see run_ops_impl)
inline int paired_bn(const wsis_op* ops, int n, int i) {
  const wsis_op& op = ops[i];
  for (int j = i + 1; j < n && j <= i + 4; ++j)
    if (ops[j].kind == WSIS_OP_BN_RELU_BWD && ops[j].in[1] == op.out[0] && (ops[j].flags & WSIS_OPF_STATS)) return j;
  return -1;
}

bool deep_op_ok(const wsis_op* ops, int n, int i, bool fwd2_on) {
  const wsis_op& op = ops[i];
  const int64_t R = deep_max_rows();
  int nw = 0, zs = 0;
  switch (op.kind) {
    case WSIS_OP_CONV:
      if (op.flags & (WSIS_OPF_BN_IN | WSIS_OPF_STAT_FIN)) return false;
      if (!use_fwd2(op, fwd2_on) || op.M_out < 1 || op.M_out > R) return false;
      if ((int64_t)op.K * op.M_out * 4 >= ((int64_t)1 << 31)) return false;
      return deep_conv_plan(op.M_out, op.K, op.Cin, op.Cout, &nw, &zs) && aligned16(op.in[0]) && aligned16(op.out[0]);
    case WSIS_OP_CONV_BWD: {
      if (op.flags & WSIS_OPF_BN_IN) return false;
      if (!op.out[0]) return op.M_in <= R;            // weight gradient only: nothing on the caller's stream
      if (!use_fwd2(op, fwd2_on) || op.M_in < 1 || op.M_in > R) return false;
      if (!deep_conv_plan(op.M_in, op.K, op.Cout, op.Cin, &nw, &zs)) return false;
      if (op.flags & WSIS_OPF_STATS) {
        const int j = paired_bn(ops, n, i);
        if (j < 0 || !ops[j].in[7] || ops[j].M_in != op.M_in || ops[j].Cin != op.Cin) return false;
      }
      return aligned16(op.in[2]) && aligned16(op.out[0]);
    }
    case WSIS_OP_BN_RELU:
      if (op.M_in < 1 || op.M_in > R || op.Cin % 32 || !deep_bn_rows_ok(op.M_in)) return false;
      if (op.flags & WSIS_OPF_TRAINING) {
        if (!(op.flags & WSIS_OPF_STATS) || !op.in[5]) return false;
        if (op.in[6] && (op.K % 32 || op.K <= 0 || op.K >= op.Cin)) return false;
      } else if (!op.out[0]) {
        return false;
      }
      return aligned16(op.in[0]) && (!op.out[0] || aligned16(op.out[0]));
    case WSIS_OP_BN_RELU_BWD:
      if (op.M_in < 1 || op.M_in > R || op.Cin % 32 || !deep_bn_rows_ok(op.M_in)) return false;
      if (!(op.flags & WSIS_OPF_TRAINING) || !(op.flags & WSIS_OPF_STATS) || !op.in[7]) return false;
      return aligned16(op.in[0]) && aligned16(op.in[1]) && aligned16(op.out[0]) && (!op.in[6] || aligned16(op.in[6]));
    case WSIS_OP_CAT:
      return op.M_in >= 1 && op.M_in <= R && vec4_ok(op, op.in[0], op.in[1], op.out[0]);
    case WSIS_OP_SPLIT:
      return op.M_in >= 1 && op.M_in <= R && vec4_ok(op, op.in[0], op.out[0], op.out[1]);
    default:
      return false;
  }
}

// end (exclusive) of the run of eligible ops that starts at i0 (i0 itself must be eligible); a dIn product and the
// BatchNorm backward that takes its partials stay together; `stop`: an op index the run must not pass (milestone)
int deep_run_end(const wsis_op* ops, int n, int i0, int stop, bool fwd2_on) {
  int i1 = i0;
  while (i1 < n && (stop < 0 || i1 <= stop) && deep_op_ok(ops, n, i1, fwd2_on)) ++i1;
  // a pair cut by the end of the run: end the run in front of the dIn product
  for (int i = i0; i < i1; ++i)
    if (ops[i].kind == WSIS_OP_CONV_BWD && ops[i].out[0] && (ops[i].flags & WSIS_OPF_STATS)) {
      const int j = paired_bn(ops, n, i);
      if (j >= i1) {
        i1 = i;
        break;
      }
    }
  // a BatchNorm backward whose producer is not in the run cannot take partials either
  for (int j = i0; j < i1; ++j)
    if (ops[j].kind == WSIS_OP_BN_RELU_BWD) {
      bool found = false;
      for (int i = j - 1; i >= i0 && i >= j - 4 && !found; --i)
        found = ops[i].kind == WSIS_OP_CONV_BWD && ops[i].out[0] == ops[j].in[1] && (ops[i].flags & WSIS_OPF_STATS);
      if (!found) {
        i1 = j;
        break;
      }
    }
  return i1;
}

constexpr int kDeepMinOps = 4;          // shorter runs stay ordinary launches
// the phases and stamps of the profiled resident launches since profiling was switched on (tools/deep_phases.py)
std::vector<std::pair<std::vector<DeepOp>, const unsigned long long*>> g_deep_log;
inline int64_t deep_table_bytes(int n_ops) { return up((int64_t)(2 * n_ops + 2) * (int64_t)sizeof(DeepOp)); }

// slab workspace of the phases of ops [i0, i1) (every phase its own region: written once inside the launch)
int64_t deep_slab_bytes(const wsis_op* ops, int i0, int i1) {
  int64_t b = 0;
  for (int i = i0; i < i1; ++i) {
    const wsis_op& op = ops[i];
    int nw = 0, zs = 1;
    if (op.kind == WSIS_OP_CONV && deep_conv_plan(op.M_out, op.K, op.Cin, op.Cout, &nw, &zs) && zs > 1)
      b += up((int64_t)zs * op.M_out * op.Cout * 4);
    if (op.kind == WSIS_OP_CONV_BWD && op.out[0] && deep_conv_plan(op.M_in, op.K, op.Cout, op.Cin, &nw, &zs) && zs > 1)
      b += up((int64_t)zs * op.M_in * op.Cin * 4);
  }
  return b;
}

// workspace of the resident launches of a pass: table + slabs of the largest run (runs execute one after the other)
int64_t deep_ws_bytes(const wsis_op* ops, int n, bool fwd2_on) {
  if (!deep_enabled()) return 0;
  int64_t best = 0;
  for (int i = 0; i < n;) {
    if (!deep_op_ok(ops, n, i, fwd2_on)) {
      ++i;
      continue;
    }
    const int i1 = deep_run_end(ops, n, i, -1, fwd2_on);
    if (i1 - i >= kDeepMinOps) best = std::max(best, deep_table_bytes(i1 - i) + deep_slab_bytes(ops, i, i1));
    i = i1 > i ? i1 : i + 1;
  }
  return best;
}

// phases of ops [i0, i1) -> table; wt_base / wt_off: the transposed weights of the pass; slab: private slab regions
int deep_build(const wsis_op* ops, int n, int i0, int i1, const char* wt_base, const std::vector<int64_t>& wt_off, char* slab,
               bool stamps, std::vector<DeepOp>& tab, std::vector<std::pair<int, std::pair<int, int>>>& conv_stamps) {
  tab.clear();
  conv_stamps.clear();
  auto push = [&](DeepOp& d) {
    d.stamp = stamps ? (int)tab.size() + 1 : -1;
    tab.push_back(d);
  };
  for (int i = i0; i < i1; ++i) {
    const wsis_op& op = ops[i];
    DeepOp d{};
    d.NW = 1;
    d.ZS = 1;
    switch (op.kind) {
      case WSIS_OP_CONV:
      case WSIS_OP_CONV_BWD: {
        const bool fwd = op.kind == WSIS_OP_CONV;
        if (!fwd && !op.out[0]) break;
        const int64_t Mo = fwd ? op.M_out : op.M_in, Mg = fwd ? op.M_in : op.M_out;
        const int Ci = fwd ? op.Cin : op.Cout, Co = fwd ? op.Cout : op.Cin;
        int nw = 1, zs = 1;
        if (!deep_conv_plan(Mo, op.K, Ci, Co, &nw, &zs)) return fail(WSIS_ERR_ARG, "deep: op %d has no plan", i);
        d.kind = DK_CONV;
        d.NW = nw;
        d.ZS = zs;
        d.K = op.K;
        d.Cin = Ci;
        d.Cout = Co;
        d.M_in = Mg;
        d.M_out = Mo;
        d.x_bytes = (uint32_t)(Mg * Ci * 4);
        const void* stats = nullptr;
        DeepOp epi{};
        if (fwd) {
          d.p[0] = op.in[0];
          d.p[1] = op.in[1];
          d.p[2] = op.in[2];
          d.p[3] = wt_base + wt_off[i];
          d.p[4] = op.in[4];
          d.p[5] = op.in[5];
          d.p[6] = op.out[0];
          d.flip = 0;
          if (op.flags & WSIS_OPF_STATS) stats = op.out[1];
        } else {
          d.p[0] = op.in[2];
          d.p[1] = op.in[5];
          d.p[2] = op.in[6];
          d.p[3] = op.in[1];
          d.p[6] = op.out[0];
          d.flip = (op.flags & WSIS_OPF_FLIP) ? 1 : 0;
          if (op.flags & WSIS_OPF_STATS) {
            const wsis_op& bn = ops[paired_bn(ops, n, i)];
            stats = bn.in[7];
            epi.p[9] = bn.in[0];
            epi.p[10] = bn.in[2];
            epi.p[11] = bn.in[3];
            epi.p[12] = bn.in[4];
            epi.p[13] = bn.in[5];
            epi.eps = bn.eps;
            epi.relu = (bn.flags & WSIS_OPF_RELU) ? 1 : 0;
          }
        }
        WSIS_REQUIRE(d.p[1] || (op.K == 1 && Mg == Mo), "nbr may be null only for the dense 1x1 case");
        const int s0 = (int)tab.size();
        if (zs > 1) {
          DeepOp r = d;              // the slab sum finishes the product: bias / residual / partials move there
          d.p[7] = slab;
          d.p[4] = d.p[5] = nullptr;
          push(d);
          r.kind = DK_REDUCE;
          r.p[7] = slab;
          r.p[8] = stats;
          for (int k = 9; k <= 13; ++k) r.p[k] = epi.p[k];
          r.eps = epi.eps;
          r.relu = epi.relu;
          push(r);
          slab += up((int64_t)zs * Mo * Co * 4);
        } else {
          d.p[8] = stats;
          for (int k = 9; k <= 13; ++k) d.p[k] = epi.p[k];
          d.eps = epi.eps;
          d.relu = epi.relu;
          push(d);
        }
        conv_stamps.push_back({i, {s0, (int)tab.size()}});      // stamps s0 (start) .. s0 + 1 (main) .. size (product done)
        break;
      }
      case WSIS_OP_BN_RELU: {
        const bool training = (op.flags & WSIS_OPF_TRAINING) != 0;
        const bool upd = (op.flags & WSIS_OPF_UPDATE_RUNNING) != 0;
        d.kind = DK_BN_FWD;
        d.training = training ? 1 : 0;
        d.relu = (op.flags & WSIS_OPF_RELU) ? 1 : 0;
        d.Cin = op.Cin;
        d.M_in = op.M_in;
        d.eps = op.eps;
        d.momentum = op.momentum;
        d.C0 = op.in[6] ? op.K : op.Cin;
        d.p[0] = op.in[0];
        d.p[1] = op.in[1];
        d.p[2] = op.in[2];
        d.p[3] = (training && !upd) ? nullptr : op.in[3];
        d.p[4] = (training && !upd) ? nullptr : op.in[4];
        d.p[5] = op.in[5];
        d.p[6] = op.in[6];
        d.p[7] = op.out[0];
        d.p[8] = op.out[1];
        d.p[9] = op.out[2];
        push(d);
        break;
      }
      case WSIS_OP_BN_RELU_BWD:
        d.kind = DK_BN_BWD;
        d.relu = (op.flags & WSIS_OPF_RELU) ? 1 : 0;
        d.Cin = op.Cin;
        d.M_in = op.M_in;
        d.eps = op.eps;
        d.p[0] = op.in[0];
        d.p[1] = op.in[1];
        d.p[2] = op.in[2];
        d.p[3] = op.in[3];
        d.p[4] = op.in[4];
        d.p[5] = op.in[5];
        d.p[6] = op.in[6];
        d.p[7] = op.in[7];
        d.p[8] = op.out[0];
        d.p[9] = op.out[1];
        d.p[10] = op.out[2];
        push(d);
        break;
      case WSIS_OP_CAT:
        d.kind = DK_CAT;
        d.Cin = op.Cin;
        d.Cout = op.Cout;
        d.M_in = op.M_in;
        d.p[0] = op.in[0];
        d.p[1] = op.in[1];
        d.p[2] = op.out[0];
        push(d);
        break;
      case WSIS_OP_SPLIT:
        d.kind = DK_SPLIT;
        d.Cin = op.Cin;
        d.Cout = op.Cout;
        d.M_in = op.M_in;
        d.p[0] = op.in[0];
        d.p[1] = op.out[0];
        d.p[2] = op.out[1];
        push(d);
        break;
      default:
        return fail(WSIS_ERR_ARG, "deep: op %d of kind %d is not a phase", i, op.kind);
    }
  }
  return WSIS_OK;
}

#else
inline int64_t deep_ws_bytes(const wsis_op*, int, bool) { return 0; }
#endif

}  // namespace

__global__ void warm_stream_kernel(int* p) {
  if (p && threadIdx.x == 0xffff) p[0] = 0;
}

extern "C" {

// Binds the library's weight-gradient side stream of `stream` to its hardware queue NOW (a stream takes its queue with
// its first command).  A process that is about to create an RCCL communicator calls it first, after warming its own
// streams the same way: HIP maps streams onto GPU_MAX_HW_QUEUES (4) hardware queues in the order of their first use,
// and with the communicator's streams in between two streams of a step end up sharing one (one-rank RCCL line 9.5 ms
// per step against 7.9 without a group, 8.9 with GPU_MAX_HW_QUEUES=6: tools/rccl_ab.sh).
int wsis_warm_streams(void* stream) {
  hipStream_t st = as_stream(stream);
  SideStream* side = side_stream_for(st);
  WSIS_REQUIRE(side != nullptr, "side stream creation failed");
  hipLaunchKernelGGL(warm_stream_kernel, dim3(1), dim3(64), 0, st, (int*)nullptr);
  hipLaunchKernelGGL(warm_stream_kernel, dim3(1), dim3(64), 0, side->stream, (int*)nullptr);
  WSIS_LAUNCH_CHECK();
  WSIS_HIP_CHECK(hipStreamSynchronize(side->stream));
  return WSIS_OK;
}

int64_t wsis_run_ops_workspace_bytes(const wsis_op* ops, int32_t n) {
  if (!ops || n < 0) return -1;
  int64_t need = ALIGN, wt = 0, dw = 0, dw_sum = 0;
  const bool on = fwd2_enabled();
  for (int i = 0; i < n; ++i) {
    int64_t b = op_ws_bytes(ops[i], on);
    if (b < 0) return -1;
    if (dw_tail_on_main(ops, n, i)) b = std::max(b, dw_ws_of(ops[i]));      // (its slabs live in the op workspace)
    if (b > need) need = b;
    if (needs_wt(ops[i], on)) wt += wt_bytes_of(ops[i]);
    if (ops[i].kind == WSIS_OP_CONV_BWD && ops[i].out[1]) {
      const int64_t d = dw_ws_of(ops[i]);
      if (d < 0) return -1;
      if (d > dw) dw = d;
      dw_sum += d;
    }
  }
  // (deferred slab sums: every product keeps its slabs until the batched launch; the switch is read per pass -- a pass
  // whose workspace was sized with it off and that runs with it on fails its own size check)
  const char* be = getenv("WSIS_DW_BATCH_REDUCE");
  const bool batch = be && atoi(be) != 0;
  return wt + (batch ? std::max(dw, dw_sum) : dw) + need + deep_ws_bytes(ops, n, on) + ALIGN;
}

int wsis_run_ops(const wsis_op* ops, int32_t n, void* d_ws, int64_t ws_bytes, void* d_sync, void* stream) {
  return wsis_run_ops_marked(ops, n, d_ws, ws_bytes, d_sync, stream, -1, nullptr);
}

static int run_ops_impl(const wsis_op* ops, int32_t n, void* d_ws, int64_t ws_bytes, void* d_sync, void* stream,
                        int32_t mark_op, void* waiter_stream, bool defer_join = false);

// A pass issued in PARTS (the host does something between two parts: the statistics exchange of a SyncBatchNorm layer,
// model/unet_native.py): parts with last == 0 leave the weight-gradient side stream un-joined, the part with last != 0
// (n may be 0) joins everything forked since.  Every part needs its OWN workspace, alive until the last part returns.
int wsis_run_ops_part(const wsis_op* ops, int32_t n, void* d_ws, int64_t ws_bytes, void* d_sync, void* stream, int32_t last) {
  return run_ops_impl(ops, n, d_ws, ws_bytes, d_sync, stream, -1, nullptr, last == 0);
}

// WSIS_GRAPH=N (EXPERIMENTAL build, default 0): the launches of a pass are recorded into HIP graphs of ~N ops each
// (N < 4: one graph per pass; the weight-gradient side stream joins the capture through its fork / join events) and
// replayed with one hipGraphLaunch per chunk; the executable graph of a chunk is kept and patched
// (hipGraphExecUpdate) when the next scene's pass has the same kernel sequence.  Measured on the C2 step: results
// identical, replay removes about 1 ms of dispatch gaps from the forward pass, but recording + patching costs the
// host about 1 ms more than launching, and the GPU cannot start a chunk before its recording ends: 12.0 -> 13.0 ms
// per step.  The way to make it pay is to patch node parameters without re-recording.
int wsis_run_ops_marked(const wsis_op* ops, int32_t n, void* d_ws, int64_t ws_bytes, void* d_sync, void* stream,
                        int32_t mark_op, void* waiter_stream) {
#if !WSIS_EXPERIMENTAL
  return run_ops_impl(ops, n, d_ws, ws_bytes, d_sync, stream, mark_op, waiter_stream);
#else
  const char* ge = getenv("WSIS_GRAPH");        // read per pass (a test switches it)
  const int graph_mode = ge ? atoi(ge) : 0;
  if (!graph_mode || mark_op >= 0 || g_prof_on || n == 0)
    return run_ops_impl(ops, n, d_ws, ws_bytes, d_sync, stream, mark_op, waiter_stream);
  hipStream_t st = as_stream(stream);
  // chunks of ~graph_mode ops: the host records chunk k + 1 while chunk k runs (one graph for the whole pass would keep
  // the GPU idle for the whole recording time).  A chunk never separates a dIn pass from the BatchNorm backward that
  // takes its epilogue partials.
  static std::map<std::pair<hipStream_t, int64_t>, hipGraphExec_t> cache;      // (stream, pass signature, chunk)
  static hipStream_t cap = nullptr;
  if (!cap) WSIS_HIP_CHECK(hipStreamCreateWithFlags(&cap, hipStreamNonBlocking));
  const int target = graph_mode >= 4 ? graph_mode : 1 << 30;
  int chunk_no = 0;
  for (int i0 = 0; i0 < n; ++chunk_no) {
    int i1 = i0, pending = 0;
    while (i1 < n) {
      const wsis_op& op = ops[i1];
      if (op.kind == WSIS_OP_CONV_BWD && (op.flags & WSIS_OPF_STATS) && op.out[0]) ++pending;
      if (op.kind == WSIS_OP_BN_RELU_BWD && (op.flags & WSIS_OPF_STATS) && pending > 0) --pending;
      ++i1;
      if (i1 - i0 >= target && pending == 0) break;
    }
    WSIS_HIP_CHECK(hipStreamBeginCapture(cap, hipStreamCaptureModeRelaxed));
    // (the sync slots follow the op's index in its chunk: chunks run one after the other on the caller's stream)
    const int rc = run_ops_impl(ops + i0, i1 - i0, d_ws, ws_bytes, d_sync, cap, -1, nullptr);
    hipGraph_t g = nullptr;
    const hipError_t ec = hipStreamEndCapture(cap, &g);
    if (ec != hipSuccess || !g) return fail(WSIS_ERR_HIP, "graph capture failed: %s", hipGetErrorString(ec));
    if (rc != WSIS_OK) {
      (void)hipGraphDestroy(g);
      return rc;
    }
    const auto key = std::make_pair(st, ((int64_t)ops[0].kind * 4096 + n) * 64 + chunk_no);
    hipGraphExec_t& ex = cache[key];
    bool ready = false;
    if (ex) {
      hipGraphNode_t bad = nullptr;
      hipGraphExecUpdateResult res;
      if (hipGraphExecUpdate(ex, g, &bad, &res) == hipSuccess && res == hipGraphExecUpdateSuccess) {
        ready = true;
      } else {
        (void)hipGetLastError();
        (void)hipGraphExecDestroy(ex);
        ex = nullptr;
      }
    }
    if (!ready) {
      const hipError_t ei = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
      if (ei != hipSuccess) {
        (void)hipGraphDestroy(g);
        ex = nullptr;
        return fail(WSIS_ERR_HIP, "graph instantiate failed: %s", hipGetErrorString(ei));
      }
    }
    const hipError_t el = hipGraphLaunch(ex, st);
    (void)hipGraphDestroy(g);
    if (el != hipSuccess) return fail(WSIS_ERR_HIP, "graph launch failed: %s", hipGetErrorString(el));
    i0 = i1;
  }
  return WSIS_OK;
#endif
}

static int run_ops_impl(const wsis_op* ops, int32_t n, void* d_ws, int64_t ws_bytes, void* d_sync, void* stream,
                        int32_t mark_op, void* waiter_stream, bool defer_join) {
  WSIS_REQUIRE((ops || n == 0) && n >= 0, "bad op list");
  WSIS_REQUIRE(mark_op < n && (mark_op < 0 || waiter_stream), "bad milestone");
  WSIS_REQUIRE((n == 0 || ws_bytes >= wsis_run_ops_workspace_bytes(ops, n)) && (d_ws || n == 0), "workspace too small");
  hipStream_t st = as_stream(stream);
  char* ws = static_cast<char*>(d_ws);
  // ---- all transposed weights of the pass, WT_MAX layers per launch; wt_off[i] = offset of op i's W^T
  const bool on = fwd2_enabled();
  std::vector<int64_t> wt_off(n, -1);
  int64_t wt_total = 0;
  // The transposes (one launch, 28 us at the top of a forward pass) depend on the weights alone: they go to the library's
  // side stream behind an event of the caller's stream, and the caller's stream waits for them in front of the first op
  // that reads a transposed weight -- the 6-channel input convolution and its BatchNorm run meanwhile.
  // WSIS_WT_SIDE=0 (read per pass): on the caller's stream as before.
  SideStream* wt_side = nullptr;
  bool wt_pending = false;
  {
    const char* we = getenv("WSIS_WT_SIDE");
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    const bool cap = hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone;
    bool any_wt = false, first_needs = n > 0 && needs_wt(ops[0], on);
    for (int i = 0; i < n && !any_wt; ++i) any_wt = needs_wt(ops[i], on);
    // (not with the resident deep-level launches of the EXPERIMENTAL build: a run of ops is issued as one launch there)
    if ((we ? atoi(we) != 0 : true) && dw_stream_enabled() && !cap && any_wt && !first_needs && !g_prof_on &&
        deep_ws_bytes(ops, n, on) == 0)
      wt_side = side_stream_for(st);
    if (wt_side) {
      hipError_t e = hipEventRecord(wt_side->wt_fork, st);
      if (e == hipSuccess) e = hipStreamWaitEvent(wt_side->stream, wt_side->wt_fork, 0);
      if (e != hipSuccess) return fail(WSIS_ERR_HIP, "weight-transpose fork failed: %s", hipGetErrorString(e));
    }
  }
  hipStream_t wt_st = wt_side ? wt_side->stream : st;
  {
    WtBatch b;
    b.n = 0;
    b.start[0] = 0;
    auto flush = [&]() -> int {
      if (b.n == 0) return WSIS_OK;
      hipLaunchKernelGGL(weight_transpose_batch_kernel, dim3(b.start[b.n]), dim3(256), 0, wt_st, b);
      WSIS_LAUNCH_CHECK();
      b.n = 0;
      return WSIS_OK;
    };
    for (int i = 0; i < n; ++i) {
      const wsis_op& op = ops[i];
      if (!needs_wt(op, on)) continue;
      const bool fwd = op.kind == WSIS_OP_CONV;
      const void* wsrc = fwd ? op.in[3] : op.in[1];
      WSIS_REQUIRE(wsrc, "convolution without weights");
      wt_off[i] = wt_total;
      b.src[b.n] = (const float*)wsrc;
      b.dst[b.n] = reinterpret_cast<float*>(ws + wt_total);
      b.K[b.n] = op.K;
      b.Cin[b.n] = op.Cin;
      b.Cout[b.n] = op.Cout;
      b.flip[b.n] = (!fwd && (op.flags & WSIS_OPF_FLIP)) ? 1 : 0;
      b.start[b.n + 1] = b.start[b.n] + op.K * ((op.Cin + WT_TILE - 1) / WT_TILE) * ((op.Cout + WT_TILE - 1) / WT_TILE);
      ++b.n;
      wt_total += wt_bytes_of(op);
      if (b.n == WT_MAX) {
        const int rc = flush();
        if (rc != WSIS_OK) return rc;
      }
    }
    const int rc = flush();
    if (rc != WSIS_OK) return rc;
    if (wt_side) {
      const hipError_t e = hipEventRecord(wt_side->wt_done, wt_side->stream);
      if (e != hipSuccess) return fail(WSIS_ERR_HIP, "weight-transpose event failed: %s", hipGetErrorString(e));
      wt_pending = true;
    }
  }
  ws += wt_total;
  ws_bytes -= wt_total;
  char* const wt_base = static_cast<char*>(d_ws);
  // dW region (shared by the dW launches, which are ordered among themselves on one stream)
  // ... or, with deferred slab sums, one region per product: its slabs live until the batched launch that sums them
  const bool dw_defer = dw_batch_reduce_enabled();
  const bool tail_main = dw_tail_main_enabled();
  // (measured in-process: 7.69 -> 7.50 ms at one scene per step, 11.70 -> 11.50 at two, 20.11 -> 20.22 at four -- with a
  // batch that fills the GPU the early product takes from the dIn launch beside it what it gains at the tail; default:
  // by the batch hint, below WSIS_DW_EARLY_ROWS active voxels)
  const char* early_env = getenv("WSIS_DW_EARLY");
  const char* early_rows_env = getenv("WSIS_DW_EARLY_ROWS");
  const int64_t early_rows = early_rows_env ? atoll(early_rows_env) : 500000;
  const bool dw_early = early_env ? atoi(early_env) != 0 : dw2_batch_rows() < early_rows;
  int64_t dw_bytes = 0, dw_sum = 0;
  int n_dw = 0;
  for (int i = 0; i < n; ++i)
    if (ops[i].kind == WSIS_OP_CONV_BWD && ops[i].out[1]) {
      const int64_t d = dw_ws_of(ops[i]);
      if (d > dw_bytes) dw_bytes = d;
      dw_sum += d;
      ++n_dw;
    }
  char* const dw_ws = ws;
  ws += dw_defer ? dw_sum : dw_bytes;
  ws_bytes -= dw_defer ? dw_sum : dw_bytes;
  if (ws_bytes < 0) return fail(WSIS_ERR_ARG, "wsis_run_ops: workspace too small for the weight-gradient slabs");
  // record slots of the pass's deferred products (stable addresses: the worker thread fills them) and what has been
  // flushed so far; the vector outlives every task that points into it (drained before this function returns)
  std::vector<DwRedRec> dw_recs((size_t)(dw_defer ? n_dw : 0));
  int dw_next = 0, dw_flushed = 0;
  int64_t dw_off = 0;
  // resident deep-level launches (deep.hip): phase table + private slab regions of the largest run
  const int64_t deep_bytes = deep_ws_bytes(ops, n, on);
  char* const deep_ws = ws;
  (void)deep_ws;
  ws += deep_bytes;
  ws_bytes -= deep_bytes;
  SideStream* side = (dw_bytes > 0 && dw_stream_enabled()) ? side_stream_for(st) : nullptr;
  if (side) side->next = 0;
  // the side-stream calls of the pass from a second host thread (not under stream capture, not while the profiler's
  // event lists are being filled: those are single-threaded)
  hipStreamCaptureStatus cap_state = hipStreamCaptureStatusNone;
  const bool capturing = hipStreamIsCapturing(st, &cap_state) != hipSuccess || cap_state != hipStreamCaptureStatusNone;
  const bool use_worker = side != nullptr && dw_thread_enabled() && !g_prof_on && !capturing;
  int cur_dev = 0;
  if (use_worker && hipGetDevice(&cur_dev) != hipSuccess) return fail(WSIS_ERR_HIP, "hipGetDevice failed");
  bool worker_busy = false;
  auto drain_worker = [&]() -> int {      // every dW task pushed so far has been issued (and did it fail?)
    if (!worker_busy) return WSIS_OK;
    worker_busy = false;
    std::string msg;
    const int wrc = dw_worker().drain(&msg);
    return wrc == WSIS_OK ? WSIS_OK : fail(wrc, "%s", msg.c_str());
  };
  // inside the op loop nothing returns directly: an error after the fork still has to reach the join below
#define RUN_LAUNCH_CHECK()                                                                              \
  do {                                                                                                  \
    const hipError_t le_ = hipGetLastError();                                                           \
    if (le_ != hipSuccess) rc = fail(WSIS_ERR_HIP, "wsis_run_ops: launch failed: %s", hipGetErrorString(le_)); \
  } while (0)
  bool forked = false;
  int first_err = WSIS_OK;
  // weight gradient of a CONV_BWD op: forked to the side stream behind everything enqueued so far on the caller's stream
  auto dw_issue = [&](const wsis_op& op) -> int {
    void* dw_stream = stream;
    char* my_ws = dw_ws;
    int64_t my_bytes = dw_bytes;
    DwRedRec* rec = nullptr;
    if (dw_defer && dw_next < n_dw) {
      my_bytes = dw_ws_of(op);
      my_ws = dw_ws + dw_off;
      dw_off += my_bytes;
      rec = &dw_recs[(size_t)dw_next++];
    }
    if (side) {   // dY is complete once everything enqueued so far on the caller's stream has run
      hipEvent_t e = next_fork_event(side);
      hipError_t he = e ? hipEventRecord(e, st) : hipErrorOutOfMemory;
      if (he == hipSuccess && !use_worker) he = hipStreamWaitEvent(side->stream, e, 0);
      if (he != hipSuccess) return fail(WSIS_ERR_HIP, "dW side-stream fork failed: %s", hipGetErrorString(he));
      dw_stream = side->stream;
      forked = true;
      if (use_worker) {     // the wait and the launches come from the worker thread, in push order
        DwTask t;
        t.op = op;
        t.ev = e;
        t.dev = cur_dev;
        t.side = side->stream;
        t.ws = my_ws;
        t.ws_bytes = my_bytes;
        t.rec = rec;
        dw_worker().push(t);
        worker_busy = true;
        return WSIS_OK;
      }
    }
    return issue_dw(op, my_ws, my_bytes, dw_stream, rec);
  };
  // ONE launch finishes the deferred products issued since the last flush (behind them on their stream)
  auto dw_flush = [&]() -> int {
    if (!dw_defer || dw_flushed == dw_next) return WSIS_OK;
    const DwRedRec* recs = dw_recs.data() + dw_flushed;
    const int cnt = dw_next - dw_flushed;
    dw_flushed = dw_next;
    if (side && use_worker) {
      DwTask t;
      t.ev = nullptr;
      t.dev = cur_dev;
      t.side = side->stream;
      t.ws = nullptr;
      t.ws_bytes = 0;
      t.flush = recs;
      t.flush_n = cnt;
      dw_worker().push(t);
      worker_busy = true;
      return WSIS_OK;
    }
    return dw2_reduce_batch(recs, cnt, side ? side->stream : st);
  };
#if WSIS_EXPERIMENTAL
  const bool deep_on = deep_bytes > 0 && d_sync != nullptr && !capturing;
  std::vector<DeepOp> deep_tab;
  std::vector<std::pair<int, std::pair<int, int>>> deep_convs;
#endif
  const char* slab_bn_env = getenv("WSIS_SLAB_BN_PARTIALS");      // one-shot path: slab-split dIn products write the
  const bool slab_bn_partials = slab_bn_env && atoi(slab_bn_env) != 0;   // BatchNorm partials too (the resident kernel's form)
  std::vector<char> bn_unfused(n, 0);     // BatchNorm backward ops whose dIn pass did not write partials this run
  for (int i = 0; i < n; ++i) {
    int rc = WSIS_OK;
    if (wt_pending && wt_off[i] >= 0) {      // the first reader of a transposed weight
      wt_pending = false;
      const hipError_t e = hipStreamWaitEvent(st, wt_side->wt_done, 0);
      if (e != hipSuccess) {
        first_err = fail(WSIS_ERR_HIP, "weight-transpose join failed: %s", hipGetErrorString(e));
        break;
      }
    }
#if WSIS_EXPERIMENTAL
    if (deep_on && deep_op_ok(ops, n, i, on)) {
      const int i1 = deep_run_end(ops, n, i, mark_op >= i ? mark_op : -1, on);
      if (i1 - i >= kDeepMinOps) {
        // ---- ops [i, i1) as ONE resident launch; their weight gradients follow on the side stream
        char* slab = deep_ws + deep_table_bytes(i1 - i);
        rc = deep_build(ops, n, i, i1, wt_base, wt_off, slab, g_prof_on, deep_tab, deep_convs);
        unsigned long long* d_stamps = nullptr;
        if (rc == WSIS_OK && g_prof_on && !deep_tab.empty()) {
          if (hipMalloc((void**)&d_stamps, (deep_tab.size() + 1) * sizeof(unsigned long long)) != hipSuccess)
            rc = fail(WSIS_ERR_HIP, "deep: stamp buffer allocation failed");
          else
            g_prof_bufs.push_back(d_stamps);
        }
        if (rc == WSIS_OK)
          rc = deep_launch(deep_tab.data(), (int)deep_tab.size(), deep_ws, deep_sync_slot(d_sync), d_stamps, st);
        if (rc == WSIS_OK && d_stamps) {
          if (g_deep_log.size() > 64) g_deep_log.clear();
          g_deep_log.push_back({deep_tab, d_stamps});
        }
        if (rc == WSIS_OK && d_stamps)
          for (const auto& cs : deep_convs) {
            ProfRec r{};
            r.d_stamps = d_stamps;
            r.s0 = cs.second.first;
            r.sm = cs.second.first + 1;
            r.s1 = cs.second.second;
            g_prof[0].push_back(r);
          }
        for (int k = i; k < i1 && rc == WSIS_OK; ++k)
          if (ops[k].kind == WSIS_OP_CONV_BWD && ops[k].out[1]) rc = dw_issue(ops[k]);
        if (rc != WSIS_OK) {
          first_err = rc;
          break;
        }
        i = i1 - 1;
        goto op_done;
      }
    }
#endif
    {
    const wsis_op& op = ops[i];
    // bench.py's BatchNorm line (wsis_prof_records(2, ...)): one event pair around ALL launches of a BatchNorm op
    const bool is_bn = op.kind == WSIS_OP_BN_RELU || op.kind == WSIS_OP_BN_RELU_BWD;
    std::optional<ProfScope> bn_prof;
    if (is_bn && g_prof_on) bn_prof.emplace(2, st);
    switch (op.kind) {
      case WSIS_OP_CONV:
        if (op.flags & (WSIS_OPF_BN_IN | WSIS_OPF_STAT_FIN)) {
#if !WSIS_EXPERIMENTAL
          rc = fail(WSIS_ERR_ARG, "op %d: the fused BatchNorm forms of the convolution are in the EXPERIMENTAL build only", i);
          break;
#else
          if (wt_off[i] < 0) {
            rc = fail(WSIS_ERR_ARG, "op %d: fused BatchNorm requested from a convolution that is not on wsis_spconv_fwd_t", i);
            break;
          }
          wsis_bn_in bi;
          bi.mean = (const float*)op.in[6];
          bi.var = (const float*)op.in[7];
          bi.gamma = (const float*)op.in[8];
          bi.beta = (const float*)op.in[9];
          bi.eps = op.eps;
          bi.relu = (op.flags & WSIS_OPF_RELU) ? 1 : 0;
          wsis_stat_target tg[2];
          int nt = 0;
          if (op.flags & WSIS_OPF_STAT_FIN) {
            tg[0].mean = (float*)op.out[2];
            tg[0].var = (float*)op.out[3];
            tg[0].running_mean = (float*)const_cast<void*>(op.in[10]);
            tg[0].running_var = (float*)const_cast<void*>(op.in[11]);
            tg[0].momentum = op.momentum;
            tg[0].reserved = 0;
            nt = 1;
            if (op.out[4]) {
              tg[1].mean = (float*)op.out[4];
              tg[1].var = (float*)op.out[5];
              tg[1].running_mean = (float*)op.out[6];
              tg[1].running_var = (float*)op.out[7];
              tg[1].momentum = op.momentum2;
              tg[1].reserved = 0;
              nt = 2;
            }
          }
          rc = wsis_spconv_fwd_f((const float*)op.in[0], (op.flags & WSIS_OPF_BN_IN) ? &bi : nullptr, (const int32_t*)op.in[1],
                                 (const int32_t*)op.in[2], reinterpret_cast<const float*>(wt_base + wt_off[i]), 0,
                                 (const float*)op.in[4], (const float*)op.in[5], (float*)op.out[0],
                                 (op.flags & WSIS_OPF_STATS) ? (float*)op.out[1] : nullptr, tg, nt, op.M_in, op.M_out, op.K,
                                 op.Cin, op.Cout, ws, ws_bytes, sync_slot(d_sync, i), stream);
          break;
#endif
        }
        if (wt_off[i] >= 0)
          rc = wsis_spconv_fwd_t((const float*)op.in[0], (const int32_t*)op.in[1], (const int32_t*)op.in[2],
                                 reinterpret_cast<const float*>(wt_base + wt_off[i]), 0, (const float*)op.in[4],
                                 (const float*)op.in[5], (float*)op.out[0],
                                 (op.flags & WSIS_OPF_STATS) ? (float*)op.out[1] : nullptr, op.M_in, op.M_out, op.K,
                                 op.Cin, op.Cout, ws, ws_bytes, sync_slot(d_sync, i), stream);
        else if (op.flags & WSIS_OPF_STATS)
          // the program was recorded for the kernel that writes BatchNorm partials; this launch cannot use it (input
          // of 2 GiB or more, or WSIS_FWD2 changed between recording and running)
          rc = fail(WSIS_ERR_ARG, "op %d: statistics requested from a convolution that is not on wsis_spconv_fwd_t", i);
        else
          rc = wsis_spconv_fwd((const float*)op.in[0], (const int32_t*)op.in[1], (const int32_t*)op.in[2],
                               (const float*)op.in[3], (const float*)op.in[4], (const float*)op.in[5],
                               (float*)op.out[0], op.M_in, op.M_out, op.K, op.Cin, op.Cout, ws, ws_bytes, stream);
        break;
      case WSIS_OP_BN_RELU: {
        const float* mean = (const float*)op.out[1];
        const float* var = (const float*)op.out[2];
        if (op.flags & WSIS_OPF_TRAINING) {
          const bool upd = (op.flags & WSIS_OPF_UPDATE_RUNNING) != 0;
          float* rm = upd ? (float*)op.in[3] : nullptr;
          float* rv = upd ? (float*)op.in[4] : nullptr;
          if ((op.flags & WSIS_OPF_STATS) && op.M_in > 0) {
            // the producers' epilogues left (sum, sum of squares) partials per 32-row slice: no pass over x
            const int64_t n_part = (op.M_in + 31) / 32;
            const int C0 = op.in[6] ? op.K : op.Cin;
            if (!op.in[6] && op.out[0]) {     // one producer: statistics finish and apply pass in one launch (falls back by itself)
              rc = wsis_bn_stats_finalize_apply((const float*)op.in[5], n_part, op.M_in, op.Cin, (float*)op.out[1],
                                                (float*)op.out[2], rm, rv, op.momentum, (const float*)op.in[0],
                                                (const float*)op.in[1], (const float*)op.in[2], op.eps,
                                                (op.flags & WSIS_OPF_RELU) ? 1 : 0, (float*)op.out[0], ws, ws_bytes,
                                                sync_slot(d_sync, i), stream);
              break;
            }
            rc = wsis_bn_stats_finalize((const float*)op.in[5], n_part, op.M_in, C0, (float*)op.out[1],
                                        (float*)op.out[2], rm, rv, op.momentum, ws, ws_bytes, sync_slot(d_sync, i), stream);
            if (rc == WSIS_OK && op.in[6])
              rc = wsis_bn_stats_finalize((const float*)op.in[6], n_part, op.M_in, op.Cin - C0, (float*)op.out[1] + C0,
                                          (float*)op.out[2] + C0, rm ? rm + C0 : nullptr, rv ? rv + C0 : nullptr,
                                          op.momentum, ws, ws_bytes, sync_slot(d_sync, i), stream);
          } else {
            rc = wsis_bn_stats((const float*)op.in[0], op.M_in, op.Cin, (float*)op.out[1], (float*)op.out[2], rm, rv,
                               op.momentum, ws, ws_bytes, stream);
          }
          if (rc != WSIS_OK) break;
        } else {
          mean = (const float*)op.in[3];
          var = (const float*)op.in[4];
        }
        if (!op.out[0]) break;      // statistics only: the consuming convolution applies the BatchNorm as it reads
        rc = wsis_bn_apply((const float*)op.in[0], mean, var, (const float*)op.in[1], (const float*)op.in[2], op.eps,
                           (op.flags & WSIS_OPF_RELU) ? 1 : 0, (float*)op.out[0], op.M_in, op.Cin, stream);
        break;
      }
      case WSIS_OP_CAT:
        if (op.M_in > 0) {
          if (vec4_ok(op, op.in[0], op.in[1], op.out[0]))
            hipLaunchKernelGGL(cat_rows_kernel<4>, dim3(grid_for(op.M_in * (op.Cin + op.Cout) / 4, 256)), dim3(256), 0,
                               st, (const float*)op.in[0], (const float*)op.in[1], (float*)op.out[0], op.M_in, op.Cin,
                               op.Cout);
          else
            hipLaunchKernelGGL(cat_rows_kernel<1>, dim3(grid_for(op.M_in * (op.Cin + op.Cout), 256)), dim3(256), 0, st,
                               (const float*)op.in[0], (const float*)op.in[1], (float*)op.out[0], op.M_in, op.Cin,
                               op.Cout);
          RUN_LAUNCH_CHECK();
        }
        break;
      case WSIS_OP_SPLIT:
        if (op.M_in > 0) {
          if (vec4_ok(op, op.in[0], op.out[0], op.out[1]))
            hipLaunchKernelGGL(split_rows_kernel<4>, dim3(grid_for(op.M_in * (op.Cin + op.Cout) / 4, 256)), dim3(256),
                               0, st, (const float*)op.in[0], (float*)op.out[0], (float*)op.out[1], op.M_in, op.Cin,
                               op.Cout);
          else
            hipLaunchKernelGGL(split_rows_kernel<1>, dim3(grid_for(op.M_in * (op.Cin + op.Cout), 256)), dim3(256), 0, st,
                               (const float*)op.in[0], (float*)op.out[0], (float*)op.out[1], op.M_in, op.Cin, op.Cout);
          RUN_LAUNCH_CHECK();
        }
        break;
      case WSIS_OP_ADD:
        if (op.M_in * op.Cin > 0) {
          hipLaunchKernelGGL(add_inplace_kernel, dim3(grid_for(op.M_in * op.Cin, 256)), dim3(256), 0, st,
                             (float*)op.out[0], (const float*)op.in[0], op.M_in * op.Cin);
          RUN_LAUNCH_CHECK();
        }
        break;
      case WSIS_OP_CONV_BWD: {
        // in: X, W, dY, nbr_f, order_f, nbr_b, order_b ; out: dX (optional), dW (optional)
        char* rest = ws;
        const int64_t rest_bytes = ws_bytes;
        // The weight gradient needs X (saved by the forward pass) and dY -- complete once everything enqueued so far has
        // run -- but NOT this op's own dIn product: forked in front of it (round 6), it starts beside the dIn launch that
        // reads the same dY instead of behind it, and the side stream runs one product further ahead all the way to the
        // tail of the pass.  WSIS_DW_EARLY=0 (read per pass): forked behind the dIn launch as before.
        bool dw_forked = false;
        if (dw_early && op.out[1] && op.out[0] && side) {
          rc = dw_issue(op);
          dw_forked = true;
          if (rc != WSIS_OK) break;
        }
        if (op.out[0]) {
          if (wt_off[i] >= 0 && (op.flags & WSIS_OPF_STATS)) {
            rc = fail(WSIS_ERR_ARG, "op %d: BatchNorm partials requested from a dIn pass that is not on wsis_spconv_fwd_t", i);
            break;
          }
          if (wt_off[i] >= 0) {
            const float* WT = reinterpret_cast<const float*>(wt_base + wt_off[i]);
            rc = wsis_spconv_fwd((const float*)op.in[2], (const int32_t*)op.in[5], (const int32_t*)op.in[6], WT, nullptr,
                                 nullptr, (float*)op.out[0], op.M_out, op.M_in, op.K, op.Cout, op.Cin, rest, rest_bytes,
                                 stream);
          } else if (op.flags & WSIS_OPF_STATS) {
            // dX is the dy of a BatchNorm backward a few ops on (WSIS_OP_BN_RELU_BWD with in[1] == this dX, flagged
            // too): the epilogue also makes that op's reduction, into its in[7]
            const wsis_op* bn = nullptr;
            for (int j = i + 1; j < n && j <= i + 4 && !bn; ++j)
              if (ops[j].kind == WSIS_OP_BN_RELU_BWD && ops[j].in[1] == op.out[0] && (ops[j].flags & WSIS_OPF_STATS))
                bn = &ops[j];
            if (!bn || !bn->in[7] || bn->M_in != op.M_in || bn->Cin != op.Cin) {
              rc = fail(WSIS_ERR_ARG, "op %d: no BatchNorm backward takes the partials of this dIn pass", i);
              break;
            }
            if (!slab_bn_partials && wsis_spconv_fwd_t_slabs(op.M_in, op.K, op.Cout, op.Cin) > 1) {
              // few slices: the product is split into offset slabs and summed by a second kernel; the BatchNorm's own
              // small-level reduction (one launch) is cheaper than a statistics variant of that sum
              bn_unfused[bn - ops] = 1;
              rc = wsis_spconv_fwd_t((const float*)op.in[2], (const int32_t*)op.in[5], (const int32_t*)op.in[6],
                                     (const float*)op.in[1], (op.flags & WSIS_OPF_FLIP) ? 1 : 0, nullptr, nullptr,
                                     (float*)op.out[0], nullptr, op.M_out, op.M_in, op.K, op.Cout, op.Cin, rest,
                                     rest_bytes, sync_slot(d_sync, i), stream);
              if (rc != WSIS_OK) break;
              goto din_done;
            }
            rc = wsis_spconv_fwd_t_bn((const float*)op.in[2], (const int32_t*)op.in[5], (const int32_t*)op.in[6],
                                      (const float*)op.in[1], (op.flags & WSIS_OPF_FLIP) ? 1 : 0, (float*)op.out[0],
                                      (float*)const_cast<void*>(bn->in[7]), (const float*)bn->in[0],
                                      (const float*)bn->in[2], (const float*)bn->in[3], (const float*)bn->in[4],
                                      (const float*)bn->in[5], bn->eps, (bn->flags & WSIS_OPF_RELU) ? 1 : 0, op.M_out,
                                      op.M_in, op.K, op.Cout, op.Cin, rest, rest_bytes, sync_slot(d_sync, i), stream);
          } else {   // the weight [K, Cin, Cout] is the B^T operand of the dIn product as it stands
            rc = wsis_spconv_fwd_t((const float*)op.in[2], (const int32_t*)op.in[5], (const int32_t*)op.in[6],
                                   (const float*)op.in[1], (op.flags & WSIS_OPF_FLIP) ? 1 : 0, nullptr, nullptr,
                                   (float*)op.out[0], nullptr, op.M_out, op.M_in, op.K, op.Cout, op.Cin, rest, rest_bytes,
                                   sync_slot(d_sync, i), stream);
          }
          if (rc != WSIS_OK) break;
        }
      din_done:
        if (op.out[1] && !dw_forked) {
          if (tail_main && side && dw_tail_on_main(ops, n, i)) {
            // (never deferred: the batched slab sum runs on the side stream, these slabs are written on this one)
            rc = issue_dw(op, ws, ws_bytes, stream, nullptr);
          } else {
            rc = dw_issue(op);
          }
        }
        break;
      }
      case WSIS_OP_BN_RELU_BWD:
        if ((op.flags & WSIS_OPF_STATS) && !bn_unfused[i]) {      // the producing dIn pass left the slice partials in in[7]
          if (!op.in[7] || !(op.flags & WSIS_OPF_TRAINING)) {
            rc = fail(WSIS_ERR_ARG, "op %d: BatchNorm backward from partials needs in[7] and training mode", i);
            break;
          }
          rc = wsis_bn_bwd_from_partials((const float*)op.in[7], (op.M_in + 31) / 32, (const float*)op.in[0],
                                         (const float*)op.in[1], (const float*)op.in[2], (const float*)op.in[3],
                                         (const float*)op.in[4], (const float*)op.in[5], op.eps,
                                         (op.flags & WSIS_OPF_RELU) ? 1 : 0, (float*)op.out[0], (float*)op.out[1],
                                         (float*)op.out[2], (const float*)op.in[6], op.M_in, op.Cin, ws, ws_bytes,
                                         sync_slot(d_sync, i), stream);
          break;
        }
        rc = wsis_bn_bwd((const float*)op.in[0], (const float*)op.in[1], (const float*)op.in[2], (const float*)op.in[3],
                         (const float*)op.in[4], (const float*)op.in[5], op.eps, (op.flags & WSIS_OPF_RELU) ? 1 : 0,
                         (op.flags & WSIS_OPF_TRAINING) ? 1 : 0, (float*)op.out[0], (float*)op.out[1], (float*)op.out[2],
                         (const float*)op.in[6], op.M_in, op.Cin, ws, ws_bytes, stream);
        break;
      default:
        rc = fail(WSIS_ERR_ARG, "wsis_run_ops: unknown op kind");
    }
    if (bn_prof) bn_prof->stop();
    }
    if (rc != WSIS_OK) {
      first_err = rc;
      break;
    }
#if WSIS_EXPERIMENTAL
  op_done:
#endif
    if (i == mark_op) {
      // milestone: waiter_stream continues once everything issued so far -- on the caller's stream AND on the
      // weight-gradient side stream -- has run (gradient exchange of the finished part of the flat buffer while the
      // rest of the pass still executes)
      SideStream* ms = side ? side : side_stream_for(st);
      {
        int wrc = dw_flush();               // the weight gradients up to here are FINAL behind their stream's event
        if (wrc == WSIS_OK) wrc = drain_worker();     // the dW launches up to here are on the side stream before its event
        if (wrc != WSIS_OK) {
          first_err = wrc;
          break;
        }
      }
      hipError_t e = ms ? hipEventRecord(ms->mark_main, st) : hipErrorOutOfMemory;
      if (e == hipSuccess) e = hipStreamWaitEvent(as_stream(waiter_stream), ms->mark_main, 0);
      if (e == hipSuccess && forked) {
        e = hipEventRecord(ms->mark_side, side->stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(as_stream(waiter_stream), ms->mark_side, 0);
      }
      if (e != hipSuccess) {
        first_err = fail(WSIS_ERR_HIP, "wsis_run_ops: milestone failed: %s", hipGetErrorString(e));
        break;
      }
    }
  }
#undef RUN_LAUNCH_CHECK
  {
    int wrc = first_err == WSIS_OK ? dw_flush() : WSIS_OK;      // (every part of a pass finishes its own products)
    if (wrc != WSIS_OK && first_err == WSIS_OK) first_err = wrc;
    wrc = drain_worker();
    if (wrc != WSIS_OK && first_err == WSIS_OK) first_err = wrc;
  }
  if (defer_join && first_err == WSIS_OK) {
    if (forked) side->pending_join = true;
    return WSIS_OK;
  }
  if (!forked && !side && dw_stream_enabled()) {      // a part without weight gradients of its own may owe an earlier part's join
    SideStream* s2 = side_stream_for(st);
    if (s2 && s2->pending_join) {
      side = s2;
      forked = true;
    }
  } else if (side && side->pending_join) {
    forked = true;
  }
  if (forked) {   // join on EVERY exit path once forked: whatever follows on the caller's stream (optimizer, gradient
                  // all-reduce, the caller's error handling) is ordered behind every dW launch already issued
    side->pending_join = false;
    const hipError_t e1 = hipEventRecord(side->join, side->stream);
    const hipError_t e2 = e1 == hipSuccess ? hipStreamWaitEvent(st, side->join, 0) : e1;
    if (e2 != hipSuccess && first_err == WSIS_OK)
      return fail(WSIS_ERR_HIP, "side-stream join failed: %s", hipGetErrorString(e2));
  }
  return first_err;
}

#if WSIS_EXPERIMENTAL
// diagnostic (not part of the ABI header): phases of profiled resident launch `which` (0 = first since profiling was
// switched on): kind, NW, ZS, rows, Cin, Cout, K and the time from the end of the previous phase to the end of this one
// (grid barrier included) in microseconds; returns the number of phases (or -1)
int wsis_debug_deep_phases(int32_t which, int32_t* info, double* us, int32_t cap) {
  if (which < 0 || which >= (int)g_deep_log.size()) return -1;
  const auto& rec = g_deep_log[which];
  const int n = (int)rec.first.size();
  if (n > cap) return -1;
  std::vector<unsigned long long> t(n + 1);
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpy(t.data(), rec.second, (n + 1) * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  for (int i = 0; i < n; ++i) {
    const DeepOp& d = rec.first[i];
    info[i * 8 + 0] = d.kind;
    info[i * 8 + 1] = d.NW;
    info[i * 8 + 2] = d.ZS;
    info[i * 8 + 3] = (int32_t)(d.kind == DK_CONV || d.kind == DK_REDUCE ? d.M_out : d.M_in);
    info[i * 8 + 4] = d.Cin;
    info[i * 8 + 5] = d.Cout;
    info[i * 8 + 6] = d.K;
    info[i * 8 + 7] = (int32_t)d.M_in;
    us[i] = (double)(t[i + 1] - t[i]) * 0.01;
  }
  return n;
}

#endif

}  // extern "C"
