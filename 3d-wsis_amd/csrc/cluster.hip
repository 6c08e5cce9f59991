// pointgroup_ops.ballquery_batch_p [UPSTREAM PG_OP ballquery_batch_p_cuda_].
// No call site in the reference; named by BASELINE.json north_star.
//
// For point p: ascending indices k of same-batch points with |x_p - x_k|^2 < r^2 (strict, p
// included), at most 1000.  Upstream hands out `start` from a global atomic cursor (order
// non-deterministic); here start = exclusive prefix sum of the counts -- a valid instance of the
// upstream contract and run-to-run deterministic.
//
// Upstream is a brute-force O(N * N_batch) scan.  Here the points are bucketed into a uniform grid with cell size
// (just above) the radius: radix sort by cell key, a hash of cell -> [start,end) in the sorted order, and each
// point only visits the 27 cells around it (count pass, exclusive scan, fill pass + per-point ascending sort).
// Distances are evaluated with exactly the same fp32 expression as the scan, so the neighbour sets are identical.
// Points with more than 1000 hits (upstream keeps the 1000 smallest indices) are finished by an ascending
// wave-wide scan over their batch item.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_select.hpp>

#include "common.h"

using namespace wsis;

namespace {

constexpr int BQ_BLOCK = 256;
constexpr int BQ_CAP = 1000;

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// ---- uniform-grid path ------------------------------------------------------------------------------------
constexpr int CELL_BITS = 18;                       // per-dimension cell coordinate bits (bias 2^17)
constexpr int CELL_BIAS = 1 << (CELL_BITS - 1);

__device__ __forceinline__ uint64_t cell_key(int b, int cx, int cy, int cz) {
  return ((uint64_t)(uint32_t)b << (3 * CELL_BITS)) | ((uint64_t)(uint32_t)(cx + CELL_BIAS) << (2 * CELL_BITS)) |
         ((uint64_t)(uint32_t)(cy + CELL_BIAS) << CELL_BITS) | (uint64_t)(uint32_t)(cz + CELL_BIAS);
}

__global__ void bq_keys_kernel(const float* __restrict__ xyz, const int32_t* __restrict__ batch_idx, int64_t N,
                               float inv_cell, uint64_t* __restrict__ keys, int32_t* __restrict__ iota) {
  for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < N; p += (int64_t)gridDim.x * blockDim.x) {
    const int cx = (int)floorf(xyz[3 * p] * inv_cell), cy = (int)floorf(xyz[3 * p + 1] * inv_cell),
              cz = (int)floorf(xyz[3 * p + 2] * inv_cell);
    keys[p] = cell_key(batch_idx[p], cx, cy, cz);
    iota[p] = (int32_t)p;
  }
}

// cell table: open addressing on the 64-bit cell key; start set by the first sorted point of a cell, end by the last
__global__ void bq_cells_kernel(const uint64_t* __restrict__ sorted, int64_t N, unsigned long long* __restrict__ hkeys,
                                int32_t* __restrict__ hstart, int32_t* __restrict__ hend, uint64_t mask) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t key = sorted[i];
    const bool first = i == 0 || sorted[i - 1] != key;
    const bool last = i == N - 1 || sorted[i + 1] != key;
    if (!first && !last) continue;
    uint64_t h = mix64(key) & mask;
    for (;;) {
      const unsigned long long old = atomicCAS(hkeys + h, ~0ull, (unsigned long long)key);
      if (old == ~0ull || old == (unsigned long long)key) break;
      h = (h + 1) & mask;
    }
    if (first) hstart[h] = (int32_t)i;
    if (last) hend[h] = (int32_t)i + 1;
  }
}

__device__ __forceinline__ bool bq_cell_range(const unsigned long long* __restrict__ hkeys,
                                              const int32_t* __restrict__ hstart, const int32_t* __restrict__ hend,
                                              uint64_t mask, uint64_t key, int32_t& s, int32_t& e) {
  uint64_t h = mix64(key) & mask;
  for (;;) {
    const unsigned long long k = hkeys[h];
    if (k == (unsigned long long)key) {
      s = hstart[h];
      e = hend[h];
      return true;
    }
    if (k == ~0ull) return false;
    h = (h + 1) & mask;
  }
}

// FILL = false: counts[p] = true number of hits (uncapped).  FILL = true: writes the hits of points with
// <= BQ_CAP hits (unsorted, then sorted ascending in place); points above the cap are left to the scan path.
template <bool FILL>
__global__ void bq_grid_kernel(const float* __restrict__ xyz, const int32_t* __restrict__ batch_idx, int64_t N,
                               float r2, float inv_cell, const int32_t* __restrict__ perm,
                               const unsigned long long* __restrict__ hkeys, const int32_t* __restrict__ hstart,
                               const int32_t* __restrict__ hend, uint64_t mask, int32_t* __restrict__ counts,
                               const int32_t* __restrict__ start_len, int32_t* __restrict__ idx) {
  for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < N; p += (int64_t)gridDim.x * blockDim.x) {
    const float px = xyz[3 * p], py = xyz[3 * p + 1], pz = xyz[3 * p + 2];
    const int b = batch_idx[p];
    const int cx = (int)floorf(px * inv_cell), cy = (int)floorf(py * inv_cell), cz = (int)floorf(pz * inv_cell);
    int32_t cnt = 0;
    int32_t* dst = nullptr;
    if (FILL) {
      if (counts[p] > BQ_CAP) continue;   // handled by the ascending scan kernel
      dst = idx + start_len[2 * p];
    }
    for (int dz = -1; dz <= 1; ++dz)
      for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
          int32_t s, e;
          if (!bq_cell_range(hkeys, hstart, hend, mask, cell_key(b, cx + dx, cy + dy, cz + dz), s, e)) continue;
          for (int32_t i = s; i < e; ++i) {
            const int32_t k = perm[i];
            const float ex = px - xyz[3 * (int64_t)k], ey = py - xyz[3 * (int64_t)k + 1],
                        ez = pz - xyz[3 * (int64_t)k + 2];
            const float d2 = ex * ex + ey * ey + ez * ez;
            if (d2 < r2) {
              if (FILL) dst[cnt] = k;
              ++cnt;
            }
          }
        }
    if (!FILL) {
      counts[p] = cnt;
    } else {
      // ascending neighbour order (upstream scans k upward): insertion sort of the short list
      for (int32_t i = 1; i < cnt; ++i) {
        const int32_t v = dst[i];
        int32_t j = i - 1;
        while (j >= 0 && dst[j] > v) {
          dst[j + 1] = dst[j];
          --j;
        }
        dst[j + 1] = v;
      }
    }
  }
}

// points with more than BQ_CAP hits: ascending scan over the batch item, first BQ_CAP hits (one wave per point)
__global__ void bq_overflow_kernel(const float* __restrict__ xyz, const int32_t* __restrict__ batch_idx,
                                   const int32_t* __restrict__ batch_off, int64_t N, float r2,
                                   const int32_t* __restrict__ counts, const int32_t* __restrict__ start_len,
                                   int32_t* __restrict__ idx) {
  const int lane = threadIdx.x & 63;
  const int64_t nw = (int64_t)gridDim.x * (blockDim.x >> 6);
  for (int64_t p = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); p < N; p += nw) {
    if (counts[p] <= BQ_CAP) continue;
    const float px = xyz[3 * p], py = xyz[3 * p + 1], pz = xyz[3 * p + 2];
    const int b = batch_idx[p];
    int32_t* dst = idx + start_len[2 * p];
    int32_t n = 0;
    for (int32_t k0 = batch_off[b]; k0 < batch_off[b + 1] && n < BQ_CAP; k0 += 64) {
      const int32_t k = k0 + lane;
      bool hit = false;
      if (k < batch_off[b + 1]) {
        const float ex = px - xyz[3 * (int64_t)k], ey = py - xyz[3 * (int64_t)k + 1], ez = pz - xyz[3 * (int64_t)k + 2];
        hit = (ex * ex + ey * ey + ez * ez) < r2;
      }
      const unsigned long long m = __ballot(hit);
      const int pos = n + __popcll(m & ((1ull << lane) - 1ull));
      if (hit && pos < BQ_CAP) dst[pos] = k;
      n += __popcll(m);
    }
  }
}

__global__ void bq_cap_kernel(int32_t* __restrict__ counts_capped, const int32_t* __restrict__ counts, int64_t N) {
  for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < N; p += (int64_t)gridDim.x * blockDim.x)
    counts_capped[p] = min(counts[p], BQ_CAP);
}

__global__ void start_len_kernel(const int32_t* __restrict__ counts, const int32_t* __restrict__ starts,
                                 int64_t N, int32_t* __restrict__ start_len, int32_t* __restrict__ total) {
  for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < N;
       p += (int64_t)gridDim.x * blockDim.x) {
    start_len[2 * p] = starts[p];
    start_len[2 * p + 1] = counts[p];
    if (p == N - 1) *total = starts[p] + counts[p];
  }
}

}  // namespace

extern "C" {

// workspace layout (all 256-byte aligned):
//   counts_true[N] i32 | counts_cap[N] i32 | starts[N] i32 | perm[N] i32 | iota[N] i32 | keys[N] u64 |
//   keys_sorted[N] u64 | hkeys[cap] u64 | hstart[cap] i32 | hend[cap] i32 | temp (sort / scan)
struct BqLayout {
  size_t counts, capped, starts, perm, iota, keys, keys_sorted, hkeys, hstart, hend, temp, total;
  int64_t cap;
};

static BqLayout bq_layout(int64_t N, size_t temp_bytes) {
  BqLayout L;
  int64_t cap = 16;
  while (cap < 2 * N) cap <<= 1;
  L.cap = cap;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    const size_t o = off;
    off += align256(bytes);
    return o;
  };
  L.counts = take((size_t)N * 4);
  L.capped = take((size_t)N * 4);
  L.starts = take((size_t)N * 4);
  L.perm = take((size_t)N * 4);
  L.iota = take((size_t)N * 4);
  L.keys = take((size_t)N * 8);
  L.keys_sorted = take((size_t)N * 8);
  L.hkeys = take((size_t)cap * 8);
  L.hstart = take((size_t)cap * 4);
  L.hend = take((size_t)cap * 4);
  L.temp = take(temp_bytes);
  L.total = off + 256;
  return L;
}

static int bq_temp_bytes(int64_t N, size_t* out) {
  size_t scan_bytes = 0, sort_bytes = 0;
  int32_t* ip = nullptr;
  uint64_t* kp = nullptr;
  if (rocprim::exclusive_scan(nullptr, scan_bytes, ip, ip, 0, (size_t)N, rocprim::plus<int32_t>(),
                              (hipStream_t)0) != hipSuccess)
    return -1;
  if (rocprim::radix_sort_pairs(nullptr, sort_bytes, kp, kp, ip, ip, (size_t)N, 0, 64, (hipStream_t)0) != hipSuccess)
    return -1;
  *out = scan_bytes > sort_bytes ? scan_bytes : sort_bytes;
  return 0;
}

int64_t wsis_ballquery_workspace_bytes(int64_t N) {
  if (N < 0) return -1;
  if (N == 0) return 256;
  size_t temp = 0;
  if (bq_temp_bytes(N, &temp) != 0) return -1;
  return (int64_t)bq_layout(N, temp).total;
}

int wsis_ballquery_count(const float* d_xyz, const int32_t* d_batch_idx, const int32_t* d_batch_off,
                         int64_t N, int32_t B, float radius, int32_t* d_start_len, int32_t* d_total,
                         void* d_ws, int64_t ws_bytes, void* stream) {
  WSIS_REQUIRE(N >= 0 && B >= 0 && radius >= 0.f && d_total, "bad args");
  hipStream_t st = as_stream(stream);
  WSIS_HIP_CHECK(hipMemsetAsync(d_total, 0, sizeof(int32_t), st));
  if (N == 0) return WSIS_OK;
  WSIS_REQUIRE(d_xyz && d_batch_idx && d_batch_off && d_start_len && d_ws, "null pointer");
  WSIS_REQUIRE(B < (1 << 10), "ball query supports < 1024 batch items");
  size_t temp_bytes = 0;
  WSIS_REQUIRE(bq_temp_bytes(N, &temp_bytes) == 0, "rocprim size query failed");
  const BqLayout L = bq_layout(N, temp_bytes);
  WSIS_REQUIRE((int64_t)L.total <= ws_bytes, "workspace too small");
  char* ws = static_cast<char*>(d_ws);
  int32_t* counts = reinterpret_cast<int32_t*>(ws + L.counts);
  int32_t* capped = reinterpret_cast<int32_t*>(ws + L.capped);
  int32_t* starts = reinterpret_cast<int32_t*>(ws + L.starts);
  int32_t* perm = reinterpret_cast<int32_t*>(ws + L.perm);
  int32_t* iota = reinterpret_cast<int32_t*>(ws + L.iota);
  uint64_t* keys = reinterpret_cast<uint64_t*>(ws + L.keys);
  uint64_t* keys_sorted = reinterpret_cast<uint64_t*>(ws + L.keys_sorted);
  unsigned long long* hkeys = reinterpret_cast<unsigned long long*>(ws + L.hkeys);
  int32_t* hstart = reinterpret_cast<int32_t*>(ws + L.hstart);
  int32_t* hend = reinterpret_cast<int32_t*>(ws + L.hend);
  void* temp = ws + L.temp;
  const float r2 = radius * radius;
  if (radius <= 0.f) {   // degenerate: no strict-inequality hit at all
    WSIS_HIP_CHECK(hipMemsetAsync(counts, 0, (size_t)N * 4, st));
  } else {
    const float inv_cell = 1.0f / (radius * 1.001f);   // cell slightly larger than r: neighbours within +-1 cell
    const int g = grid_for(N, 256);
    hipLaunchKernelGGL(bq_keys_kernel, dim3(g), dim3(256), 0, st, d_xyz, d_batch_idx, N, inv_cell, keys, iota);
    WSIS_LAUNCH_CHECK();
    size_t tb = temp_bytes;
    WSIS_HIP_CHECK(rocprim::radix_sort_pairs(temp, tb, keys, keys_sorted, iota, perm, (size_t)N, 0, 64, st));
    WSIS_HIP_CHECK(hipMemsetAsync(hkeys, 0xFF, (size_t)L.cap * 8, st));
    hipLaunchKernelGGL(bq_cells_kernel, dim3(g), dim3(256), 0, st, keys_sorted, N, hkeys, hstart, hend,
                       (uint64_t)(L.cap - 1));
    WSIS_LAUNCH_CHECK();
    hipLaunchKernelGGL(bq_grid_kernel<false>, dim3(g), dim3(256), 0, st, d_xyz, d_batch_idx, N, r2, inv_cell, perm,
                       hkeys, hstart, hend, (uint64_t)(L.cap - 1), counts, (const int32_t*)nullptr, (int32_t*)nullptr);
    WSIS_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(bq_cap_kernel, dim3(grid_for(N, 256)), dim3(256), 0, st, capped, counts, N);
  WSIS_LAUNCH_CHECK();
  size_t tb = temp_bytes;
  WSIS_HIP_CHECK(rocprim::exclusive_scan(temp, tb, capped, starts, 0, (size_t)N, rocprim::plus<int32_t>(), st));
  hipLaunchKernelGGL(start_len_kernel, dim3(grid_for(N, 256)), dim3(256), 0, st, capped, starts, N, d_start_len,
                     d_total);
  WSIS_LAUNCH_CHECK();
  return WSIS_OK;
}

int wsis_ballquery_fill(const float* d_xyz, const int32_t* d_batch_idx, const int32_t* d_batch_off,
                        int64_t N, int32_t B, float radius, const int32_t* d_start_len, int32_t* d_idx,
                        int64_t total, void* d_ws, int64_t ws_bytes, void* stream) {
  WSIS_REQUIRE(N >= 0 && B >= 0 && radius >= 0.f && total >= 0, "bad args");
  if (N == 0 || total == 0) return WSIS_OK;
  WSIS_REQUIRE(d_xyz && d_batch_idx && d_batch_off && d_start_len && d_idx && d_ws, "null pointer");
  // the workspace still holds the grid of wsis_ballquery_count (same N, same stream order)
  size_t temp_bytes = 0;
  WSIS_REQUIRE(bq_temp_bytes(N, &temp_bytes) == 0, "rocprim size query failed");
  const BqLayout L = bq_layout(N, temp_bytes);
  WSIS_REQUIRE((int64_t)L.total <= ws_bytes, "workspace too small");
  char* ws = static_cast<char*>(d_ws);
  int32_t* counts = reinterpret_cast<int32_t*>(ws + L.counts);
  int32_t* perm = reinterpret_cast<int32_t*>(ws + L.perm);
  unsigned long long* hkeys = reinterpret_cast<unsigned long long*>(ws + L.hkeys);
  int32_t* hstart = reinterpret_cast<int32_t*>(ws + L.hstart);
  int32_t* hend = reinterpret_cast<int32_t*>(ws + L.hend);
  hipStream_t st = as_stream(stream);
  const float r2 = radius * radius;
  const float inv_cell = 1.0f / (radius * 1.001f);
  hipLaunchKernelGGL(bq_grid_kernel<true>, dim3(grid_for(N, 256)), dim3(256), 0, st, d_xyz, d_batch_idx, N, r2,
                     inv_cell, perm, hkeys, hstart, hend, (uint64_t)(L.cap - 1), counts, d_start_len, d_idx);
  WSIS_LAUNCH_CHECK();
  hipLaunchKernelGGL(bq_overflow_kernel, dim3(grid_for(N * 64, 256)), dim3(256), 0, st, d_xyz, d_batch_idx, d_batch_off,
                     N, r2, counts, d_start_len, d_idx);
  WSIS_LAUNCH_CHECK();
  return WSIS_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------------------------
// bfs_cluster on the device (upstream PointGroup runs it on the host).  Same results as the
// host FIFO walk, element for element:
//   1. connected components of the ball-query graph restricted to equal semantic labels by lock-free union-find
//      (the larger root is hooked under the smaller one, so a component's root is its SMALLEST point index = the
//      seed the host walk starts it from);
//   2. the caller keeps the components with >= threshold points, numbers them in seed order and lays their point
//      lists out back to back (a few torch calls + the one host read of the output sizes);
//   3. one workgroup per kept component replays the FIFO order level by level, with its slice of the output as the
//      queue: every frontier point stamps its unvisited neighbours with atomicMin(its queue position); a neighbour
//      belongs to the parent whose stamp survived; parents append their own children in list order (ball-query lists
//      are ascending) at offsets from a prefix sum over the frontier -- exactly the order in which the host queue would
//      have received them.
namespace {

__device__ __forceinline__ int32_t uf_find(int32_t* parent, int32_t v) {
  int32_t p = parent[v];
  while (p != v) {          // path halving
    const int32_t g = parent[p];
    if (g != p) parent[v] = g;
    v = p;
    p = g;
  }
  return v;
}

__global__ void cc_init_kernel(int32_t* __restrict__ parent, int64_t N) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x)
    parent[i] = (int32_t)i;
}

// one thread per point: union with every same-label neighbour of larger index (each undirected pair once)
__global__ void cc_union_kernel(const int32_t* __restrict__ sem, const int32_t* __restrict__ idx,
                                const int32_t* __restrict__ start_len, int32_t* parent, int64_t N) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
    const int32_t s = start_len[2 * i], n = start_len[2 * i + 1], lab = sem[i];
    for (int32_t j = 0; j < n; ++j) {
      const int32_t v = idx[s + j];
      if (v <= (int32_t)i || sem[v] != lab) continue;
      int32_t a = uf_find(parent, (int32_t)i), b = uf_find(parent, v);
      while (a != b) {
        if (a < b) {
          const int32_t t = a;
          a = b;
          b = t;
        }                                     // hook the larger root a under the smaller root b
        const int32_t old = atomicCAS(&parent[a], a, b);
        if (old == a) break;
        a = uf_find(parent, old);
        b = uf_find(parent, b);
      }
    }
  }
}

__global__ void cc_flatten_kernel(int32_t* parent, int32_t* __restrict__ root, int32_t* __restrict__ size, int64_t N) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
    const int32_t r = uf_find(parent, (int32_t)i);
    root[i] = r;
    atomicAdd(&size[r], 1);
  }
}

constexpr int BFS_THREADS = 256;

// one workgroup per kept component c: queue = out[offset[c] .. offset[c+1]) (second column of cluster_idxs)
__global__ __launch_bounds__(BFS_THREADS) void bfs_order_kernel(const int32_t* __restrict__ idx,
                                                                const int32_t* __restrict__ start_len,
                                                                const int32_t* __restrict__ sem,
                                                                const int32_t* __restrict__ seeds,
                                                                const int32_t* __restrict__ offsets, int32_t* pos,
                                                                int32_t* stamp, int32_t* __restrict__ cluster_idxs) {
  __shared__ int32_t scan[BFS_THREADS];
  __shared__ int32_t carry;
  const int c = blockIdx.x;
  const int32_t base = offsets[c], total = offsets[c + 1] - base;
  const int32_t seed = seeds[c], lab = sem[seed];
  int32_t* queue = cluster_idxs;       // rows (cluster id, point): queue[2 * (base + i) + 1]
  if (threadIdx.x == 0) {
    queue[2 * base] = c;
    queue[2 * base + 1] = seed;
    pos[seed] = 0;
  }
  __syncthreads();
  int32_t lo = 0, hi = 1;
  while (lo < hi && hi < total) {
    // pass 1: stamp the unvisited same-label neighbours with the parent's queue position
    for (int32_t i = lo + threadIdx.x; i < hi; i += BFS_THREADS) {
      const int32_t p = queue[2 * (base + i) + 1];
      const int32_t s = start_len[2 * p], n = start_len[2 * p + 1];
      for (int32_t j = 0; j < n; ++j) {
        const int32_t v = idx[s + j];
        if (sem[v] == lab && pos[v] < 0) atomicMin(&stamp[v], i);
      }
    }
    __syncthreads();
    // pass 2 + 3: children per parent, prefix sum over the frontier in queue order, append
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int32_t i0 = lo; i0 < hi; i0 += BFS_THREADS) {
      const int32_t i = i0 + threadIdx.x;
      int32_t cnt = 0, p = -1, s = 0, n = 0;
      if (i < hi) {
        p = queue[2 * (base + i) + 1];
        s = start_len[2 * p];
        n = start_len[2 * p + 1];
        for (int32_t j = 0; j < n; ++j) {
          const int32_t v = idx[s + j];
          if (sem[v] == lab && pos[v] < 0 && stamp[v] == i) ++cnt;
        }
      }
      scan[threadIdx.x] = cnt;
      __syncthreads();
      for (int off = 1; off < BFS_THREADS; off <<= 1) {     // inclusive scan
        const int32_t t = threadIdx.x >= off ? scan[threadIdx.x - off] : 0;
        __syncthreads();
        scan[threadIdx.x] += t;
        __syncthreads();
      }
      const int32_t before = carry + scan[threadIdx.x] - cnt;
      const int32_t chunk_total = scan[BFS_THREADS - 1];
      __syncthreads();
      if (i < hi && cnt > 0) {
        int32_t w = hi + before;
        for (int32_t j = 0; j < n; ++j) {
          const int32_t v = idx[s + j];
          // pos[v] is written only by the parent that owns v (stamp[v] == i) -- every v exactly once per level
          if (sem[v] == lab && stamp[v] == i && pos[v] < 0) {
            queue[2 * (base + w)] = c;
            queue[2 * (base + w) + 1] = v;
            pos[v] = w;
            ++w;
          }
        }
      }
      if (threadIdx.x == 0) carry += chunk_total;
      __syncthreads();
    }
    lo = hi;
    hi += carry;
    __syncthreads();
  }
}

}  // namespace

extern "C" {

int wsis_cc_same_label(const int32_t* d_sem, const int32_t* d_idx, const int32_t* d_start_len, int64_t N,
                       int32_t* d_parent, int32_t* d_root, int32_t* d_size, void* stream) {
  WSIS_REQUIRE(N >= 0, "bad size");
  if (N == 0) return WSIS_OK;
  WSIS_REQUIRE(d_sem && d_start_len && d_parent && d_root && d_size, "null pointer");
  WSIS_REQUIRE(N < ((int64_t)1 << 31), "point count exceeds int32");
  hipStream_t st = as_stream(stream);
  WSIS_HIP_CHECK(hipMemsetAsync(d_size, 0, sizeof(int32_t) * (size_t)N, st));
  hipLaunchKernelGGL(cc_init_kernel, dim3(grid_for(N, 256)), dim3(256), 0, st, d_parent, N);
  hipLaunchKernelGGL(cc_union_kernel, dim3(grid_for(N, 256)), dim3(256), 0, st, d_sem, d_idx, d_start_len, d_parent, N);
  hipLaunchKernelGGL(cc_flatten_kernel, dim3(grid_for(N, 256)), dim3(256), 0, st, d_parent, d_root, d_size, N);
  WSIS_LAUNCH_CHECK();
  return WSIS_OK;
}

int wsis_bfs_order(const int32_t* d_idx, const int32_t* d_start_len, const int32_t* d_sem, const int32_t* d_seeds,
                   const int32_t* d_offsets, int64_t n_clusters, int32_t* d_pos, int32_t* d_stamp,
                   int32_t* d_cluster_idxs, void* stream) {
  WSIS_REQUIRE(n_clusters >= 0, "bad size");
  if (n_clusters == 0) return WSIS_OK;
  WSIS_REQUIRE(d_start_len && d_sem && d_seeds && d_offsets && d_pos && d_stamp && d_cluster_idxs, "null pointer");
  hipLaunchKernelGGL(bfs_order_kernel, dim3((unsigned)n_clusters), dim3(BFS_THREADS), 0, as_stream(stream), d_idx,
                     d_start_len, d_sem, d_seeds, d_offsets, d_pos, d_stamp, d_cluster_idxs);
  WSIS_LAUNCH_CHECK();
  return WSIS_OK;
}

}  // extern "C"
