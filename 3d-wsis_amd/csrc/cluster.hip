// pointgroup_ops.ballquery_batch_p [UPSTREAM PG_OP ballquery_batch_p_cuda_].
// No call site in the reference; named by BASELINE.json north_star.
//
// For point p: ascending indices k of same-batch points with |x_p - x_k|^2 < r^2 (strict, p
// included), at most 1000.  Upstream hands out `start` from a global atomic cursor (order
// non-deterministic); here start = exclusive prefix sum of the counts -- a valid instance of the
// upstream contract and run-to-run deterministic.  Count pass + fill pass, candidate points are
// staged through LDS in tiles so every HBM byte of xyz is read once per workgroup.
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

template <bool FILL>
__global__ __launch_bounds__(BQ_BLOCK) void ballquery_kernel(
    const float* __restrict__ xyz, const int32_t* __restrict__ batch_idx, const int32_t* __restrict__ batch_off,
    int64_t N, float r2, int32_t* __restrict__ counts, const int32_t* __restrict__ start_len,
    int32_t* __restrict__ idx) {
  __shared__ float tile[BQ_BLOCK * 3];
  __shared__ int32_t range[2];
  const int64_t p = (int64_t)blockIdx.x * BQ_BLOCK + threadIdx.x;
  const bool live = p < N;
  float px = 0.f, py = 0.f, pz = 0.f;
  int32_t lo = 0, hi = 0;
  if (live) {
    px = xyz[3 * p];
    py = xyz[3 * p + 1];
    pz = xyz[3 * p + 2];
    const int32_t b = batch_idx[p];
    lo = batch_off[b];
    hi = batch_off[b + 1];
  }
  // candidate range of the whole workgroup = union of its points' batch ranges
  if (threadIdx.x == 0) {
    const int64_t first = (int64_t)blockIdx.x * BQ_BLOCK;
    const int64_t last = min(N, first + BQ_BLOCK) - 1;
    range[0] = batch_off[batch_idx[first]];
    range[1] = batch_off[batch_idx[last] + 1];
  }
  __syncthreads();
  const int32_t glo = range[0], ghi = range[1];
  int32_t cnt = 0;
  int32_t wpos = 0;
  if (FILL && live) wpos = start_len[2 * p];
  for (int32_t t0 = glo; t0 < ghi; t0 += BQ_BLOCK) {
    const int32_t k = t0 + threadIdx.x;
    if (k < ghi) {
      tile[3 * threadIdx.x] = xyz[3 * (int64_t)k];
      tile[3 * threadIdx.x + 1] = xyz[3 * (int64_t)k + 1];
      tile[3 * threadIdx.x + 2] = xyz[3 * (int64_t)k + 2];
    }
    __syncthreads();
    if (live && cnt < BQ_CAP) {
      const int32_t a = max(t0, lo), bnd = min(min(t0 + BQ_BLOCK, ghi), hi);
      for (int32_t kk = a; kk < bnd; ++kk) {
        const int j = kk - t0;
        const float dx = px - tile[3 * j], dy = py - tile[3 * j + 1], dz = pz - tile[3 * j + 2];
        const float d2 = dx * dx + dy * dy + dz * dz;
        if (d2 < r2) {
          if (FILL) idx[(int64_t)wpos + cnt] = kk;
          ++cnt;
          if (cnt >= BQ_CAP) break;
        }
      }
    }
    __syncthreads();
  }
  if (!FILL && live) counts[p] = cnt;
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

int64_t wsis_ballquery_workspace_bytes(int64_t N) {
  if (N < 0) return -1;
  if (N == 0) return 256;
  size_t scan_bytes = 0;
  int32_t* ip = nullptr;
  if (rocprim::exclusive_scan(nullptr, scan_bytes, ip, ip, 0, (size_t)N, rocprim::plus<int32_t>(),
                              (hipStream_t)0) != hipSuccess)
    return -1;
  return (int64_t)(2 * align256((size_t)N * 4) + align256(scan_bytes) + 256);
}

int wsis_ballquery_count(const float* d_xyz, const int32_t* d_batch_idx, const int32_t* d_batch_off,
                         int64_t N, int32_t B, float radius, int32_t* d_start_len, int32_t* d_total,
                         void* d_ws, int64_t ws_bytes, void* stream) {
  WSIS_REQUIRE(N >= 0 && B >= 0 && radius >= 0.f && d_total, "bad args");
  hipStream_t st = as_stream(stream);
  WSIS_HIP_CHECK(hipMemsetAsync(d_total, 0, sizeof(int32_t), st));
  if (N == 0) return WSIS_OK;
  WSIS_REQUIRE(d_xyz && d_batch_idx && d_batch_off && d_start_len && d_ws, "null pointer");
  char* ws = static_cast<char*>(d_ws);
  const size_t a = align256((size_t)N * 4);
  WSIS_REQUIRE((int64_t)(2 * a) < ws_bytes, "workspace too small");
  int32_t* counts = reinterpret_cast<int32_t*>(ws);
  int32_t* starts = reinterpret_cast<int32_t*>(ws + a);
  void* temp = ws + 2 * a;
  size_t temp_bytes = (size_t)ws_bytes - 2 * a;
  const unsigned grid = (unsigned)ceil_div(N, BQ_BLOCK);
  hipLaunchKernelGGL(ballquery_kernel<false>, dim3(grid), dim3(BQ_BLOCK), 0, st, d_xyz, d_batch_idx,
                     d_batch_off, N, radius * radius, counts, (const int32_t*)nullptr, (int32_t*)nullptr);
  WSIS_LAUNCH_CHECK();
  size_t need = 0;
  WSIS_HIP_CHECK(rocprim::exclusive_scan(nullptr, need, counts, starts, 0, (size_t)N, rocprim::plus<int32_t>(), st));
  WSIS_REQUIRE(need <= temp_bytes, "workspace too small for scan");
  WSIS_HIP_CHECK(rocprim::exclusive_scan(temp, temp_bytes, counts, starts, 0, (size_t)N,
                                         rocprim::plus<int32_t>(), st));
  hipLaunchKernelGGL(start_len_kernel, dim3(grid_for(N, 256)), dim3(256), 0, st, counts, starts, N, d_start_len,
                     d_total);
  WSIS_LAUNCH_CHECK();
  return WSIS_OK;
}

int wsis_ballquery_fill(const float* d_xyz, const int32_t* d_batch_idx, const int32_t* d_batch_off,
                        int64_t N, int32_t B, float radius, const int32_t* d_start_len, int32_t* d_idx,
                        int64_t total, void* d_ws, int64_t ws_bytes, void* stream) {
  WSIS_REQUIRE(N >= 0 && B >= 0 && radius >= 0.f && total >= 0, "bad args");
  (void)d_ws;
  (void)ws_bytes;
  if (N == 0 || total == 0) return WSIS_OK;
  WSIS_REQUIRE(d_xyz && d_batch_idx && d_batch_off && d_start_len && d_idx, "null pointer");
  const unsigned grid = (unsigned)ceil_div(N, BQ_BLOCK);
  hipLaunchKernelGGL(ballquery_kernel<true>, dim3(grid), dim3(BQ_BLOCK), 0, as_stream(stream), d_xyz,
                     d_batch_idx, d_batch_off, N, radius * radius, (int32_t*)nullptr, d_start_len, d_idx);
  WSIS_LAUNCH_CHECK();
  return WSIS_OK;
}

}  // extern "C"
