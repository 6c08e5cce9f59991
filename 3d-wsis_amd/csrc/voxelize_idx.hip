// GPU voxelization_idx: same contract as the host operator (libwsis_host.so /
// [UPSTREAM PG_OP.voxelize_idx], call site modules/datasets/scannetv2_dataset.py:449) -- voxel ids in order of first
// occurrence, per-voxel point lists ascending -- computed on the device so that batch assembly need not wait for
// a CPU worker.
//
//   1. hash: slot -> smallest point index of the voxel (CAS on an empty slot, coordinates compared through the
//      representative, atomicMin on a match)                      => first[p] = representative of p's voxel
//   2. is_first flags + exclusive scan                              => voxel id of every first point (= rank)
//   3. p2v[p] = vid[first[p]];  stable radix sort of (vid, p)       => points grouped per voxel, ascending p
//   4. counts / offsets from the sorted ids, v2p rows, voxel_locs   (count-then-fill: M and max_active are read
//      back by the caller between the two entry points)
// Integer work, bit-exact against the host operator (tests/test_gpu_ops.py).
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "common.h"

using namespace wsis;

namespace {

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

__device__ __forceinline__ uint64_t hash4(const int64_t* c) {
  uint64_t h = mix64((uint64_t)c[0] + 0x9e3779b97f4a7c15ULL);
  h = mix64(h ^ (uint64_t)c[1]);
  h = mix64(h ^ (uint64_t)c[2]);
  h = mix64(h ^ (uint64_t)c[3]);
  return h;
}
__device__ __forceinline__ bool same4(const int64_t* a, const int64_t* b) {
  return a[0] == b[0] && a[1] == b[1] && a[2] == b[2] && a[3] == b[3];
}

__global__ void vi_insert_kernel(const int64_t* __restrict__ coords, int64_t N, int32_t* __restrict__ slot,
                                 uint64_t mask) {
  for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < N; p += (int64_t)gridDim.x * blockDim.x) {
    const int64_t* c = coords + 4 * p;
    uint64_t h = hash4(c) & mask;
    for (;;) {
      const int32_t old = atomicCAS(slot + h, -1, (int32_t)p);
      if (old == -1) break;                                  // claimed an empty slot
      if (same4(coords + 4 * (int64_t)old, c)) {             // any representative of the voxel compares equal
        atomicMin(slot + h, (int32_t)p);
        break;
      }
      h = (h + 1) & mask;
    }
  }
}

__global__ void vi_first_kernel(const int64_t* __restrict__ coords, int64_t N, const int32_t* __restrict__ slot,
                                uint64_t mask, int32_t* __restrict__ first, int32_t* __restrict__ is_first) {
  for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < N; p += (int64_t)gridDim.x * blockDim.x) {
    const int64_t* c = coords + 4 * p;
    uint64_t h = hash4(c) & mask;
    for (;;) {
      const int32_t r = slot[h];
      if (same4(coords + 4 * (int64_t)r, c)) {
        first[p] = r;
        is_first[p] = (r == (int32_t)p) ? 1 : 0;
        break;
      }
      h = (h + 1) & mask;
    }
  }
}

__global__ void vi_p2v_kernel(const int32_t* __restrict__ first, const int32_t* __restrict__ rank, int64_t N,
                              int32_t* __restrict__ p2v, uint32_t* __restrict__ keys, int32_t* __restrict__ iota,
                              int32_t* __restrict__ vox_count) {
  for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < N; p += (int64_t)gridDim.x * blockDim.x) {
    const int32_t v = rank[first[p]];
    p2v[p] = v;
    keys[p] = (uint32_t)v;
    iota[p] = (int32_t)p;
    atomicAdd(vox_count + v, 1);
  }
}

__global__ void vi_summary_kernel(const int32_t* __restrict__ is_first, const int32_t* __restrict__ rank,
                                  const int32_t* __restrict__ vox_count, int64_t N, int32_t* __restrict__ counts2) {
  // counts2[0] = M, counts2[1] = max_active (atomicMax over the voxel counts)
  for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < N; p += (int64_t)gridDim.x * blockDim.x) {
    if (p == N - 1) counts2[0] = rank[p] + is_first[p];
    if (is_first[p]) atomicMax(counts2 + 1, vox_count[rank[p]]);
  }
}

__global__ void vi_fill_kernel(const int64_t* __restrict__ coords, const uint32_t* __restrict__ sorted_vid,
                               const int32_t* __restrict__ sorted_p, int64_t N, int32_t stride,
                               int64_t* __restrict__ voxel_locs, int32_t* __restrict__ v2p) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t v = sorted_vid[i];
    // position inside the voxel's run: distance to the run start (runs are short: <= max_active)
    int64_t s = i;
    while (s > 0 && sorted_vid[s - 1] == v) --s;
    const int32_t pos = (int32_t)(i - s);
    const int32_t p = sorted_p[i];
    int32_t* row = v2p + (int64_t)v * stride;
    row[1 + pos] = p;
    if (pos == 0) {
      for (int e = 0; e < 4; ++e) voxel_locs[4 * (int64_t)v + e] = coords[4 * (int64_t)p + e];
    }
    if (i == N - 1 || sorted_vid[i + 1] != v) row[0] = pos + 1;
  }
}

struct ViLayout {
  size_t slot, first, is_first, rank, vox_count, keys, keys_sorted, iota, sorted_p, temp, total;
  int64_t cap;
};

ViLayout vi_layout(int64_t N, size_t temp_bytes) {
  ViLayout L;
  int64_t cap = 16;
  while (cap < 2 * N) cap <<= 1;
  L.cap = cap;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    const size_t o = off;
    off += align256(bytes);
    return o;
  };
  L.slot = take((size_t)cap * 4);
  L.first = take((size_t)N * 4);
  L.is_first = take((size_t)N * 4);
  L.rank = take((size_t)N * 4);
  L.vox_count = take((size_t)N * 4);
  L.keys = take((size_t)N * 4);
  L.keys_sorted = take((size_t)N * 4);
  L.iota = take((size_t)N * 4);
  L.sorted_p = take((size_t)N * 4);
  L.temp = take(temp_bytes);
  L.total = off + 256;
  return L;
}

int vi_temp_bytes(int64_t N, size_t* out) {
  size_t scan_bytes = 0, sort_bytes = 0;
  int32_t* ip = nullptr;
  uint32_t* kp = nullptr;
  if (rocprim::exclusive_scan(nullptr, scan_bytes, ip, ip, 0, (size_t)N, rocprim::plus<int32_t>(),
                              (hipStream_t)0) != hipSuccess)
    return -1;
  if (rocprim::radix_sort_pairs(nullptr, sort_bytes, kp, kp, ip, ip, (size_t)N, 0, 32, (hipStream_t)0) != hipSuccess)
    return -1;
  *out = scan_bytes > sort_bytes ? scan_bytes : sort_bytes;
  return 0;
}

}  // namespace

extern "C" {

int64_t wsis_voxelize_idx_workspace_bytes(int64_t N) {
  if (N < 0) return -1;
  if (N == 0) return 256;
  size_t temp = 0;
  if (vi_temp_bytes(N, &temp) != 0) return -1;
  return (int64_t)vi_layout(N, temp).total;
}

int wsis_voxelize_idx_map(const int64_t* d_coords, int64_t N, int32_t* d_p2v, int32_t* d_counts2, void* d_ws,
                          int64_t ws_bytes, void* stream) {
  WSIS_REQUIRE(N >= 0 && d_counts2, "bad args");
  WSIS_REQUIRE(N < ((int64_t)1 << 30), "N too large for int32 maps");
  hipStream_t st = as_stream(stream);
  WSIS_HIP_CHECK(hipMemsetAsync(d_counts2, 0, 2 * sizeof(int32_t), st));
  if (N == 0) return WSIS_OK;
  WSIS_REQUIRE(d_coords && d_p2v && d_ws, "null pointer");
  size_t temp_bytes = 0;
  WSIS_REQUIRE(vi_temp_bytes(N, &temp_bytes) == 0, "rocprim size query failed");
  const ViLayout L = vi_layout(N, temp_bytes);
  WSIS_REQUIRE((int64_t)L.total <= ws_bytes, "workspace too small");
  char* ws = static_cast<char*>(d_ws);
  int32_t* slot = reinterpret_cast<int32_t*>(ws + L.slot);
  int32_t* first = reinterpret_cast<int32_t*>(ws + L.first);
  int32_t* is_first = reinterpret_cast<int32_t*>(ws + L.is_first);
  int32_t* rank = reinterpret_cast<int32_t*>(ws + L.rank);
  int32_t* vox_count = reinterpret_cast<int32_t*>(ws + L.vox_count);
  uint32_t* keys = reinterpret_cast<uint32_t*>(ws + L.keys);
  uint32_t* keys_sorted = reinterpret_cast<uint32_t*>(ws + L.keys_sorted);
  int32_t* iota = reinterpret_cast<int32_t*>(ws + L.iota);
  int32_t* sorted_p = reinterpret_cast<int32_t*>(ws + L.sorted_p);
  void* temp = ws + L.temp;
  const int g = grid_for(N, 256);
  WSIS_HIP_CHECK(hipMemsetAsync(slot, 0xFF, (size_t)L.cap * 4, st));
  WSIS_HIP_CHECK(hipMemsetAsync(vox_count, 0, (size_t)N * 4, st));
  hipLaunchKernelGGL(vi_insert_kernel, dim3(g), dim3(256), 0, st, d_coords, N, slot, (uint64_t)(L.cap - 1));
  WSIS_LAUNCH_CHECK();
  hipLaunchKernelGGL(vi_first_kernel, dim3(g), dim3(256), 0, st, d_coords, N, slot, (uint64_t)(L.cap - 1), first,
                     is_first);
  WSIS_LAUNCH_CHECK();
  size_t tb = temp_bytes;
  WSIS_HIP_CHECK(rocprim::exclusive_scan(temp, tb, is_first, rank, 0, (size_t)N, rocprim::plus<int32_t>(), st));
  hipLaunchKernelGGL(vi_p2v_kernel, dim3(g), dim3(256), 0, st, first, rank, N, d_p2v, keys, iota, vox_count);
  WSIS_LAUNCH_CHECK();
  hipLaunchKernelGGL(vi_summary_kernel, dim3(g), dim3(256), 0, st, is_first, rank, vox_count, N, d_counts2);
  WSIS_LAUNCH_CHECK();
  tb = temp_bytes;
  WSIS_HIP_CHECK(rocprim::radix_sort_pairs(temp, tb, keys, keys_sorted, iota, sorted_p, (size_t)N, 0, 32, st));
  return WSIS_OK;
}

int wsis_voxelize_idx_fill(const int64_t* d_coords, int64_t N, int64_t M, int32_t max_active,
                           int64_t* d_voxel_locs, int32_t* d_v2p, void* d_ws, int64_t ws_bytes, void* stream) {
  WSIS_REQUIRE(N >= 0 && M >= 0 && max_active >= 0, "bad args");
  if (N == 0 || M == 0) return WSIS_OK;
  WSIS_REQUIRE(d_coords && d_voxel_locs && d_v2p && d_ws, "null pointer");
  size_t temp_bytes = 0;
  WSIS_REQUIRE(vi_temp_bytes(N, &temp_bytes) == 0, "rocprim size query failed");
  const ViLayout L = vi_layout(N, temp_bytes);
  WSIS_REQUIRE((int64_t)L.total <= ws_bytes, "workspace too small");
  char* ws = static_cast<char*>(d_ws);
  const uint32_t* keys_sorted = reinterpret_cast<const uint32_t*>(ws + L.keys_sorted);
  const int32_t* sorted_p = reinterpret_cast<const int32_t*>(ws + L.sorted_p);
  hipStream_t st = as_stream(stream);
  const int32_t stride = 1 + max_active;
  WSIS_HIP_CHECK(hipMemsetAsync(d_v2p, 0, sizeof(int32_t) * (size_t)M * stride, st));
  hipLaunchKernelGGL(vi_fill_kernel, dim3(grid_for(N, 256)), dim3(256), 0, st, d_coords, keys_sorted, sorted_p, N,
                     stride, d_voxel_locs, d_v2p);
  WSIS_LAUNCH_CHECK();
  return WSIS_OK;
}

}  // extern "C"
