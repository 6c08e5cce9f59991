// libwsis_hip.so: error plumbing, voxelization fwd/bwd (a2), row gather (a14).
#include <cstdarg>

#include "common.h"

namespace wsis {
std::string& err_slot() {
  thread_local std::string s;
  return s;
}
int fail(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  err_slot() = buf;
  return code;
}
}  // namespace wsis

using namespace wsis;

// ---- a2: voxelization (mean pool of the points of each voxel) ------------------------------
// One thread per (voxel, channel); the n_m <= max_active points are summed in list order in
// fp32, exactly like the upstream kernel, so the result is order-defined.
// HBM-bound: N*C*4 read + M*C*4 write + v2p bytes.
__global__ void voxelize_fwd_kernel(const float* __restrict__ feats, const int32_t* __restrict__ v2p,
                                    float* __restrict__ out, int64_t M, int C, int stride, int mode) {
  const int64_t total = M * C;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const RowCol rc = row_col(t, C, total);
    const int64_t m = rc.row;
    const int c = rc.col;
    const int32_t* row = v2p + m * stride;
    const int n = row[0];
    const float w = (mode == 4 && n > 0) ? 1.0f / (float)n : 1.0f;
    float acc = 0.0f;
    for (int i = 0; i < n; ++i) acc += w * feats[(int64_t)row[1 + i] * C + c];
    out[t] = acc;
  }
}

__global__ void voxelize_bwd_kernel(const float* __restrict__ dout, const int32_t* __restrict__ v2p,
                                    float* __restrict__ dfeats, int64_t M, int C, int stride, int mode) {
  const int64_t total = M * C;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const RowCol rc = row_col(t, C, total);
    const int64_t m = rc.row;
    const int c = rc.col;
    const int32_t* row = v2p + m * stride;
    const int n = row[0];
    const float w = (mode == 4 && n > 0) ? 1.0f / (float)n : 1.0f;
    const float g = w * dout[t];
    // each point belongs to exactly one voxel: plain stores, no contention
    for (int i = 0; i < n; ++i) dfeats[(int64_t)row[1 + i] * C + c] += g;
  }
}

// ---- a14: row gather -------------------------------------------------------------------------
template <typename IdxT>
__global__ void gather_rows_kernel(const float* __restrict__ src, const IdxT* __restrict__ idx,
                                   float* __restrict__ out, int64_t N, int C) {
  if ((C & 3) == 0) {
    const int C4 = C >> 2;
    const int64_t total = N * C4;
    const float4* s4 = reinterpret_cast<const float4*>(src);
    float4* o4 = reinterpret_cast<float4*>(out);
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
      const RowCol rc = row_col(t, C4, total);
      const int64_t p = rc.row;
      const int c = rc.col;
      o4[t] = s4[(int64_t)idx[p] * C4 + c];
    }
  } else {
    const int64_t total = N * C;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
      const RowCol rc = row_col(t, C, total);
      const int64_t p = rc.row;
      const int c = rc.col;
      out[t] = src[(int64_t)idx[p] * C + c];
    }
  }
}

extern "C" {

int wsis_version(void) { return WSIS_ABI_VERSION; }
const char* wsis_last_error(void) { return err_slot().c_str(); }
int wsis_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int wsis_voxelize_fwd(const float* d_feats, const int32_t* d_v2p, float* d_out, int64_t M, int32_t C,
                      int32_t stride, int32_t mode, void* stream) {
  WSIS_REQUIRE(M >= 0 && C > 0 && stride >= 1, "bad sizes");
  if (M == 0) return WSIS_OK;
  WSIS_REQUIRE(d_feats && d_v2p && d_out, "null pointer");
  const int block = 256;
  hipLaunchKernelGGL(voxelize_fwd_kernel, dim3(grid_for(M * C, block)), dim3(block), 0, as_stream(stream),
                     d_feats, d_v2p, d_out, M, C, stride, mode);
  WSIS_LAUNCH_CHECK();
  return WSIS_OK;
}

int wsis_voxelize_bwd(const float* d_dout, const int32_t* d_v2p, float* d_dfeats, int64_t M, int32_t C,
                      int32_t stride, int32_t mode, void* stream) {
  WSIS_REQUIRE(M >= 0 && C > 0 && stride >= 1, "bad sizes");
  if (M == 0) return WSIS_OK;
  WSIS_REQUIRE(d_dout && d_v2p && d_dfeats, "null pointer");
  const int block = 256;
  hipLaunchKernelGGL(voxelize_bwd_kernel, dim3(grid_for(M * C, block)), dim3(block), 0, as_stream(stream),
                     d_dout, d_v2p, d_dfeats, M, C, stride, mode);
  WSIS_LAUNCH_CHECK();
  return WSIS_OK;
}

int wsis_gather_rows(const float* d_src, const void* d_idx, int32_t idx_is_64, float* d_out, int64_t N,
                     int32_t C, void* stream) {
  WSIS_REQUIRE(N >= 0 && C > 0, "bad sizes");
  if (N == 0) return WSIS_OK;
  WSIS_REQUIRE(d_src && d_idx && d_out, "null pointer");
  const int block = 256;
  const int64_t work = (C & 3) == 0 ? N * (C >> 2) : N * C;
  if (idx_is_64)
    hipLaunchKernelGGL(gather_rows_kernel<int64_t>, dim3(grid_for(work, block)), dim3(block), 0,
                       as_stream(stream), d_src, (const int64_t*)d_idx, d_out, N, C);
  else
    hipLaunchKernelGGL(gather_rows_kernel<int32_t>, dim3(grid_for(work, block)), dim3(block), 0,
                       as_stream(stream), d_src, (const int32_t*)d_idx, d_out, N, C);
  WSIS_LAUNCH_CHECK();
  return WSIS_OK;
}

}  // extern "C"

// ---- diagnostic (not part of the ABI header): what a dependent kernel boundary costs on this stream for kernels of
// different shapes -- n back-to-back launches of an (almost) empty kernel; variant bits: 1 = 32 KB of dynamic LDS,
// 2 = 64-thread workgroups x 4800 instead of 256 x 256, 4 = alternate LDS / no LDS, 8 = 40 bytes of LDS-free kernel
// arguments replaced by a 160-byte struct.  tools/gap_probe.py times the chain with events.
namespace {
struct GapArgs {
  float* p;
  long long pad[19];
};
__global__ void gap_probe_kernel(float* p) {
  extern __shared__ float sm[];
  if (threadIdx.x == 0 && blockIdx.x == 0xffffff) p[0] = sm[0];
}
__global__ void gap_probe_big_kernel(GapArgs a) {
  extern __shared__ float sm[];
  if (threadIdx.x == 0 && blockIdx.x == 0xffffff) a.p[0] = sm[0] + (float)a.pad[3];
}
}  // namespace

extern "C" int32_t wsis_experimental(void) { return WSIS_EXPERIMENTAL ? 1 : 0; }

extern "C" int wsis_debug_gap_probe(int variant, int n, float* d_buf, void* stream) {
  hipStream_t st = wsis::as_stream(stream);
  const bool small = (variant & 2) != 0;
  const dim3 grid(small ? 4800 : 256), block(small ? 64 : 256);
  static bool attr = false;
  if (!attr) {
    WSIS_HIP_CHECK(hipFuncSetAttribute((const void*)gap_probe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    WSIS_HIP_CHECK(hipFuncSetAttribute((const void*)gap_probe_big_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    attr = true;
  }
  for (int i = 0; i < n; ++i) {
    size_t lds = (variant & 1) ? 32768 : 0;
    if ((variant & 4) && (i & 1)) lds = 0;
    if (variant & 16) {      // the launch form of the conv / BatchNorm entry points (event pair, null outside profiling)
      if (variant & 8) {
        GapArgs a{};
        a.p = d_buf;
        hipExtLaunchKernelGGL(gap_probe_big_kernel, grid, block, lds, st, nullptr, nullptr, 0u, a);
      } else {
        hipExtLaunchKernelGGL(gap_probe_kernel, grid, block, lds, st, nullptr, nullptr, 0u, d_buf);
      }
    } else if (variant & 8) {
      GapArgs a{};
      a.p = d_buf;
      hipLaunchKernelGGL(gap_probe_big_kernel, grid, block, lds, st, a);
    } else {
      hipLaunchKernelGGL(gap_probe_kernel, grid, block, lds, st, d_buf);
    }
  }
  WSIS_LAUNCH_CHECK();
  return WSIS_OK;
}
