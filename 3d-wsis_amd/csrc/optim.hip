// AdamW over a table of parameter segments in ONE launch (train_scannetv2.py:251 `optimizer.step()`, optimizer from
// config/ScanNet_v2_3D_WSIS.yaml:58-61: AdamW, lr 1e-3, weight_decay 1e-4).  HBM bound: reads p, g, m, v and writes
// p, m, v = 28 bytes per parameter (11.1 M parameters -> 311 MB).  torch's multi-tensor fused AdamW hands 65,536
// elements to a 512-thread block: the 361 tensors of this model become ~85 blocks per launch in 6 launches (a third of
// the CUs, 1.3 TB/s, 233 us); here a workgroup takes 1,024 elements, every tensor in the same launch.
#include "common.h"

namespace wsis {
namespace {

constexpr int AD_CHUNK = 1024;   // elements per workgroup (256 threads x float4)

struct AdamSeg {
  float* p;
  const float* g;
  float* m;
  float* v;
  int64_t n;            // 0: parameter without a gradient this step -> untouched (torch skips it too)
  float step_size;      // lr / (1 - beta1^t), t = number of updates of THIS parameter (torch counts per parameter)
  float inv_sqrt_bc2;   // 1 / sqrt(1 - beta2^t)
  float g_clamp;        // > 0: the gradient is clamped to [-g_clamp, g_clamp] first AND written back (the reference's
  float pad_;           //      `p.grad.data.clamp_(-1, 1)` over the ECC parameters, train_scannetv2.py:247-249, without
};                      //      its 20 small launches in front of the optimizer)

// decay = 1 - lr * weight_decay, omb1 = 1 - beta1, omb2 = 1 - beta2: formed in double on the host and rounded once,
// as torch does (1.0f - 0.999f is off by 5e-5 relative)
__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float decay, float omb1, float b2,
                                         float eps, float omb2, float step_size, float inv_sqrt_bc2) {
  p = p * decay;
  m = m + (g - m) * omb1;                        // lerp form, as torch
  v = b2 * v + omb2 * (g * g);
  const float denom = sqrtf(v) * inv_sqrt_bc2 + eps;
  p = p - step_size * (m / denom);
}

// p.grad.data.clamp_(-gc, gc) of train_scannetv2.py:246-248: torch's clamp propagates NaN (fminf / fmaxf would
// return the bound and hide a diverged branch)
__device__ __forceinline__ float clamp_keep_nan(float g, float gc) { return g != g ? g : fminf(fmaxf(g, -gc), gc); }

__global__ __launch_bounds__(256) void adamw_kernel(const AdamSeg* __restrict__ segs,
                                                    const int32_t* __restrict__ blocks, float lr, float b1, float b2,
                                                    float eps, float wd) {   // lr = decay, b1 = 1-beta1, wd = 1-beta2
  const int s = blocks[2 * blockIdx.x], chunk = blocks[2 * blockIdx.x + 1];
  const AdamSeg sg = segs[s];
  const float step_size = sg.step_size, inv_sqrt_bc2 = sg.inv_sqrt_bc2;
  const float gc = sg.g_clamp;
  float* const gw = const_cast<float*>(sg.g);
  const int64_t i0 = (int64_t)chunk * AD_CHUNK + (int64_t)threadIdx.x * 4;
  if (i0 >= sg.n) return;
  const bool vec = i0 + 4 <= sg.n && ((reinterpret_cast<uintptr_t>(sg.p + i0) | reinterpret_cast<uintptr_t>(sg.g + i0) |
                                       reinterpret_cast<uintptr_t>(sg.m + i0) | reinterpret_cast<uintptr_t>(sg.v + i0)) & 15) == 0;
  if (vec) {
    float4 p = *reinterpret_cast<float4*>(sg.p + i0);
    float4 g = *reinterpret_cast<const float4*>(sg.g + i0);
    if (gc > 0.0f) {
      g.x = clamp_keep_nan(g.x, gc);
      g.y = clamp_keep_nan(g.y, gc);
      g.z = clamp_keep_nan(g.z, gc);
      g.w = clamp_keep_nan(g.w, gc);
      *reinterpret_cast<float4*>(gw + i0) = g;
    }
    float4 m = *reinterpret_cast<float4*>(sg.m + i0);
    float4 v = *reinterpret_cast<float4*>(sg.v + i0);
    adam_one(p.x, g.x, m.x, v.x, lr, b1, b2, eps, wd, step_size, inv_sqrt_bc2);
    adam_one(p.y, g.y, m.y, v.y, lr, b1, b2, eps, wd, step_size, inv_sqrt_bc2);
    adam_one(p.z, g.z, m.z, v.z, lr, b1, b2, eps, wd, step_size, inv_sqrt_bc2);
    adam_one(p.w, g.w, m.w, v.w, lr, b1, b2, eps, wd, step_size, inv_sqrt_bc2);
    *reinterpret_cast<float4*>(sg.p + i0) = p;
    *reinterpret_cast<float4*>(sg.m + i0) = m;
    *reinterpret_cast<float4*>(sg.v + i0) = v;
  } else {
    for (int64_t i = i0; i < i0 + 4 && i < sg.n; ++i) {
      float p = sg.p[i], m = sg.m[i], v = sg.v[i];
      float g = sg.g[i];
      if (gc > 0.0f) {
        g = clamp_keep_nan(g, gc);
        gw[i] = g;
      }
      adam_one(p, g, m, v, lr, b1, b2, eps, wd, step_size, inv_sqrt_bc2);
      sg.p[i] = p;
      sg.m[i] = m;
      sg.v[i] = v;
    }
  }
}

}  // namespace
}  // namespace wsis

using namespace wsis;

extern "C" {

int32_t wsis_adamw_segment_bytes(void) { return (int32_t)sizeof(AdamSeg); }
int32_t wsis_adamw_chunk(void) { return AD_CHUNK; }

int wsis_adamw_step(const void* d_segments, const int32_t* d_blocks, int64_t n_blocks, double lr, double beta1,
                    double beta2, double eps, double weight_decay, void* stream) {
  WSIS_REQUIRE(n_blocks >= 0, "bad args");
  if (n_blocks == 0) return WSIS_OK;
  WSIS_REQUIRE(d_segments && d_blocks, "null pointer");
  WSIS_REQUIRE(n_blocks < ((int64_t)1 << 31), "too many blocks");
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)n_blocks), dim3(256), 0, as_stream(stream),
                     static_cast<const AdamSeg*>(d_segments), d_blocks, (float)(1.0 - lr * weight_decay),
                     (float)(1.0 - beta1), (float)beta2, (float)eps, (float)(1.0 - beta2));
  WSIS_LAUNCH_CHECK();
  return WSIS_OK;
}

}  // extern "C"
