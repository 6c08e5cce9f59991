// Phase table of the resident deep-level kernel (deep.hip), built by the executor (executor.hip).
#pragma once
#include <cstddef>
#include <cstdint>

#include <hip/hip_runtime.h>

namespace wsis {

enum DeepKind : int32_t { DK_CONV = 1, DK_REDUCE = 2, DK_BN_FWD = 3, DK_BN_BWD = 4, DK_CAT = 5, DK_SPLIT = 6 };

// one phase (what used to be one launch); plain pointers, read by every workgroup through the scalar cache.
//   DK_CONV    p0 X  p1 nbr  p2 order  p3 W^T  p4 bias  p5 residual  p6 out  p7 slab partials (ZS > 1)  p8 slice partials
//              (BatchNorm statistics, or with p9 != 0 the backward sums)  p9 bn x  p10 mean  p11 var  p12 gamma  p13 beta
//   DK_REDUCE  p4 bias  p5 residual  p6 out  p7 slab partials  p8 slice partials  p9..p13 as above
//   DK_BN_FWD  p0 x  p1 gamma  p2 beta  p3 running mean  p4 running var  p5 partials  p6 partials of the second producer
//              (concatenation; C0 = channels of the first)  p7 y (0: statistics only)  p8 mean  p9 var
//   DK_BN_BWD  p0 x  p1 dy  p2 mean  p3 var  p4 gamma  p5 beta  p6 addend  p7 partials  p8 dx  p9 dgamma  p10 dbeta
//   DK_CAT     p0 a [M, Cin]  p1 b [M, Cout]  p2 out         DK_SPLIT  p0 in  p1 a  p2 b
struct DeepOp {
  int32_t kind, NW, ZS, relu;
  int32_t K, Cin, Cout, flip;
  int64_t M_in, M_out;
  uint32_t x_bytes;
  int32_t C0;
  float eps, momentum;
  int32_t training, stamp;      // stamp: profile stamp written when the phase is complete (-1: none); stamp 0 = launch start
  const void* p[16];
};
static_assert(sizeof(DeepOp) == 200, "phase record layout");

bool deep_enabled();
int64_t deep_max_rows();
// launch plan of a product inside the resident kernel = the one-shot kernel's (same order of additions); false: not eligible
bool deep_conv_plan(int64_t M_out, int K, int Cin, int Cout, int* NW, int* ZS);
bool deep_bn_rows_ok(int64_t M);      // rows a BatchNorm phase can take (two-level finish of <= 16 chunks)
constexpr size_t kDeepSyncBytes = 4096;
int deep_launch(const DeepOp* h_ops, int n, void* d_table, void* d_sync, unsigned long long* d_stamps, hipStream_t st);

}  // namespace wsis
