"""Per-wave timeline of the wave-autonomous conv kernel (DIAG build: s_memtime / s_memrealtime stamps per workgroup):
   python tools/conv2_stamps.py [level] [variant]      variant 0: B direct / ring 2, 1: both rings / ring 3,
   n >= 2: 4 waves per work item and n - 1 offset slabs, 100 + z: the persistent form with z slabs;
   + 256: weights loaded for the first step only, + 512: every gathered row is row 0 (what the operand streams cost)"""
import importlib, sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); importlib.import_module("3d-wsis_amd")
import numpy as np, torch, harness, wsis_native as _n
from spconv import ops
level = int(sys.argv[1]) if len(sys.argv) > 1 else 0
variant = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = 'cuda:0'
b = harness.collate([harness.make_scene(1 + i) for i in range(int(os.environ.get('CONV2_SCENES', '1')))])
idx = b['voxel_locs'].int().to(dev).contiguous(); shape = [int(s) for s in b['spatial_shape']]
for l in range(level):
    rd = ops.build_down_rulebook(idx, shape, [2]*3, [2]*3, [0]*3); idx, shape = rd.out_indices, rd.out_shape
rb = ops.build_subm_rulebook(idx, shape, [3]*3, [1]*3)
C = 32 * (level + 1); M = idx.shape[0]
X = torch.randn(M, C, device=dev); W = torch.randn(27, C, C, device=dev) * 0.05
WT = ops._weight_t(W, 0)
n_wg = (M + 31) // 32 * (C // 32)
out = torch.empty(max((variant & 0xff) - 1, 1) * M, C, device=dev)   # variant >= 2: the slabs land here
dbg = torch.zeros(n_wg * 8, dtype=torch.int64, device=dev)
lib = _n.hip()
fn = lib.wsis_debug_spconv2_diag
fn.restype = ctypes.c_int32
fn.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int64] + [ctypes.c_int32] * 4 + [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
for _ in range(3):
    _n.check(fn(X.data_ptr(), rb.nbr_p.data_ptr(), rb.order.data_ptr(), WT.data_ptr(), out.data_ptr(), M, 27, C, C,
                variant, dbg.data_ptr(), _n.sync_block().data_ptr(), _n.stream_ptr()), "diag")
torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(-1, 8).astype(np.float64)
r0, r1 = d[:, 0], d[:, 1]
t_first = r0.min()
start_us, end_us = (r0 - t_first) / 100.0, (r1 - t_first) / 100.0          # 100 MHz
pro, walk, epi, T = d[:, 2], d[:, 3], d[:, 4], d[:, 5]
print(f"level {level} variant {variant}: {n_wg} workgroups, kernel span {end_us.max():.1f} us")
print(f"start time  us: p50 {np.median(start_us):.1f} p90 {np.percentile(start_us, 90):.1f} max {start_us.max():.1f}")
print(f"lifetime    us: mean {(end_us - start_us).mean():.1f} p50 {np.median(end_us - start_us):.1f} max {(end_us - start_us).max():.1f}")
print(f"steps: mean {T.mean():.1f} max {T.max():.0f};  prologue cycles: p50 {np.median(pro):.0f} p90 {np.percentile(pro, 90):.0f};"
      f"  epilogue cycles p50 {np.median(epi):.0f} p90 {np.percentile(epi, 90):.0f}")
per = walk / np.maximum(T, 1)
for lo, hi in ((1, 6), (6, 12), (12, 20), (20, 200)):
    m = (T >= lo) & (T < hi)
    if m.any():
        print(f"  steps {lo:2d}-{hi:3d}: {m.sum():5d} waves, cycles/step p50 {np.median(per[m]):.0f} p90 {np.percentile(per[m], 90):.0f}, "
              f"start p50 {np.median(start_us[m]):.1f} us, end p50 {np.median(end_us[m]):.1f} max {end_us[m].max():.1f} us")
# concurrency over time
for t in np.arange(0, end_us.max(), end_us.max() / 12):
    print(f"  t={t:5.1f} us: {int(((start_us <= t) & (end_us > t)).sum()):5d} waves resident")
for lo, hi in ((0, 1), (1, 15), (15, 22), (22, 1e9)):
    m = (start_us >= lo) & (start_us < hi)
    if m.any():
        print(f"  started {lo:4.0f}-{hi:4.0f} us: {m.sum():5d} wgs, steps p50 {np.median(T[m]):.0f}; us: prologue p50 {np.median(pro[m]) / 2100:.1f} "
              f"p90 {np.percentile(pro[m], 90) / 2100:.1f}, steps p50 {np.median(walk[m]) / 2100:.1f} p90 {np.percentile(walk[m], 90) / 2100:.1f}, "
              f"epilogue p50 {np.median(epi[m]) / 2100:.1f} p90 {np.percentile(epi[m], 90) / 2100:.1f}, lifetime p50 {np.median((end_us - start_us)[m]):.1f}")
late = np.argsort(-end_us)[:8]
print("last finishers: " + ", ".join(f"(wg {i} steps {int(T[i])} start {start_us[i]:.1f} end {end_us[i]:.1f} cyc/step {per[i]:.0f} pro {pro[i] / 2100:.1f} epi {epi[i] / 2100:.1f} us)" for i in late))
