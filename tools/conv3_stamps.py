"""Per-wave phase times of the persistent conv kernel (DIAG build of spconv_fwd3_kernel):
   python tools/conv3_stamps.py [level] [slabs]"""
import importlib, sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); importlib.import_module("3d-wsis_amd")
import numpy as np, torch, harness, wsis_native as _n
from spconv import ops
level = int(sys.argv[1]) if len(sys.argv) > 1 else 1
zs = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = 'cuda:0'
b = harness.collate([harness.make_scene(1)])
idx = b['voxel_locs'].int().to(dev).contiguous(); shape = [int(s) for s in b['spatial_shape']]
for l in range(level):
    rd = ops.build_down_rulebook(idx, shape, [2]*3, [2]*3, [0]*3); idx, shape = rd.out_indices, rd.out_shape
rb = ops.build_subm_rulebook(idx, shape, [3]*3, [1]*3)
C = 32 * (level + 1); M = idx.shape[0]
X = torch.randn(M, C, device=dev); W = torch.randn(27, C, C, device=dev) * 0.05
WT = ops._weight_t(W, 0)
out = torch.empty(zs * M, C, device=dev)
dbg = torch.zeros(1 << 20, dtype=torch.int64, device=dev)
lib = _n.hip()
fn = lib.wsis_debug_spconv2_diag
fn.restype = ctypes.c_int32
fn.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int64] + [ctypes.c_int32] * 4 + [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
for it in range(3):
    if it == 2: s.record()
    _n.check(fn(X.data_ptr(), rb.nbr_p.data_ptr(), rb.order.data_ptr(), WT.data_ptr(), out.data_ptr(), M, 27, C, C,
                100 + zs, dbg.data_ptr(), _n.sync_block().data_ptr(), _n.stream_ptr()), "diag")
e.record(); torch.cuda.synchronize()
n_sl = (M + 31) // 32
P = max(1, min(n_sl, 768 // ((C // 32) * zs)))
nw = P * (C // 32) * zs * 4
d = dbg.cpu().numpy()[: nw * 8].reshape(-1, 8).astype(np.float64)
tot, pro, steps, wait, epi, nst, nsl = d[:, 0], d[:, 1], d[:, 2], d[:, 3], d[:, 4], d[:, 5], d[:, 6]
print(f"level {level}, {zs} slab(s): {n_sl} slices, {P} workgroups per block x {C // 32} blocks, kernel {s.elapsed_time(e) * 1e3:.1f} us")
pr = lambda name, v: print(f"  {name:28s} p10 {np.percentile(v, 10):9.0f} p50 {np.median(v):9.0f} p90 {np.percentile(v, 90):9.0f} max {v.max():9.0f}")
pr("lifetime (cycles)", tot); pr("prologue", pro); pr("steps (incl. chain waits)", steps); pr("wait at the slice barrier", wait); pr("epilogue", epi)
pr("steps per wave", nst); pr("slices per wave", nsl); pr("cycles per step", steps / np.maximum(nst, 1))
print(f"  share of the wave time: prologue {pro.sum() / tot.sum():.2f} steps {steps.sum() / tot.sum():.2f} barrier wait {wait.sum() / tot.sum():.2f} epilogue {epi.sum() / tot.sum():.2f}")
