"""Per-wave timeline of the weight-gradient kernel (DIAG build of csrc/spconv_dw2.hip: cycle stamps per wave):
   python tools/dw2_stamps.py [level]"""
import importlib, sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); importlib.import_module("3d-wsis_amd")
import numpy as np, torch, harness, wsis_native as _n
from spconv import ops
level = int(sys.argv[1]) if len(sys.argv) > 1 else 0
dev = 'cuda:0'
b = harness.collate([harness.make_scene(1)])
idx = b['voxel_locs'].int().to(dev).contiguous(); shape = [int(s) for s in b['spatial_shape']]
for l in range(level):
    rd = ops.build_down_rulebook(idx, shape, [2]*3, [2]*3, [0]*3); idx, shape = rd.out_indices, rd.out_shape
rb = ops.build_subm_rulebook(idx, shape, [3]*3, [1]*3)
C = 32 * (level + 1); M = idx.shape[0]
X = torch.randn(M, C, device=dev); dY = torch.randn(M, C, device=dev)
lib = _n.hip()
ws = torch.empty(lib.wsis_spconv_dw_workspace_bytes(M, 27, C, C), dtype=torch.uint8, device=dev)
dbg = torch.zeros(1 << 20, dtype=torch.int64, device=dev)
fn = lib.wsis_debug_dw2_diag
fn.restype = ctypes.c_int32
fn.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int64] * 2 + [ctypes.c_int32] * 3 + [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                                                                 ctypes.POINTER(ctypes.c_int64), ctypes.c_void_p]
nw = ctypes.c_int64(0)
s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
for it in range(3):
    if it == 2: s.record()
    _n.check(fn(X.data_ptr(), rb.nbr_p.data_ptr(), rb.order.data_ptr(), dY.data_ptr(), M, M, 27, C, C, ws.data_ptr(),
                dbg.data_ptr(), dbg.numel() * 8, ctypes.byref(nw), _n.stream_ptr()), "diag")
e.record(); torch.cuda.synchronize()
us = s.elapsed_time(e) * 1e3
d = dbg.cpu().numpy()[: nw.value * 10].reshape(-1, 10).astype(np.float64)
st, lp, le, en, steps, sl, dwait, dtop, dchain, dbot = d.T
print(f"level {level}: {nw.value} waves, kernel {us:.1f} us (= {us * 2400:.0f} cycles at 2.4 GHz); per-wave phases in counter ticks:")
pr = lambda name, v: print(f"  {name:22s} p10 {np.percentile(v, 10):9.1f} p50 {np.median(v):9.1f} p90 {np.percentile(v, 90):9.1f} max {v.max():9.1f}")
run = lp > 0
pr("prologue", (lp - st)[run]); pr("loop", (le - lp)[run]); pr("epilogue", (en - le)[run]); pr("lifetime", en - st)
pr("steps", steps); pr("slices", sl); pr("loop ticks / step", ((le - lp) / np.maximum(steps, 1))[run])
for nm, v in (("  wait for the tile", dwait), ("  top (frags, prepare)", dtop), ("  chain (16 MFMA + issue)", dchain), ("  bottom + control", dbot)):
    pr(nm + " / step", (v / np.maximum(steps, 1))[run])
print(f"  total steps {steps.sum():.0f} ({steps.sum() * 16:.0f} MFMAs), active (slice,offset) fraction {steps.sum() / (sl.sum() * 27 / 4 + 1e-9):.2f}")
