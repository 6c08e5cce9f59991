"""Where the waves of the ring convolution (csrc/spconv3.hip, DIAG build) spend their cycles:
   python tools/ring_stamps.py [level] [nt] [kind]      kind: subm (default) | down | up | 1x1
Per role: total cycles, cycles inside each kind of wait, steps / items (medians and maxima over the workgroups)."""
import importlib, sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); importlib.import_module("3d-wsis_amd")
import numpy as np, torch, harness, wsis_native as _n
from spconv import ops
level = int(sys.argv[1]) if len(sys.argv) > 1 else 0
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 1
kind = sys.argv[3] if len(sys.argv) > 3 else "subm"
dev = 'cuda:0'
b = harness.collate([harness.make_scene(1 + i) for i in range(int(os.environ.get('CONV2_SCENES', '1')))])
idx = b['voxel_locs'].int().to(dev).contiguous(); shape = [int(s) for s in b['spatial_shape']]
for l in range(level):
    rd = ops.build_down_rulebook(idx, shape, [2]*3, [2]*3, [0]*3); idx, shape = rd.out_indices, rd.out_shape
planes = [32, 64, 96, 128, 160]
C = planes[level]; M = idx.shape[0]
if kind == "subm":
    rb = ops.build_subm_rulebook(idx, shape, [3]*3, [1]*3)
    nbr, order, K, cin, cout, Mi, Mo = rb.nbr_p, rb.order, 27, C, C, M, M
elif kind == "1x1":
    nbr, order, K, cin, cout, Mi, Mo = None, None, 1, 2 * C, C, M, M
else:
    rd = ops.build_down_rulebook(idx, shape, [2]*3, [2]*3, [0]*3)
    Mc = rd.out_indices.shape[0]
    if kind == "down":
        nbr, order, K, cin, cout, Mi, Mo = rd.nbr_p, rd.order, 8, C, planes[level + 1], M, Mc
    else:
        nbr, order, K, cin, cout, Mi, Mo = rd.nbr_up_p, rd.order_up, 8, planes[level + 1], C, Mc, M
X = torch.randn(Mi, cin, device=dev); W = torch.randn(K, cin, cout, device=dev) * 0.05
WT = ops._weight_t(W, 0)
out = torch.empty(Mo, cout, device=dev)
items = (Mo + 31) // 32 * (cout // 32)
grid = min(256, (items + 4 // nt - 1) // (4 // nt))
dbg = torch.zeros(grid * 12 * 8, dtype=torch.int64, device=dev)
lib = _n.hip()
fn = lib.wsis_debug_ring_diag
fn.restype = ctypes.c_int32
fn.argtypes = [ctypes.c_int32] + [ctypes.c_void_p] * 5 + [ctypes.c_int64] * 2 + [ctypes.c_int32] * 3 + [ctypes.c_void_p, ctypes.c_void_p]
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for it in range(4):
    if it == 3: ev[0].record()
    _n.check(fn(nt | (int(os.environ.get('RING_EXP', '0')) << 8), X.data_ptr(), _n.ptr(nbr), _n.ptr(order), WT.data_ptr(), out.data_ptr(), Mi, Mo, K, cin, cout,
                dbg.data_ptr(), _n.stream_ptr()), "diag")
ev[1].record(); torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(grid, 12, 8).astype(np.float64)
print(f"level {level} {kind} {cin}->{cout} K={K} M_out={Mo} nt={nt}: {items} items, grid {grid}, launch {ev[0].elapsed_time(ev[1]) * 1e3:.1f} us (DIAG build)")
def row(name, a, cols):
    tot = a[..., 0].reshape(-1); live = tot > 0
    s = f"{name:9s} n={int(live.sum()):4d} total p50 {np.median(tot[live]):7.0f} max {tot[live].max():7.0f} cyc |"
    for j, lab in cols:
        v = a[..., j].reshape(-1)[live]
        s += f" {lab} p50 {np.median(v):7.0f} max {v.max():7.0f} |"
    print(s)
row("consumer", d[:, 0:4], [(1, "wait_prod"), (6, "prod stalls"), (2, "wait_mask"), (3, "7-MFMA block"), (7, "wait_B"), (4, "steps"), (5, "items")])
row("loader", d[:, 4:8], [(1, "step passes"), (7, "ring-full passes"), (6, "max vm out"), (5, "steps")])
row("helper", d[:, 8:12], [(1, "header cyc"), (2, "finish cyc"), (3, "items finished")])
c = d[:, 0:4]; live = c[..., 0] > 0
steps = c[..., 4][live]; tot = c[..., 0][live]; wp = c[..., 1][live]
print(f"consumer cycles per step: p50 {np.median(tot / np.maximum(steps, 1)):.0f}; busy (total - waits) per step p50 "
      f"{np.median((tot - wp - c[..., 2][live] - c[..., 3][live]) / np.maximum(steps, 1)):.0f}; steps per consumer p50 {np.median(steps):.0f} max {steps.max():.0f}")
