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
if os.environ.get('STAMPS_DUMP'):
    np.save(os.environ['STAMPS_DUMP'], dbg.cpu().numpy().reshape(-1, 8))      # raw stamps for offline models
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

# ---- per compute unit: how the dispatcher spread the work items (HW_ID: cu_id 11:8, sh_id 12, se_id 15:13; XCC_ID 3:0)
hw, xcc = d[:, 6].astype(np.int64), d[:, 7].astype(np.int64) & 15
cu_key = (xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15)
keys, inv = np.unique(cu_key, return_inverse=True)
n_cu = len(keys)
wg_per = np.bincount(inv, minlength=n_cu)
steps_per = np.bincount(inv, weights=T, minlength=n_cu)
end_per = np.zeros(n_cu); np.maximum.at(end_per, inv, end_us)
first_wave_steps = np.zeros(n_cu)
print(f"compute units seen: {n_cu}; workgroups per CU min {wg_per.min()} p50 {int(np.median(wg_per))} max {wg_per.max()}")
print(f"steps (wave 0) per CU: min {steps_per.min():.0f} p10 {np.percentile(steps_per, 10):.0f} p50 {np.median(steps_per):.0f} "
      f"p90 {np.percentile(steps_per, 90):.0f} max {steps_per.max():.0f}  (mean {steps_per.mean():.1f})")
print(f"last end per CU us: min {end_per.min():.1f} p10 {np.percentile(end_per, 10):.1f} p50 {np.median(end_per):.1f} p90 {np.percentile(end_per, 90):.1f} max {end_per.max():.1f}")
cc = np.corrcoef(steps_per, end_per)[0, 1]
print(f"correlation(steps per CU, last end per CU) = {cc:.2f}")
order = np.argsort(-end_per)[:6]
for j in order:
    m = inv == j
    print(f"  CU {keys[j]:#06x}: {m.sum()} wgs, steps {T[m].astype(int).tolist()[:16]}, starts {np.round(start_us[m], 1).tolist()[:16]}, last end {end_per[j]:.1f}")
# the first 3 workgroups by index: which CU
print("CU of workgroups 0..15:", [hex(int(k)) for k in cu_key[:16]])

# ---- does workgroup i go to the CU of workgroup i - 256?  and what would a balanced deal of these items give
idx_all = np.arange(len(cu_key))
same = np.mean(cu_key[256:] == cu_key[:-256]) if len(cu_key) > 256 else float('nan')
print(f"fraction of workgroups i >= 256 on the CU of workgroup i - 256: {same:.2f}")
for j in order[:3]:
    print(f"  CU {keys[j]:#06x}: workgroup ids {idx_all[inv == j].tolist()[:24]}")
Ts = np.sort(T)[::-1]
def deal(seq, n=256, snake=True):
    load = np.zeros(n)
    for i, w in enumerate(seq):
        b, c = divmod(i, n)
        load[(n - 1 - c) if (snake and b & 1) else c] += w
    return load
def lpt(seq, n=256):
    load = np.zeros(n)
    for w in seq:
        load[np.argmin(load)] += w
    return load
print(f"max steps per CU: measured {steps_per.max():.0f} | dealt i % 256 in launch order {deal(T, snake=False).max():.0f} | "
      f"sorted + snake {deal(Ts).max():.0f} | sorted + least-loaded (LPT) {lpt(Ts).max():.0f} | mean {T.sum() / 256:.1f}")
