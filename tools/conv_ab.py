"""Per-layer A/B of the forward / dIn product (wsis_spconv_fwd_t) under environment settings that are read per call:
   python tools/conv_ab.py "WSIS_FWD2_DA_NW4=2" "WSIS_FWD2_DA_NW4=3" ...        (first setting = baseline)
For every UNet layer shape of the C2 scene (CONV2_SCENES=4: the 4-scene batch): us per launch under each setting,
max |out - baseline out| (0 expected: the order of additions of an output row depends on the offset index alone) and the
error of the baseline against an fp64 gather-GEMM on sampled rows; last line: the estimated conv time per step."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); importlib.import_module("3d-wsis_amd")
import torch, harness
from spconv import ops

dev = 'cuda:0'
settings = [dict(kv.split("=") for kv in a.split(",") if kv) for a in sys.argv[1:]] or [{}]
keys = sorted({k for s in settings for k in s})


def apply(s):
    for k in keys:
        if k in s:
            os.environ[k] = s[k]
        else:
            os.environ.pop(k, None)


def timeit(f, n=40):
    for _ in range(3): f()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3


def ref_rows(X, nbr, W, rows):
    Xd, Wd = X.double(), W.double()
    out = torch.zeros(len(rows), W.shape[2], dtype=torch.float64, device=X.device)
    for k in range(W.shape[0]):
        g = nbr[k][rows].long() if nbr is not None else rows
        ok = g >= 0
        out[ok] += Xd[g[ok]] @ Wd[k]
    return out


ns = int(os.environ.get("CONV2_SCENES", "1"))
b = harness.collate([harness.make_scene(1 + i) for i in range(ns)])
idx = b['voxel_locs'].int().to(dev).contiguous(); shape = [int(s) for s in b['spatial_shape']]
planes = [32, 64, 96, 128, 160]
g = torch.Generator(device=dev); g.manual_seed(7)
tot = [0.0] * len(settings)
# products per step (forward + dIn) of each shape: block_reps = 2 residual blocks of two convs each way
for l in range(5):
    C = planes[l]; M = idx.shape[0]
    rb = ops.build_subm_rulebook(idx, shape, [3] * 3, [1] * 3)
    cases = [("subm", C, C, rb.nbr_p, rb.order, rb.nbr, 27, M, M, (8 if l < 4 else 4) * 2)]
    if l < 4:
        cases.append(("subm", 2 * C, C, rb.nbr_p, rb.order, rb.nbr, 27, M, M, 2))
        cases.append(("1x1 ", 2 * C, C, None, None, None, 1, M, M, 2))
        rd = ops.build_down_rulebook(idx, shape, [2] * 3, [2] * 3, [0] * 3)
        Mo = rd.out_indices.shape[0]
        cases.append(("down", C, planes[l + 1], rd.nbr_p, rd.order, rd.nbr, 8, M, Mo, 2))
        cases.append(("up  ", planes[l + 1], C, rd.nbr_up_p, rd.order_up, rd.nbr_up, 8, Mo, M, 2))
    for name, cin, cout, nbr_p, order, nbr, K, Mi, Mo_, cnt in cases:
        X = torch.randn(Mi, cin, device=dev, generator=g); W = torch.randn(K, cin, cout, device=dev, generator=g) * 0.05
        WT = ops._weight_t(W, 0)
        outs, ts = [], []
        for s in settings:
            apply(s)
            outs.append(ops._conv_t(X, nbr_p, order, WT, 0, None, None, Mo_))
            ts.append(timeit(lambda: ops._conv_t(X, nbr_p, order, WT, 0, None, None, Mo_)))
        rows = torch.randint(0, Mo_, (min(Mo_, 1024),), device=dev, generator=g)
        want = ref_rows(X, nbr, W, rows)
        err = float((outs[0][rows].double() - want).abs().max()) / max(float(want.abs().max()), 1e-30)
        line = f"L{l} {name} {cin:3d}->{cout:3d} K={K:2d} M={Mo_:6d} x{cnt:2d}: " + " | ".join(
            f"{t:6.1f}us d {float((o - outs[0]).abs().max()):.0e}" for t, o in zip(ts, outs)) + f" | err64 {err:.1e}"
        print(line, flush=True)
        for i, t in enumerate(ts):
            tot[i] += t * cnt
    if l < 4:
        idx, shape = rd.out_indices, rd.out_shape
for s, t in zip(settings, tot):
    print(f"{s}: estimated conv fwd + dIn per step {t / 1e3:.3f} ms")
