"""Per-layer check + microbenchmark of the wave-autonomous conv kernel (csrc/spconv2.hip) against the round-1 kernel on
the C2 scene's real rulebooks (GPU box):
   python tools/conv2_bench.py [plan ...]     plan = NB,NW,ZS,DA (0 = automatic), e.g. 1,0,0,3
For every UNet layer shape: max |new - old| (bit-identical expected where one wave owns a work item), error of both
against an fp64 gather-GEMM on 2048 sampled rows, us per launch of old / new, algorithmic GB/s."""
import importlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); importlib.import_module("3d-wsis_amd")
import numpy as np, torch, harness
from spconv import ops

dev = 'cuda:0'


def build_levels():
    ns = int(os.environ.get("CONV2_SCENES", "1"))
    b = harness.collate([harness.make_scene(1 + i) for i in range(ns)])
    idx = b['voxel_locs'].int().to(dev).contiguous(); shape = [int(s) for s in b['spatial_shape']]
    levels = []
    cur_idx, cur_shape = idx, shape
    for l in range(5):
        rb = ops.build_subm_rulebook(cur_idx, cur_shape, [3] * 3, [1] * 3)
        ent = {'M': cur_idx.shape[0], 'P': int((rb.nbr >= 0).sum()), 'subm': rb}
        if l < 4:
            rd = ops.build_down_rulebook(cur_idx, cur_shape, [2] * 3, [2] * 3, [0] * 3)
            ent['down'] = rd; cur_idx, cur_shape = rd.out_indices, rd.out_shape
        levels.append(ent)
    return levels


def timeit(f, n=30):
    for _ in range(3): f()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3


def ref_rows(X, nbr, W, rows):
    """fp64 gather-GEMM of the sampled output rows from the UNPACKED table nbr [K, M]"""
    Xd, Wd = X.double(), W.double()
    out = torch.zeros(len(rows), W.shape[2], dtype=torch.float64, device=X.device)
    for k in range(W.shape[0]):
        g = nbr[k][rows].long()
        ok = g >= 0
        out[ok] += Xd[g[ok]] @ Wd[k]
    return out


def run(plan):
    nb, nw, zs, da = plan
    os.environ["WSIS_FWD2_NB"] = str(nb or 1)
    for k, v in (("WSIS_FWD2_NW", nw), ("WSIS_FWD2_ZS", zs)):
        os.environ[k] = str(v)
    os.environ["WSIS_FWD2_DA"] = str(da or 3)
    levels = build_levels()
    planes = [32, 64, 96, 128, 160]
    tot_old = tot_new = 0.0
    g = torch.Generator(device=dev).manual_seed(0)
    for l, ent in enumerate(levels):
        C = planes[l]; rb = ent['subm']; M, P = ent['M'], ent['P']
        shapes = [(C, C, 8 if l < 4 else 4)] + ([(2 * C, C, 1)] if l < 4 else [])
        for (cin, cout, cnt) in shapes:
            X = torch.randn(M, cin, device=dev, generator=g); W = torch.randn(27, cin, cout, device=dev, generator=g) * 0.05
            res = torch.randn(M, cout, device=dev, generator=g)
            WT = ops._weight_t(W, 0)
            old = ops._conv(X, rb.nbr_p, rb.order, W, None, res, M)
            new = ops._conv_t(X, rb.nbr_p, rb.order, WT, 0, None, res, M)
            rows = torch.randint(0, M, (min(M, 2048),), device=dev, generator=g)
            want = ref_rows(X, rb.nbr, W, rows) + res[rows].double()
            scale = float(want.abs().max())
            e_old = float((old[rows].double() - want).abs().max()) / scale
            e_new = float((new[rows].double() - want).abs().max()) / scale
            diff = float((old - new).abs().max())
            # dIn through the same kernel: weight as B^T with flip
            dY = torch.randn(M, cout, device=dev, generator=g)
            dold = ops._conv(dY, rb.nbr_p, rb.order, ops._weight_t(W, 1), None, None, M)
            dnew = ops._conv_t(dY, rb.nbr_p, rb.order, W, 1, None, None, M)
            ddiff = float((dold - dnew).abs().max()) / max(float(dold.abs().max()), 1e-30)
            t_old = timeit(lambda: ops._conv(X, rb.nbr_p, rb.order, W, None, None, M))
            t_new = timeit(lambda: ops._conv_t(X, rb.nbr_p, rb.order, WT, 0, None, None, M))
            by = P * (cin + cout) * 4 + P * 8
            print(f"L{l} subm {cin:3d}->{cout:3d} x{cnt}: old {t_old:6.1f}us new {t_new:6.1f}us ({by / t_new / 1e3:7.0f} GB/s alg) "
                  f"| max|old-new| {diff:.2e} dIn rel {ddiff:.1e} | err vs fp64: old {e_old:.1e} new {e_new:.1e}", flush=True)
            tot_old += cnt * t_old * 2; tot_new += cnt * t_new * 2
        if 'down' in ent:
            rd = ent['down']; Mo = rd.out_indices.shape[0]; cin, cout = C, planes[l + 1]
            X = torch.randn(M, cin, device=dev, generator=g); W = torch.randn(8, cin, cout, device=dev, generator=g) * 0.05
            dY = torch.randn(Mo, cout, device=dev, generator=g)
            WT = ops._weight_t(W, 0)
            old = ops._conv(X, rd.nbr_p, rd.order, W, None, None, Mo); new = ops._conv_t(X, rd.nbr_p, rd.order, WT, 0, None, None, Mo)
            uold = ops._conv(dY, rd.nbr_up_p, rd.order_up, WT, None, None, M); unew = ops._conv_t(dY, rd.nbr_up_p, rd.order_up, W, 0, None, None, M)
            t1o = timeit(lambda: ops._conv(X, rd.nbr_p, rd.order, W, None, None, Mo))
            t1n = timeit(lambda: ops._conv_t(X, rd.nbr_p, rd.order, WT, 0, None, None, Mo))
            t3o = timeit(lambda: ops._conv(dY, rd.nbr_up_p, rd.order_up, WT, None, None, M))
            t3n = timeit(lambda: ops._conv_t(dY, rd.nbr_up_p, rd.order_up, W, 0, None, None, M))
            print(f"L{l} down {cin:3d}->{cout:3d}: fwd old {t1o:6.1f} new {t1n:6.1f}us diff {float((old - new).abs().max()):.1e} | "
                  f"up/dIn old {t3o:6.1f} new {t3n:6.1f}us diff {float((uold - unew).abs().max()):.1e}", flush=True)
            tot_old += 2 * (t1o + t3o); tot_new += 2 * (t1n + t3n)
    print("plan NB,NW,ZS,DA = %s: estimated per-step conv fwd+dIn: old %.2f ms, new %.2f ms" % (plan, tot_old / 1e3, tot_new / 1e3), flush=True)


if __name__ == "__main__":
    plans = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]] or [(1, 0, 0, 3)]
    if len(plans) == 1:
        run(plans[0])
    else:       # the plan knobs are read once per process (static): one child per plan
        for p in plans:
            subprocess.run([sys.executable, os.path.abspath(__file__), ",".join(str(x) for x in p)])
