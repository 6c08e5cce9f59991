"""Per-layer check + microbenchmark of the weight-gradient kernels on the C2 scene's real rulebooks (GPU box):
   python tools/dw2_bench.py          runs itself twice: WSIS_DW2=0 (round-1 kernel) and WSIS_DW2=1 (csrc/spconv_dw2.hip)
For every UNet layer shape: relative error of dW against an fp64 gather-GEMM, two runs bit-identical, us per launch."""
import importlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); importlib.import_module("3d-wsis_amd")
import torch, harness
from spconv import ops

dev = 'cuda:0'


def timeit(f, n=30):
    for _ in range(3): f()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3


def ref_dw(X, nbr, dY):
    K = nbr.shape[0]
    out = torch.zeros(K, X.shape[1], dY.shape[1], dtype=torch.float64, device=X.device)
    for k in range(K):
        g = nbr[k].long(); ok = g >= 0
        out[k] = X[g[ok]].double().t() @ dY[ok].double()
    return out


def run():
    ns = int(os.environ.get("CONV2_SCENES", "1"))
    b = harness.collate([harness.make_scene(1 + i) for i in range(ns)])
    idx = b['voxel_locs'].int().to(dev).contiguous(); shape = [int(s) for s in b['spatial_shape']]
    planes = [32, 64, 96, 128, 160]
    g = torch.Generator(device=dev).manual_seed(0)
    tot = 0.0
    cur_idx, cur_shape = idx, shape
    for l in range(5):
        C = planes[l]
        rb = ops.build_subm_rulebook(cur_idx, cur_shape, [3] * 3, [1] * 3)
        M = cur_idx.shape[0]
        shapes = [(C, C, 8 if l < 4 else 4)] + ([(2 * C, C, 1)] if l < 4 else [])
        for (cin, cout, cnt) in shapes:
            X = torch.randn(M, cin, device=dev, generator=g); dY = torch.randn(M, cout, device=dev, generator=g)
            dW = ops._dw(X, rb.nbr_p, rb.order, dY, 27, X.shape[1], dY.shape[1])
            dW2 = ops._dw(X, rb.nbr_p, rb.order, dY, 27, X.shape[1], dY.shape[1])
            want = ref_dw(X, rb.nbr, dY)
            err = float((dW.double() - want).abs().max()) / float(want.abs().max())
            t = timeit(lambda: ops._dw(X, rb.nbr_p, rb.order, dY, 27, X.shape[1], dY.shape[1]))
            print(f"L{l} subm {cin:3d}->{cout:3d} x{cnt}: {t:6.1f}us  rel err {err:.1e}  repeat-identical {bool(torch.equal(dW, dW2))}", flush=True)
            tot += cnt * t
        if l < 4:      # the 1x1 projection of the decoder blocks (no table: row t pairs with itself)
            X = torch.randn(M, 2 * C, device=dev, generator=g); dY = torch.randn(M, C, device=dev, generator=g)
            dW = ops._dw(X, None, None, dY, 1, 2 * C, C)
            want = (X.double().t() @ dY.double())[None]
            err = float((dW.double() - want).abs().max()) / float(want.abs().max())
            t = timeit(lambda: ops._dw(X, None, None, dY, 1, 2 * C, C))
            print(f"L{l} 1x1  {2 * C:3d}->{C:3d} x1: {t:6.1f}us  rel err {err:.1e}", flush=True)
            tot += t
        if l < 4:
            rd = ops.build_down_rulebook(cur_idx, cur_shape, [2] * 3, [2] * 3, [0] * 3)
            Mo = rd.out_indices.shape[0]; cin, cout = C, planes[l + 1]
            X = torch.randn(M, cin, device=dev, generator=g); dY = torch.randn(Mo, cout, device=dev, generator=g)
            dW = ops._dw(X, rd.nbr_p, rd.order, dY, 8, X.shape[1], dY.shape[1])
            want = ref_dw(X, rd.nbr, dY)
            err = float((dW.double() - want).abs().max()) / float(want.abs().max())
            t1 = timeit(lambda: ops._dw(X, rd.nbr_p, rd.order, dY, 8, X.shape[1], dY.shape[1]))
            # inverse conv: dW of the up path gathers dY-side rows at the fine level
            Xu = torch.randn(Mo, cout, device=dev, generator=g); dYu = torch.randn(M, cin, device=dev, generator=g)
            dWu = ops._dw(Xu, rd.nbr_up_p, rd.order_up, dYu, 8, Xu.shape[1], dYu.shape[1])
            wantu = ref_dw(Xu, rd.nbr_up, dYu)
            erru = float((dWu.double() - wantu).abs().max()) / float(wantu.abs().max())
            t2 = timeit(lambda: ops._dw(Xu, rd.nbr_up_p, rd.order_up, dYu, 8, Xu.shape[1], dYu.shape[1]))
            print(f"L{l} down {cin:3d}->{cout:3d}: {t1:6.1f}us err {err:.1e} | up {cout:3d}->{cin:3d}: {t2:6.1f}us err {erru:.1e}", flush=True)
            tot += t1 + t2
            cur_idx, cur_shape = rd.out_indices, rd.out_shape
    # point-level Linear(32 -> 20) weight gradient (semantic head): dense rows, partial output block
    N = 199790
    X = torch.randn(N, 32, device=dev, generator=g); dY = torch.randn(N, 20, device=dev, generator=g)
    dW = ops._dw(X, None, None, dY, 1, 32, 20)
    want = (X.double().t() @ dY.double())[None]
    err = float((dW.double() - want).abs().max()) / float(want.abs().max())
    t = timeit(lambda: ops._dw(X, None, None, dY, 1, 32, 20))
    print(f"head Linear 32->20 dW over {N} points: {t:6.1f}us  rel err {err:.1e}", flush=True)
    print("WSIS_DW2=%s: estimated per-step dW: %.2f ms" % (os.environ.get("WSIS_DW2", "1"), tot / 1e3), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run()
    else:
        for v in ("0", "1"):
            subprocess.run([sys.executable, os.path.abspath(__file__), "x"], env=dict(os.environ, WSIS_DW2=v))
