"""Role-split ring convolution (csrc/spconv3.hip) against spconv_fwd2_kernel on the C2 scene's real rulebooks (GPU box):
   python tools/ring_bench.py [nt ...]        nt = team sizes to try (0 = the plan's own choice), default 0 1 2 4
Per UNet layer shape: value check against an fp64 gather-GEMM on sampled rows, max |ring - fwd2| (bit-identical when the
team size equals fwd2's wave count), the statistics / BatchNorm-backward epilogues, us per launch of both."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); importlib.import_module("3d-wsis_amd")
import numpy as np, torch, harness
import wsis_native as _n
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
    Xd, Wd = X.double(), W.double()
    out = torch.zeros(len(rows), W.shape[2], dtype=torch.float64, device=X.device)
    for k in range(W.shape[0]):
        g = nbr[k][rows].long() if nbr is not None else rows
        ok = g >= 0
        out[ok] += Xd[g[ok]] @ Wd[k]
    return out


def ring(on, nt=0):
    os.environ["WSIS_RING"] = "1" if on else "0"
    os.environ["WSIS_RING_NT"] = str(nt)
    os.environ["WSIS_RING_MIN_ITEMS"] = "1"


def conv_bn(X, nbr, order, WT, flip, M_out, bn_x, mean, var, gamma, beta, relu):
    """dIn product with the BatchNorm-backward slice partials in its epilogue (wsis_spconv_fwd_t_bn)"""
    K, Cout, Cin = WT.shape
    lib = _n.hip()
    out = torch.empty((M_out, Cout), device=X.device)
    parts = torch.zeros(((M_out + 31) // 32, 2, Cout), device=X.device)
    wsb = lib.wsis_spconv_fwd_t_workspace_bytes(M_out, K, Cin, Cout)
    ws = torch.empty(max(wsb, 256), dtype=torch.uint8, device=X.device)
    _n.check(lib.wsis_spconv_fwd_t_bn(_n.ptr(X), _n.ptr(nbr), _n.ptr(order), _n.ptr(WT), int(flip), _n.ptr(out), _n.ptr(parts),
                                      _n.ptr(bn_x), _n.ptr(mean), _n.ptr(var), _n.ptr(gamma), _n.ptr(beta), 1e-4, int(relu),
                                      X.shape[0], M_out, K, Cin, Cout, _n.ptr(ws), wsb, _n.ptr(_n.sync_block(X.device)),
                                      _n.stream_ptr()), "spconv_fwd_t_bn")
    return out, parts


def check(tag, X, nbr_p, nbr, order, W, flip, M, nts, P, g):
    """W: [K, Cin, Cout] weights of the product out = sum_k X[nbr[k]] @ W[k]"""
    K, cin, cout = W.shape
    WT = ops._weight_t(W, 0)
    res = torch.randn(M, cout, device=dev, generator=g)
    rows = torch.randint(0, M, (min(M, 2048),), device=dev, generator=g)
    want = ref_rows(X, nbr, W, rows)
    scale = float(want.abs().max())
    ring(False)
    old = ops._conv_t(X, nbr_p, order, WT, 0, None, None, M)
    st_old = torch.zeros((M + 31) // 32, 2, cout, device=dev)
    old_r = ops._conv_t(X, nbr_p, order, WT, 0, None, res, M, stats=st_old)
    t_old = timeit(lambda: ops._conv_t(X, nbr_p, order, WT, 0, None, None, M))
    line = f"{tag} {cin:3d}->{cout:3d} K={K:2d} M={M}: fwd2 {t_old:6.1f}us |"
    bn_x = torch.randn(M, cout, device=dev, generator=g); mean = bn_x.mean(0); var = bn_x.var(0, unbiased=False)
    gamma = torch.rand(cout, device=dev, generator=g) + 0.5; beta = torch.randn(cout, device=dev, generator=g) * 0.1
    ob, pb = conv_bn(X, nbr_p, order, WT, 0, M, bn_x, mean, var, gamma, beta, 1)
    for nt in nts:
        ring(True, nt)
        new = ops._conv_t(X, nbr_p, order, WT, 0, None, None, M)
        torch.cuda.synchronize()
        err = float((new[rows].double() - want).abs().max()) / scale
        diff = float((new - old).abs().max())
        st_new = torch.zeros((M + 31) // 32, 2, cout, device=dev)
        new_r = ops._conv_t(X, nbr_p, order, WT, 0, None, res, M, stats=st_new)
        dres = float((new_r - old_r).abs().max())
        dst = float((st_new - st_old).abs().max()) / max(float(st_old.abs().max()), 1e-30)
        nb_, pn = conv_bn(X, nbr_p, order, WT, 0, M, bn_x, mean, var, gamma, beta, 1)
        dbn = float((nb_ - ob).abs().max()); dpb = float((pn - pb).abs().max()) / max(float(pb.abs().max()), 1e-30)
        t_new = timeit(lambda: ops._conv_t(X, nbr_p, order, WT, 0, None, None, M))
        by = P * (cin + cout) * 4 + P * 8
        line += f" nt{nt}: {t_new:6.1f}us ({by / t_new / 1e3:5.0f} GB/s) err64 {err:.1e} d {diff:.1e} res {dres:.1e} st {dst:.1e} bn {dbn:.1e}/{dpb:.1e} |"
    ring(True, 0)
    err = int(_n.sync_block(X.device)[76:80].view(torch.int32).item())      # SyncSlot.err of slot 0
    if err:
        line += f" ERR 0x{err:x}"
        _n.sync_block(X.device)[76:80].zero_()
    print(line, flush=True)


def main():
    nts = [int(a) for a in sys.argv[1:]] or [0, 1, 2, 4]
    only = os.environ.get("RING_LEVELS")
    levels = build_levels()
    planes = [32, 64, 96, 128, 160]
    g = torch.Generator(device=dev).manual_seed(0)
    for l, ent in enumerate(levels):
        if only and str(l) not in only.split(","):
            continue
        C = planes[l]; rb = ent['subm']; M, P = ent['M'], ent['P']
        shapes = [(C, C)] + ([(2 * C, C)] if l < 4 else [])
        for (cin, cout) in shapes:
            X = torch.randn(M, cin, device=dev, generator=g); W = torch.randn(27, cin, cout, device=dev, generator=g) * 0.05
            check(f"L{l} subm", X, rb.nbr_p, rb.nbr, rb.order, W, 0, M, nts, P, g)
        if l < 4:      # 1x1 conv of the tail block (no table)
            X = torch.randn(M, 2 * C, device=dev, generator=g); W = torch.randn(1, 2 * C, C, device=dev, generator=g) * 0.05
            check(f"L{l} 1x1 ", X, None, None, None, W, 0, M, nts, M, g)
        if 'down' in ent:
            rd = ent['down']; Mo = rd.out_indices.shape[0]; cin, cout = C, planes[l + 1]
            X = torch.randn(M, cin, device=dev, generator=g); W = torch.randn(8, cin, cout, device=dev, generator=g) * 0.05
            Pd = int((rd.nbr >= 0).sum())
            check(f"L{l} down", X, rd.nbr_p, rd.nbr, rd.order, W, 0, Mo, nts, Pd, g)
            dY = torch.randn(Mo, cout, device=dev, generator=g)
            Wu = W.transpose(1, 2).contiguous()          # dIn of the strided conv / the inverse conv: [8, cout, cin]
            check(f"L{l} up  ", dY, rd.nbr_up_p, rd.nbr_up, rd.order_up, Wu, 0, M, nts, Pd, g)


if __name__ == "__main__":
    main()
