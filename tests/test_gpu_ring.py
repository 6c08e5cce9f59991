"""The role-split ring convolution (csrc/spconv3.hip, WSIS_RING=1: consumer / loader / helper waves of a resident
workgroup) on the C2 scene's real tables.  Parity: values against an fp64 gather-GEMM of the product
out[r] = sum_k X[nbr[k][r]] @ W[k] on sampled rows; and, because a team of NT consumers splits the
offsets of a work item exactly as NT waves of spconv_fwd2_kernel do, outputs EQUAL to that kernel's wherever its launch
plan takes the same number of waves -- with bias / residual, the BatchNorm statistics partials and the BatchNorm-backward
partials of the dIn epilogue.  The launch must leave the error word of the sync slot zero (no wait gave up)."""
import pytest
import torch

import harness
import wsis_native as _n
from spconv import ops

pytestmark = [pytest.mark.gpu, pytest.mark.experimental]
DEV = "cuda"


@pytest.fixture(scope="module")
def levels():
    b = harness.collate([harness.bench_scene(1)])
    idx = b["voxel_locs"].int().to(DEV).contiguous()
    shape = [int(s) for s in b["spatial_shape"]]
    out = []
    for _ in range(3):
        rb = ops.build_subm_rulebook(idx, shape, [3] * 3, [1] * 3)
        rd = ops.build_down_rulebook(idx, shape, [2] * 3, [2] * 3, [0] * 3)
        out.append((rb, rd))
        idx, shape = rd.out_indices, rd.out_shape
    return out


def _ring(monkeypatch, on, nt=0):
    monkeypatch.setenv("WSIS_RING", "1" if on else "0")
    monkeypatch.setenv("WSIS_RING_NT", str(nt))
    monkeypatch.setenv("WSIS_RING_MIN_ITEMS", "1")


def _err_word():
    return int(_n.sync_block()[76:80].view(torch.int32).item())


def _ref_rows(X, nbr, W, rows):
    Xd, Wd = X.double(), W.double()
    out = torch.zeros(len(rows), W.shape[2], dtype=torch.float64, device=X.device)
    for k in range(W.shape[0]):
        g = nbr[k][rows].long() if nbr is not None else rows
        ok = g >= 0
        out[ok] += Xd[g[ok]] @ Wd[k]
    return out


def _bn_conv(X, nbr, order, WT, M_out, bn_x, mean, var, gamma, beta):
    K, Cout, Cin = WT.shape
    lib = _n.hip()
    out = torch.empty((M_out, Cout), device=X.device)
    parts = torch.zeros(((M_out + 31) // 32, 2, Cout), device=X.device)
    wsb = lib.wsis_spconv_fwd_t_workspace_bytes(M_out, K, Cin, Cout)
    ws = torch.empty(max(wsb, 256), dtype=torch.uint8, device=X.device)
    _n.check(lib.wsis_spconv_fwd_t_bn(_n.ptr(X), _n.ptr(nbr), _n.ptr(order), _n.ptr(WT), 0, _n.ptr(out), _n.ptr(parts),
                                      _n.ptr(bn_x), _n.ptr(mean), _n.ptr(var), _n.ptr(gamma), _n.ptr(beta), 1e-4, 1,
                                      X.shape[0], M_out, K, Cin, Cout, _n.ptr(ws), wsb, _n.ptr(_n.sync_block(X.device)),
                                      _n.stream_ptr()), "spconv_fwd_t_bn")
    return out, parts


CASES = [   # level, kind, cin, cout, nt, equal to spconv_fwd2_kernel's plan (its wave count = nt)
    (0, "subm", 32, 32, 1, True), (0, "subm", 64, 32, 1, True), (0, "1x1", 64, 32, 1, True), (0, "up", 64, 32, 1, True),
    (0, "down", 32, 64, 2, False), (1, "subm", 64, 64, 4, True), (1, "subm", 128, 64, 2, False), (1, "1x1", 128, 64, 1, True),
    (1, "up", 96, 64, 4, False), (2, "subm", 96, 96, 4, True), (2, "down", 96, 128, 4, False), (2, "subm", 96, 96, 1, False),
]


@pytest.mark.parametrize("level,kind,cin,cout,nt,equal", CASES)
def test_ring_conv_values_and_epilogues(monkeypatch, levels, level, kind, cin, cout, nt, equal):
    rb, rd = levels[level]
    if kind == "subm":
        nbr_p, nbr, order, K, Mi, Mo = rb.nbr_p, rb.nbr, rb.order, 27, rb.in_indices.shape[0], rb.in_indices.shape[0]
    elif kind == "1x1":
        nbr_p, nbr, order, K, Mi, Mo = None, None, None, 1, rb.in_indices.shape[0], rb.in_indices.shape[0]
    elif kind == "down":
        nbr_p, nbr, order, K, Mi, Mo = rd.nbr_p, rd.nbr, rd.order, 8, rd.in_indices.shape[0], rd.out_indices.shape[0]
    else:
        nbr_p, nbr, order, K, Mi, Mo = rd.nbr_up_p, rd.nbr_up, rd.order_up, 8, rd.out_indices.shape[0], rd.in_indices.shape[0]
    g = torch.Generator(device=DEV).manual_seed(level * 31 + cin + cout + nt)
    X = torch.randn(Mi, cin, device=DEV, generator=g)
    W = torch.randn(K, cin, cout, device=DEV, generator=g) * 0.05
    bias = torch.randn(cout, device=DEV, generator=g)
    res = torch.randn(Mo, cout, device=DEV, generator=g)
    bn_x = torch.randn(Mo, cout, device=DEV, generator=g)
    mean, var = bn_x.mean(0), bn_x.var(0, unbiased=False)
    gamma = torch.rand(cout, device=DEV, generator=g) + 0.5
    beta = torch.randn(cout, device=DEV, generator=g) * 0.1
    WT = ops._weight_t(W, 0)
    n_part = (Mo + 31) // 32

    def run():
        st = torch.full((n_part, 2, cout), float("nan"), device=DEV)
        y = ops._conv_t(X, nbr_p, order, WT, 0, bias, res, Mo, stats=st)
        plain = ops._conv_t(X, nbr_p, order, WT, 0, None, None, Mo)
        yb, pb = _bn_conv(X, nbr_p, order, WT, Mo, bn_x, mean, var, gamma, beta)
        torch.cuda.synchronize()
        return y, st, plain, yb, pb

    _ring(monkeypatch, False)
    y0, s0, p0, b0, q0 = run()
    _n.sync_block()[76:80].zero_()
    _ring(monkeypatch, True, nt)
    y1, s1, p1, b1, q1 = run()
    assert _err_word() == 0, "a wait of the ring kernel gave up"
    # values: fp64 gather-GEMM on sampled rows (fp32 product of <= 27 * 192 terms: 2e-6 of the largest value)
    rows = torch.randint(0, Mo, (min(Mo, 2048),), device=DEV, generator=g)
    want = _ref_rows(X, nbr, W, rows)
    scale = float(want.abs().max())
    assert float((p1[rows].double() - want).abs().max()) <= 3e-6 * scale
    assert float((y1[rows].double() - (want + bias.double() + res[rows].double())).abs().max()) <= 3e-6 * max(scale, 4.0)
    if equal:
        assert torch.equal(p0, p1) and torch.equal(y0, y1) and torch.equal(b0, b1)
    else:       # another split of the offsets over the waves: same terms, another order of additions
        assert float((p0 - p1).abs().max()) <= 3e-6 * scale and float((y0 - y1).abs().max()) <= 3e-6 * max(scale, 4.0)
        assert float((b0 - b1).abs().max()) <= 3e-6 * scale
    # slice partials: (sum, centred sum of squares) and (sum dz, sum dz xhat) -- same terms, the finisher's own order
    assert float((s0 - s1).abs().max()) <= 2e-5 * float(s0.abs().max())
    assert float((q0 - q1).abs().max()) <= 2e-5 * float(q0.abs().max())


def test_ring_conv_run_to_run_identical_and_last_slice(monkeypatch, levels):
    """the same launch three times: bit-equal outputs (no atomics, fixed order), also on a tensor whose last slice is
    partial (153,685 rows = 4,802 x 32 + 21) and under a tile order"""
    rb, _ = levels[0]
    M = rb.in_indices.shape[0]
    assert M % 32 != 0
    g = torch.Generator(device=DEV).manual_seed(11)
    X = torch.randn(M, 32, device=DEV, generator=g)
    W = torch.randn(27, 32, 32, device=DEV, generator=g) * 0.05
    WT = ops._weight_t(W, 0)
    _ring(monkeypatch, True, 1)
    outs = [ops._conv_t(X, rb.nbr_p, rb.order, WT, 0, None, None, M) for _ in range(3)]
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    last = torch.arange(M - 21, M, device=DEV)
    want = _ref_rows(X, rb.nbr, W, rb.order[last].long())
    assert float((outs[0][rb.order[last].long()].double() - want).abs().max()) <= 3e-6 * float(want.abs().max())
    assert _err_word() == 0
