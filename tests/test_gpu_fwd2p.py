"""The persistent form of the wave-autonomous convolution (spconv_fwd2p_kernel: resident workgroups draw work items from
a sharded ticket counter, csrc/spconv2.hip) against the one-shot launch (WSIS_FWD2P=0) on the C2 scene's real tables:
a work item is computed by the same code whoever draws it, so outputs, BatchNorm slice partials and the backward
partials must be EQUAL; the counters in the sync slot must be zero again after every launch."""
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
    for _ in range(2):
        rb = ops.build_subm_rulebook(idx, shape, [3] * 3, [1] * 3)
        rd = ops.build_down_rulebook(idx, shape, [2] * 3, [2] * 3, [0] * 3)
        out.append((rb, rd))
        idx, shape = rd.out_indices, rd.out_shape
    return out


def _both(monkeypatch, fn):
    res = []
    for v in ("0", "2"):                 # 2: the persistent form wherever it applies (the default takes it from 4 rounds on)
        monkeypatch.setenv("WSIS_FWD2P", v)
        res.append(fn())
    torch.cuda.synchronize()
    return res


@pytest.mark.parametrize("level,cin,cout,residual", [(0, 32, 32, True), (0, 64, 32, False), (1, 64, 64, True), (1, 128, 64, False)])
def test_persistent_conv_equals_one_shot(monkeypatch, levels, level, cin, cout, residual):
    rb, _ = levels[level]
    M = rb.in_indices.shape[0]
    assert (M + 31) // 32 * (cout // 32) > (3072 if level == 0 else 1024), "the case must take the persistent form"
    g = torch.Generator(device=DEV).manual_seed(level * 7 + cin)
    X = torch.randn(M, cin, device=DEV, generator=g)
    W = torch.randn(27, cin, cout, device=DEV, generator=g) * 0.05
    res = torch.randn(M, cout, device=DEV, generator=g) if residual else None
    WT = ops._weight_t(W, 0)
    n_part = (M + 31) // 32

    def run():
        st = torch.full((n_part, 2, cout), float("nan"), device=DEV)
        y = ops._conv_t(X, rb.nbr_p, rb.order, WT, 0, None, res, M, stats=st)
        d = ops._conv_t(X[:, :cout].contiguous(), rb.nbr_p, rb.order, W[:, :cout, :cin].contiguous(), 1, None, None, M)
        return y, st, d
    (y0, s0, d0), (y1, s1, d1) = _both(monkeypatch, run)
    assert torch.equal(y0, y1) and torch.equal(s0, s1) and torch.equal(d0, d1)
    assert not _n.sync_block()[:64 * 4096].any(), "ticket counters must be zero again after every launch"
    # three launches in a row through the same slot (the last draw of a launch resets the counter for the next)
    monkeypatch.setenv("WSIS_FWD2P", "2")
    for _ in range(3):
        assert torch.equal(ops._conv_t(X, rb.nbr_p, rb.order, WT, 0, None, res, M), y0)


def test_persistent_strided_and_backward_partials(monkeypatch, levels):
    """the strided product of level 0 -> 1 (K = 8 table) and a dIn product that writes the BatchNorm-backward partials"""
    lib = _n.hip()
    rb, rd = levels[0]
    M, Mo = rb.in_indices.shape[0], rd.out_indices.shape[0]
    g = torch.Generator(device=DEV).manual_seed(5)
    X = torch.randn(M, 32, device=DEV, generator=g)
    W = torch.randn(8, 32, 64, device=DEV, generator=g) * 0.1
    WT = ops._weight_t(W, 0)
    dY = torch.randn(Mo, 64, device=DEV, generator=g)
    xbn = torch.randn(M, 32, device=DEV, generator=g)
    mean, var = xbn.mean(0).contiguous(), xbn.var(0, unbiased=False).contiguous()
    gamma, beta = torch.rand(32, device=DEV, generator=g) + 0.5, torch.randn(32, device=DEV, generator=g)

    def run():
        y = ops._conv_t(X, rd.nbr_p, rd.order, WT, 0, None, None, Mo)
        dx = torch.empty(M, 32, device=DEV)
        part = torch.full(((M + 31) // 32, 2, 32), float("nan"), device=DEV)
        ws = torch.empty(lib.wsis_spconv_fwd_t_workspace_bytes(M, 8, 64, 32), dtype=torch.uint8, device=DEV)
        _n.check(lib.wsis_spconv_fwd_t_bn(_n.ptr(dY), _n.ptr(rd.nbr_up_p), _n.ptr(rd.order_up), _n.ptr(W), 0, _n.ptr(dx),
                                          _n.ptr(part), _n.ptr(xbn), _n.ptr(mean), _n.ptr(var), _n.ptr(gamma), _n.ptr(beta),
                                          1e-4, 1, Mo, M, 8, 64, 32, _n.ptr(ws), ws.numel(), _n.ptr(_n.sync_block()),
                                          _n.stream_ptr()), "fwd_t_bn")
        return y, dx, part
    (y0, dx0, p0), (y1, dx1, p1) = _both(monkeypatch, run)
    assert torch.equal(y0, y1) and torch.equal(dx0, dx1) and torch.equal(p0, p1)
    assert not _n.sync_block()[:64 * 4096].any()
