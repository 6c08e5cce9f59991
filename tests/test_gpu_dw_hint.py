"""The launch-plan hint of the weight-gradient products (include/wsis_hip.h: wsis_hint_batch_rows): with
either plan -- one or two 4-wave workgroups per CU and combination -- dW = sum_r X[nbr[k][r]]^T (x) dY[r] matches an fp64
gather-GEMM, repeats bit for bit, and fits the workspace the library asked for BEFORE the hint changed (it sizes for the
larger plan).  The forward pass sets the hint from its tensor's row count (spconv.ops.prebuild_unet_rulebooks)."""
import pytest
import torch

import harness
import wsis_native as _n
from spconv import ops

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _ref_dw(X, nbr, dY):
    out = torch.zeros(nbr.shape[0], X.shape[1], dY.shape[1], dtype=torch.float64, device=X.device)
    for k in range(nbr.shape[0]):
        g = nbr[k].long()
        ok = g >= 0
        out[k] = X[g[ok]].double().t() @ dY[ok].double()
    return out


def test_weight_gradient_under_both_plans():
    lib = _n.hip()
    b = harness.collate([harness.bench_scene(1)])
    idx = b["voxel_locs"].int().to(DEV).contiguous()
    shape = [int(s) for s in b["spatial_shape"]]
    rd = ops.build_down_rulebook(idx, shape, [2] * 3, [2] * 3, [0] * 3)
    rb = ops.build_subm_rulebook(rd.out_indices, rd.out_shape, [3] * 3, [1] * 3)      # level 1: 26,819 rows
    M = int(rd.out_indices.shape[0])
    gen = torch.Generator(device=DEV)
    gen.manual_seed(5)
    X = torch.randn(M, 64, device=DEV, generator=gen)
    dY = torch.randn(M, 64, device=DEV, generator=gen)
    want = _ref_dw(X, rb.nbr, dY)
    scale = float(want.abs().max())
    try:
        got = {}
        for rows in (0, 153685, 625000):          # no hint / one scene / four scenes
            _n.check(lib.wsis_hint_batch_rows(rows), "hint")
            a = ops._dw(X, rb.nbr_p, rb.order, dY, 27, 64, 64)
            again = ops._dw(X, rb.nbr_p, rb.order, dY, 27, 64, 64)
            assert torch.equal(a, again), rows
            assert float((a.double() - want).abs().max()) / scale < 2e-6, rows
            got[rows] = a
        assert torch.equal(got[0], got[153685])      # below the threshold: the same plan, the same bits
        assert lib.wsis_hint_batch_rows(-1) != 0     # refused, and the message says why
        assert b"row count" in lib.wsis_last_error()
    finally:
        _n.check(lib.wsis_hint_batch_rows(0), "hint")


def test_forward_pass_sets_the_hint(monkeypatch):
    """prebuild_unet_rulebooks -- the first thing either training path does with a batch -- hands the library the rows"""
    lib = _n.hip()
    seen = []
    real = lib.wsis_hint_batch_rows

    class Spy(object):
        def __getattr__(self, name):
            if name == "wsis_hint_batch_rows":
                return lambda rows: (seen.append(int(rows)), real(rows))[1]
            return getattr(lib, name)

    monkeypatch.setattr(_n, "hip", lambda: Spy())
    import spconv
    b = harness.collate([harness.make_scene(3, room=(2.0, 1.6, 1.2), n_box=2)])
    idx = b["voxel_locs"].int().to(DEV).contiguous()
    feats = torch.randn(idx.shape[0], 6, device=DEV)
    t = spconv.SparseConvTensor(feats, idx, b["spatial_shape"], 1)
    ops.prebuild_unet_rulebooks(t, 3)
    assert seen and seen[-1] == int(idx.shape[0])
    monkeypatch.undo()
    _n.check(lib.wsis_hint_batch_rows(0), "hint")
