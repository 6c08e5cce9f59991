"""The walks of the one-wave work items of the forward / dIn convolution (csrc/spconv2_body.h) on the C2 scene's real level-0
tables (4,803 items: the launches that take the table-in-registers kernel with runs of 64 items per XCD).

Default build: the product equals an fp64 gather-GEMM of out[r] = sum_k X[nbr[k][r]] @ W[k] on sampled
rows -- including the rows of the ragged last slice -- for the submanifold table (32 -> 32, 64 -> 32: two chunks per
offset), the strided pair of tables (down: many pairs per row; up: exactly one) and a ragged row count; two launches give
the same bits.

EXPERIMENTAL build (`-m experimental`): the retired walks -- both operands straight to registers (WSIS_FWD2_RG), the walk
chosen per item (WSIS_FWD2_RGH), the third ring slot (WSIS_FWD2_TR_DA=3), the table image in LDS (WSIS_FWD2_TR=0), the plain
snake instead of the XCD runs (WSIS_FWD2_XCD=0) -- produce the SAME bits as the default: the order of additions of an
output row depends on the offset index alone."""
import pytest
import torch

import harness
import wsis_native as _n
from spconv import ops

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def level0():
    b = harness.collate([harness.bench_scene(1)])
    idx = b["voxel_locs"].int().to(DEV).contiguous()
    shape = [int(s) for s in b["spatial_shape"]]
    rb = ops.build_subm_rulebook(idx, shape, [3] * 3, [1] * 3)
    rd = ops.build_down_rulebook(idx, shape, [2] * 3, [2] * 3, [0] * 3)
    return rb, rd, int(idx.shape[0]), int(rd.out_indices.shape[0])


def _ref_rows(X, nbr, W, rows):
    Xd, Wd = X.double(), W.double()
    out = torch.zeros(len(rows), W.shape[2], dtype=torch.float64, device=X.device)
    for k in range(W.shape[0]):
        g = nbr[k][rows].long()
        ok = g >= 0
        out[ok] += Xd[g[ok]] @ Wd[k]
    return out


def _cases(level0):
    rb, rd, M, Mo = level0
    return [("subm 32->32", 32, 32, rb.nbr_p, rb.order, rb.nbr, 27, M, M),
            ("subm 64->32", 64, 32, rb.nbr_p, rb.order, rb.nbr, 27, M, M),
            ("up   64->32", 64, 32, rd.nbr_up_p, rd.order_up, rd.nbr_up, 8, Mo, M),
            ("down 32->64", 32, 64, rd.nbr_p, rd.order, rd.nbr, 8, M, Mo)]


def _run(case, gen):
    name, cin, cout, nbr_p, order, nbr, K, Mi, Mo = case
    X = torch.randn(Mi, cin, device=DEV, generator=gen)
    W = torch.randn(K, cin, cout, device=DEV, generator=gen) * 0.05
    WT = ops._weight_t(W, 0)
    return X, W, WT, lambda: ops._conv_t(X, nbr_p, order, WT, 0, None, None, Mo)


def test_table_in_registers_kernel_matches_fp64_gather_gemm(level0):
    gen = torch.Generator(device=DEV)
    gen.manual_seed(11)
    for case in _cases(level0):
        name, cin, cout, nbr_p, order, nbr, K, Mi, Mo = case
        X, W, WT, f = _run(case, gen)
        out = f()
        again = f()
        assert torch.equal(out, again), name
        # sampled rows + the rows of the last (ragged) slice in table order
        rows = torch.cat([torch.randint(0, Mo, (2048,), device=DEV, generator=gen),
                          order[-64:].long() if order is not None else torch.arange(Mo - 64, Mo, device=DEV)])
        want = _ref_rows(X, nbr, W, rows)
        err = float((out[rows].double() - want).abs().max()) / max(float(want.abs().max()), 1e-30)
        assert err < 2e-6, (name, err)          # fp32 accumulation of <= 27 * 64 products against fp64


@pytest.mark.experimental
@pytest.mark.parametrize("env", [{"WSIS_FWD2_RG": "1"}, {"WSIS_FWD2_RGH": "12"}, {"WSIS_FWD2_RGH": "200"},
                                 {"WSIS_FWD2_RGH": "12", "WSIS_FWD2_RGH_LATE": "0"}, {"WSIS_FWD2_TR_DA": "3"},
                                 {"WSIS_FWD2_TR": "0"}, {"WSIS_FWD2_XCD": "0"}, {"WSIS_FWD2_XCD": "8"}],
                         ids=lambda e: ",".join(f"{k[10:]}={v}" for k, v in e.items()))
def test_retired_walks_produce_the_same_bits(level0, env, monkeypatch):
    _n.require_experimental("the retired walks of the one-wave items")
    gen = torch.Generator(device=DEV)
    gen.manual_seed(12)
    for case in _cases(level0):
        X, W, WT, f = _run(case, gen)
        for k in ("WSIS_FWD2_RG", "WSIS_FWD2_RGH", "WSIS_FWD2_RGH_LATE", "WSIS_FWD2_TR_DA", "WSIS_FWD2_TR", "WSIS_FWD2_XCD"):
            monkeypatch.delenv(k, raising=False)
        base = f()
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        got = f()
        assert torch.equal(got, base), (case[0], env)
