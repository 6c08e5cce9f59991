"""The superpoint-level heads as one operator (csrc/heads.hip, wsis_ops.sp_heads) against the module chains of the
reference (backbone_3D_WSIS.py:59-64 ``head(cin, cout)``, :195-216, :253) evaluated by plain PyTorch in float64:
outputs, input gradient, every parameter gradient, running statistics; ragged row counts, more slices than gradient
partials, eval mode, unused outputs, frozen parameters, run-to-run identity."""
import copy

import pytest
import torch
import torch.nn as nn

import wsis_ops

pytestmark = pytest.mark.gpu


def _head(cout, seed):
    torch.manual_seed(seed)
    hd = nn.Sequential(nn.Linear(64, 64), nn.BatchNorm1d(64, eps=1e-4, momentum=0.1), nn.ReLU(inplace=True),
                       nn.Linear(64, cout))
    with torch.no_grad():
        hd[1].weight.uniform_(0.5, 1.5)
        hd[1].bias.uniform_(-0.3, 0.3)
        hd[1].running_mean.uniform_(-0.2, 0.2)
        hd[1].running_var.uniform_(0.5, 1.5)
    return hd


def _setup(S, couts=(20, 3, 1, 1), n_lin=3, seed=0):
    heads = [_head(c, seed + i) for i, c in enumerate(couts)]
    torch.manual_seed(seed + 100)
    lins = [nn.Linear(64, 64, bias=False) for _ in range(n_lin)]
    x = torch.randn(S, 64) * 1.5 + 0.3
    return heads, lins, x


def _reference(heads, lins, x, weights, train=True):
    """float64 module chains on the CPU; loss = sum_k <out_k, weights_k>"""
    h64 = [copy.deepcopy(h).double().train(train) for h in heads]
    l64 = [copy.deepcopy(l).double() for l in lins]
    x64 = x.double().clone().requires_grad_(True)
    outs = [h(x64) for h in h64] + [l(x64) for l in l64]
    loss = sum((o * w.double()).sum() for o, w in zip(outs, weights) if w is not None)
    loss.backward()
    return outs, x64.grad, h64, l64


def _run(heads, lins, x, weights, train=True):
    hg = [copy.deepcopy(h).cuda().train(train) for h in heads]
    lg = [copy.deepcopy(l).cuda() for l in lins]
    xg = x.cuda().clone().requires_grad_(True)
    got = wsis_ops.sp_heads(xg, hg, lg)
    assert got is not None
    outs = got[0] + got[1]
    loss = sum((o * w.cuda()).sum() for o, w in zip(outs, weights) if w is not None)
    loss.backward()
    torch.cuda.synchronize()
    return outs, xg.grad, hg, lg


def _close_params(hg, hr, tol=1e-4, train=True):
    """every parameter gradient of a head; the bias of the first Linear sits in front of a training-mode BatchNorm: its
    true gradient is zero (the reference holds 1e-15), so it is compared on the scale of the layer's weight gradient"""
    wscale = float(hr[0].weight.grad.abs().max())
    for (n, pg), (_, pr) in zip(hg.named_parameters(), hr.named_parameters()):
        _close(pg.grad, pr.grad, tol, scale=wscale if (n == "0.bias" and train) else None)


def _close(a, b, tol=2e-5, scale=None):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    scale = max(float(b.abs().max()), 1e-6) if scale is None else scale
    err = float((a - b).abs().max()) / scale
    assert err < tol, err


@pytest.mark.parametrize("S", [2289, 33, 4100, 5000])
def test_heads_match_module_chains_in_float64(S):
    heads, lins, x = _setup(S)
    torch.manual_seed(7)
    weights = [torch.randn(S, c) for c in (20, 3, 1, 1)] + [torch.randn(S, 64) for _ in lins]
    ro, rdx, rh, rl = _reference(heads, lins, x, weights)
    go, gdx, gh, gl = _run(heads, lins, x, weights)
    for a, b in zip(go, ro):
        _close(a, b)
    _close(gdx, rdx, 5e-5)
    for hg, hr in zip(gh, rh):
        _close_params(hg, hr)
        _close(hg[1].running_mean, hr[1].running_mean)
        _close(hg[1].running_var, hr[1].running_var)
    for lg, lr in zip(gl, rl):
        _close(lg.weight.grad, lr.weight.grad, 1e-4)


def test_heads_eval_mode_uses_running_statistics():
    S = 700
    heads, lins, x = _setup(S, couts=(7,), n_lin=0)
    weights = [torch.randn(S, 7)]
    ro, rdx, rh, _ = _reference(heads, lins, x, weights, train=False)
    go, gdx, gh, _ = _run(heads, lins, x, weights, train=False)
    _close(go[0], ro[0])
    _close(gdx, rdx, 5e-5)
    _close_params(gh[0], rh[0], train=False)
    assert torch.equal(gh[0][1].running_mean.cpu(), heads[0][1].running_mean)       # untouched


def test_unused_outputs_and_frozen_parameters():
    S = 1000
    heads, lins, x = _setup(S)
    for p in heads[1].parameters():
        p.requires_grad_(False)
    lins[2].weight.requires_grad_(False)
    weights = [torch.randn(S, 20), torch.randn(S, 3), None, torch.randn(S, 1), torch.randn(S, 64), None, torch.randn(S, 64)]
    ro, rdx, rh, rl = _reference(heads, lins, x, weights)
    go, gdx, gh, gl = _run(heads, lins, x, weights)
    _close(gdx, rdx, 5e-5)
    assert all(p.grad is None for p in gh[1].parameters()) and gl[2].weight.grad is None
    for i in (0, 3):
        _close_params(gh[i], rh[i])
    # head 2's output had no consumer: its gradients are exact zeros (the Linear / BatchNorm parameters still get tensors)
    assert all(float(p.grad.abs().max()) == 0.0 for p in gh[2].parameters())
    _close(gl[0].weight.grad, rl[0].weight.grad, 1e-4)
    assert float(gl[1].weight.grad.abs().max()) == 0.0


def test_heads_are_reproducible_run_to_run():
    S = 2289
    heads, lins, x = _setup(S)
    weights = [torch.randn(S, c) for c in (20, 3, 1, 1)] + [torch.randn(S, 64) for _ in lins]
    a = _run(heads, lins, x, weights)
    b = _run(heads, lins, x, weights)
    for u, v in zip(a[0], b[0]):
        assert torch.equal(u, v)
    assert torch.equal(a[1], b[1])
    for ha, hb in zip(a[2], b[2]):
        for pa, pb in zip(ha.parameters(), hb.parameters()):
            assert torch.equal(pa.grad, pb.grad)


def test_fallback_conditions():
    heads, lins, x = _setup(64)
    assert wsis_ops.sp_heads(x, heads, lins) is None                      # CPU tensor: the caller walks the modules
    wide = nn.Sequential(nn.Linear(64, 64), nn.BatchNorm1d(64), nn.ReLU(), nn.Linear(64, 40)).cuda()
    assert wsis_ops.sp_heads(x.cuda(), [wide]) is None                    # cout > 32
    hg = [h.cuda() for h in heads]
    hg[0].eval()
    assert wsis_ops.sp_heads(x.cuda(), hg) is None                        # mixed train / eval


@pytest.mark.parametrize("E", [20054, 1, 300])
def test_position_encoding_matches_modules_in_float64(E):
    """fc_position(centre[u] - centre[v]) (backbone_3D_WSIS.py:54-58, 222-224) as one operator"""
    torch.manual_seed(3)
    S = 500
    fc = nn.Sequential(nn.Linear(3, 16), nn.ReLU(), nn.Linear(16, 1))
    centre = torch.randn(S, 3) * 2
    eu, ev = torch.randint(0, S, (E,)), torch.randint(0, S, (E,))
    w = torch.randn(E)
    f64 = copy.deepcopy(fc).double()
    ref = f64(centre.double()[eu] - centre.double()[ev]).reshape(-1)
    (ref * w.double()).sum().backward()
    g = copy.deepcopy(fc).cuda()
    pos = wsis_ops.edge_position_encoding(g, centre.cuda(), eu.cuda(), ev.cuda())
    assert pos is not None and pos.shape == (E,)
    (pos * w.cuda()).sum().backward()
    _close(pos, ref)
    for pg, pr in zip(g.parameters(), f64.parameters()):
        _close(pg.grad, pr.grad, 1e-4)
    pos2 = wsis_ops.edge_position_encoding(copy.deepcopy(fc).cuda(), centre.cuda(), eu.cuda(), ev.cuda())
    assert torch.equal(pos, pos2)
