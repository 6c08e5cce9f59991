"""GPU parity tests: every HIP operator (called through the C ABI via the drop-in packages) against the
CPU oracle on the same seeded inputs.  Integer outputs bit-exact; fp32 tensors within the stated
tolerance (rtol 1e-4 / atol 1e-5 per operator output, SURVEY 8a notes)."""
import numpy as np
import pytest
import torch

import pointgroup_ops
import spconv
import torch_scatter
import wsis_ops
from oracle import affinity_ref, pg_ops, scatter_ref
from oracle import spconv_ref as ref
from tests.util import random_sparse_coords

pytestmark = pytest.mark.gpu
DEV = "cuda"
RTOL, ATOL = 1e-4, 1e-5


def close(a, b, rtol=RTOL, atol=ATOL):
    a, b = a.detach().cpu().double(), torch.as_tensor(b).detach().cpu().double()
    scale = max(1.0, float(b.abs().max())) if b.numel() else 1.0
    ok = torch.allclose(a, b, rtol=rtol, atol=atol * scale)
    if not ok:
        print("max abs err", float((a - b).abs().max()), "scale", scale)
    return ok


# ---------------------------------------------------------------- voxelization (a1 + a2)
@pytest.mark.parametrize("N,C", [(1, 6), (5000, 6), (3000, 3)])
def test_voxelization_fwd_bwd(N, C):
    rng = np.random.default_rng(N)
    coords = np.concatenate([rng.integers(0, 2, (N, 1)), rng.integers(0, 14, (N, 3))], 1).astype(np.int64)
    feats = rng.standard_normal((N, C)).astype(np.float32)
    locs, p2v, v2p = pointgroup_ops.voxelization_idx(torch.from_numpy(coords), 2, 4)
    f = torch.from_numpy(feats).to(DEV).requires_grad_(True)
    out = pointgroup_ops.voxelization(f, v2p.to(DEV), 4)
    expect = pg_ops.voxelization(feats, v2p.numpy(), 4)
    assert np.array_equal(out.detach().cpu().numpy(), expect), "sequential fp32 mean must be bit-exact"
    g = rng.standard_normal(expect.shape).astype(np.float32)
    out.backward(torch.from_numpy(g).to(DEV))
    assert np.array_equal(f.grad.cpu().numpy(), pg_ops.voxelization_backward(g, v2p.numpy(), N, 4))


# ---------------------------------------------------------------- rulebooks (a5, a6)
def _pairs_sets(pairs_t, num_t):
    pairs_t, num_t = pairs_t.cpu().numpy(), num_t.cpu().numpy()
    return [set(zip(pairs_t[k, 0, :num_t[k]].tolist(), pairs_t[k, 1, :num_t[k]].tolist())) for k in range(len(num_t))]


@pytest.mark.parametrize("ksize,pad", [(3, 1), ((1, 3, 3), (0, 1, 1)), ((3, 1, 3), (1, 0, 1))])
@pytest.mark.parametrize("shape,density", [((9, 8, 7), 0.3), ((24, 20, 16), 0.05)])
def test_rulebook_subm(ksize, pad, shape, density):
    idx = random_sparse_coords(11, 2, shape, density)
    t = torch.from_numpy(idx).to(DEV)
    rb = spconv.ops.build_subm_rulebook(t, list(shape), spconv.ops._triple(ksize), spconv.ops._triple(pad))
    got = _pairs_sets(*rb.to_pairs())
    exp = ref.subm_pairs(idx, shape, ksize, pad)
    assert len(got) == len(exp)
    for k, (i_rows, o_rows) in enumerate(exp):
        assert got[k] == set(zip(i_rows.tolist(), o_rows.tolist())), f"offset {k}"
    if rb.order is not None:
        assert sorted(rb.order.cpu().tolist()) == list(range(idx.shape[0]))


@pytest.mark.parametrize("shape", [(8, 8, 8), (9, 7, 8), (33, 20, 17)])
@pytest.mark.parametrize("k,s,p", [(2, 2, 0), (3, 2, 1), (3, 1, 1)])
def test_rulebook_down(shape, k, s, p):
    idx = random_sparse_coords(12, 3, shape, 0.1)
    t = torch.from_numpy(idx).to(DEV)
    rb = spconv.ops.build_down_rulebook(t, list(shape), [k] * 3, [s] * 3, [p] * 3)
    out_idx, out_shape, exp = ref.down_pairs(idx, shape, k, s, p)
    assert rb.out_shape == out_shape
    assert np.array_equal(rb.out_indices.cpu().numpy(), out_idx), "output rows = ascending linear index, exact"
    got = _pairs_sets(*rb.to_pairs())
    for kk, (i_rows, o_rows) in enumerate(exp):
        assert got[kk] == set(zip(i_rows.tolist(), o_rows.tolist()))
    # nbr_up is the transposed table
    up = rb.nbr_up.cpu().numpy()
    down = rb.nbr.cpu().numpy()
    for kk in range(down.shape[0]):
        o = np.nonzero(down[kk] >= 0)[0]
        assert np.array_equal(up[kk][down[kk][o]], o)
        assert (up[kk] >= 0).sum() == len(o)


def test_rulebook_empty_and_single():
    t = torch.zeros((0, 4), dtype=torch.int32, device=DEV)
    rb = spconv.ops.build_subm_rulebook(t, [8, 8, 8], [3, 3, 3], [1, 1, 1])
    assert rb.nbr.shape == (27, 0)
    one = torch.tensor([[0, 3, 3, 3]], dtype=torch.int32, device=DEV)
    rb = spconv.ops.build_subm_rulebook(one, [8, 8, 8], [3, 3, 3], [1, 1, 1])
    n = rb.nbr.cpu().numpy().ravel()
    assert n[13] == 0 and (np.delete(n, 13) == -1).all()
    rbd = spconv.ops.build_down_rulebook(one, [8, 8, 8], [2] * 3, [2] * 3, [0] * 3)
    assert rbd.out_indices.cpu().tolist() == [[0, 1, 1, 1]]


# ---------------------------------------------------------------- sparse conv fwd/bwd (a7-a11)
def _conv_case(kind, cin, cout, shape, density, seed, bias=False, surface=False):
    idx = random_sparse_coords(seed, 2, shape, density, surface=surface)
    M = idx.shape[0]
    g = torch.Generator().manual_seed(seed)
    t_idx = torch.from_numpy(idx).to(DEV)
    if kind == "subm":
        mod = spconv.SubMConv3d(cin, cout, 3, padding=1, bias=bias, indice_key="k").to(DEV)
        pairs = ref.subm_pairs(idx, shape, 3, 1)
        M_out, x_rows = M, M
    elif kind == "subm1":
        mod = spconv.SubMConv3d(cin, cout, 1, bias=bias).to(DEV)
        pairs = [(np.arange(M), np.arange(M))]
        M_out, x_rows = M, M
    elif kind == "down":
        mod = spconv.SparseConv3d(cin, cout, 2, stride=2, bias=bias, indice_key="k").to(DEV)
        out_idx, _, pairs = ref.down_pairs(idx, shape, 2, 2, 0)
        M_out, x_rows = out_idx.shape[0], M
    elif kind == "subm5":           # 125 offsets: several table groups, uniform dW chunk plan
        mod = spconv.SubMConv3d(cin, cout, 5, padding=2, bias=bias, indice_key="k5").to(DEV)
        pairs = ref.subm_pairs(idx, shape, 5, 2)
        M_out, x_rows = M, M
    elif kind == "subm133":         # anisotropic 1x3x3 kernel (9 offsets)
        mod = spconv.SubMConv3d(cin, cout, (1, 3, 3), padding=(0, 1, 1), bias=bias, indice_key="k133").to(DEV)
        pairs = ref.subm_pairs(idx, shape, (1, 3, 3), (0, 1, 1))
        M_out, x_rows = M, M
    elif kind == "down3":           # overlapping strided conv k3 s2 p1 (27 offsets, several parents per voxel)
        mod = spconv.SparseConv3d(cin, cout, 3, stride=2, padding=1, bias=bias, indice_key="k3s2").to(DEV)
        out_idx, _, pairs = ref.down_pairs(idx, shape, 3, 2, 1)
        M_out, x_rows = out_idx.shape[0], M
    else:
        raise ValueError(kind)
    x = torch.randn(x_rows, cin, generator=g)
    xg = x.clone().to(DEV).requires_grad_(True)
    inp = spconv.SparseConvTensor(xg, t_idx, np.array(shape), 2)
    out = mod(inp)
    w = mod.weight.detach().cpu().double().requires_grad_(True)
    b = mod.bias.detach().cpu().double() if bias else None
    xr = x.double().requires_grad_(True)
    expect = ref.pairs_conv(xr, w, pairs, M_out, b)
    assert out.features.shape == expect.shape
    assert close(out.features, expect), f"forward {kind} {cin}->{cout}"
    go = torch.randn(expect.shape, generator=g)
    out.features.backward(go.to(DEV))
    expect.backward(go.double())
    assert close(xg.grad, xr.grad), f"dIn {kind} {cin}->{cout}"
    assert close(mod.weight.grad, w.grad), f"dW {kind} {cin}->{cout}"
    return out, inp, mod


@pytest.mark.parametrize("cin,cout", [(6, 32), (32, 32), (64, 32), (96, 96), (160, 160), (5, 7), (33, 70), (256, 128)])
def test_subm_conv(cin, cout):
    _conv_case("subm", cin, cout, (12, 11, 10), 0.25, 21)


def test_subm_conv_bias_and_surface_tiles():
    _conv_case("subm", 32, 64, (40, 40, 12), 0.9, 22, bias=True, surface=True)


@pytest.mark.parametrize("cin,cout", [(32, 64), (128, 160), (64, 96)])
def test_down_conv(cin, cout):
    _conv_case("down", cin, cout, (13, 12, 11), 0.3, 23)


@pytest.mark.parametrize("kind,cin,cout", [("subm5", 16, 32), ("subm5", 40, 24), ("subm133", 32, 32),
                                           ("down3", 32, 48)])
def test_other_kernel_volumes(kind, cin, cout):
    """kernel volumes the reference's UNet does not use but the operator surface accepts (upstream spconv does)"""
    _conv_case(kind, cin, cout, (14, 12, 10), 0.3, 27)


@pytest.mark.parametrize("kind,cin,cout,shape,density", [("subm", 32, 32, (12, 11, 10), 0.25), ("subm", 64, 32, (12, 11, 10), 0.25),
                                                         ("subm", 96, 96, (23, 19, 9), 0.3), ("subm", 160, 160, (12, 11, 10), 0.25),
                                                         ("subm", 32, 20, (17, 13, 11), 0.3), ("down", 64, 96, (13, 12, 11), 0.3),
                                                         ("subm1", 256, 128, (10, 10, 10), 0.3)])
def test_register_gather_weight_gradient_kernel(monkeypatch, kind, cin, cout, shape, density):
    """round 6: spconv_dw3_kernel (WSIS_DW3=1; the gathered operand by coalesced loads straight into MFMA fragments, no
    LDS staging) against the same fp64 autograd as the default spconv_dw2_kernel -- row counts that are not multiples
    of 4 (the header's 16-byte DMA from a 4-byte-aligned table line), a partial output block, the strided and the
    dense 1x1 form -- and bit-reproducible run to run"""
    monkeypatch.setenv("WSIS_DW3", "1")
    _, inp, mod = _conv_case(kind, cin, cout, shape, density, 29)
    g1 = mod.weight.grad.clone()
    mod.weight.grad = None
    out = mod(spconv.SparseConvTensor(inp.features.detach().clone().requires_grad_(True), inp.indices,
                                      inp.spatial_shape, inp.batch_size))
    torch.manual_seed(5)
    go = torch.randn_like(out.features)
    out.features.backward(go)
    a = mod.weight.grad.clone()
    mod.weight.grad = None
    out = mod(spconv.SparseConvTensor(inp.features.detach().clone().requires_grad_(True), inp.indices,
                                      inp.spatial_shape, inp.batch_size))
    out.features.backward(go)
    assert torch.equal(a, mod.weight.grad)
    assert g1.shape == a.shape


@pytest.mark.experimental
@pytest.mark.parametrize("cin,cout,shape", [(32, 32, (24, 21, 12)), (96, 96, (23, 19, 9)), (64, 32, (12, 11, 10))])
def test_offset_groups_of_the_weight_gradient_kernel_do_not_enter_the_arithmetic(monkeypatch, cin, cout, shape):
    """round 6: which workgroup owns which of the 27 offsets (grouped by index, WSIS_DW_BAL=0, or the activity-balanced
    partition, the default) and how many slabs a product is split into by the round-down rule (WSIS_DW2_PFLOOR) change
    the schedule only: with the same slab count the weight gradient is the same bit for bit"""
    grads = {}
    for bal in ("0", "1"):
        monkeypatch.setenv("WSIS_DW_BAL", bal)
        monkeypatch.setenv("WSIS_DW2_PFLOOR", "0")
        _, inp, mod = _conv_case("subm", cin, cout, shape, 0.3, 31)
        grads[bal] = mod.weight.grad.clone()
    assert torch.equal(grads["0"], grads["1"])


@pytest.mark.parametrize("cin,cout", [(64, 32), (256, 128), (192, 96)])
def test_1x1_conv(cin, cout):
    _conv_case("subm1", cin, cout, (10, 10, 10), 0.3, 24)


def test_inverse_conv_roundtrip_shapes_and_values():
    shape = (13, 12, 11)
    idx = random_sparse_coords(25, 2, shape, 0.3)
    M = idx.shape[0]
    g = torch.Generator().manual_seed(25)
    x = torch.randn(M, 32, generator=g)
    down = spconv.SparseConv3d(32, 64, 2, stride=2, bias=False, indice_key="sp").to(DEV)
    up = spconv.SparseInverseConv3d(64, 32, 2, indice_key="sp", bias=False).to(DEV)
    xg = x.clone().to(DEV).requires_grad_(True)
    t = spconv.SparseConvTensor(xg, torch.from_numpy(idx).to(DEV), np.array(shape), 2)
    mid = down(t)
    back = up(mid)
    assert back.indices.data_ptr() == t.indices.data_ptr() and list(back.spatial_shape) == list(shape)
    out_idx, _, pairs = ref.down_pairs(idx, shape, 2, 2, 0)
    wd = down.weight.detach().cpu().double().requires_grad_(True)
    wu = up.weight.detach().cpu().double().requires_grad_(True)
    xr = x.double().requires_grad_(True)
    e_mid = ref.pairs_conv(xr, wd, pairs, out_idx.shape[0])
    e_back = ref.pairs_conv(e_mid, wu, ref.inverse_pairs(pairs), M)
    assert close(back.features, e_back)
    go = torch.randn(e_back.shape, generator=g)
    back.features.backward(go.to(DEV))
    e_back.backward(go.double())
    assert close(xg.grad, xr.grad) and close(down.weight.grad, wd.grad) and close(up.weight.grad, wu.grad)
    # inputs on the dropped odd plane get exactly zero rows
    dropped = (idx[:, 1] // 2 >= 6) | (idx[:, 2] // 2 >= 6) | (idx[:, 3] // 2 >= 5)
    if dropped.any():
        assert float(back.features[torch.from_numpy(dropped).to(DEV)].abs().max()) == 0.0


def test_conv_is_deterministic_and_order_independent(monkeypatch):
    shape = (30, 30, 10)
    idx = random_sparse_coords(26, 2, shape, 0.8, surface=True)
    g = torch.Generator().manual_seed(26)
    x = torch.randn(idx.shape[0], 32, generator=g).to(DEV)
    mod = spconv.SubMConv3d(32, 32, 3, padding=1, bias=False, indice_key="k").to(DEV)
    outs = []
    for flag in ("1", "1", "0"):
        monkeypatch.setenv("WSIS_MASK_ORDER", flag)
        t = spconv.SparseConvTensor(x, torch.from_numpy(idx).to(DEV), np.array(shape), 2)
        outs.append(mod(t).features.clone())
    assert torch.equal(outs[0], outs[1]), "run-to-run bit-exact"
    assert torch.equal(outs[0], outs[2]), "tile ordering must not change any value (same per-row sum order)"


# ---------------------------------------------------------------- scatter (a15)
@pytest.mark.parametrize("reduce", ["sum", "mean", "max", "min"])
@pytest.mark.parametrize("N,S,C", [(5000, 300, 32), (777, 50, 3), (1000, 400, 1), (300, 7, 70)])
def test_scatter(reduce, N, S, C):
    g = torch.Generator().manual_seed(N + C)
    index = torch.randint(0, S, (N,), generator=g)
    index[0] = S - 1
    src = torch.randn(N, C, generator=g) if C > 1 else torch.randn(N, generator=g)
    xs = src.clone().to(DEV).requires_grad_(True)
    out = torch_scatter.scatter(xs, index.to(DEV), dim=0, reduce=reduce)
    xr = src.clone().double().requires_grad_(True)
    expect = scatter_ref.scatter(xr, index, 0, None, reduce)
    assert out.shape == expect.shape
    assert close(out, expect)
    go = torch.randn(expect.shape, generator=g)
    out.backward(go.to(DEV))
    expect.backward(go.double())
    assert close(xs.grad, xr.grad)


def test_scatter_empty_segments_and_determinism():
    index = torch.tensor([5, 5, 0, 9], device=DEV)
    src = torch.tensor([[1.0], [3.0], [-2.0], [4.0]], device=DEV)
    assert torch_scatter.scatter(src, index, 0, reduce="mean").flatten().tolist() == [-2, 0, 0, 0, 0, 2, 0, 0, 0, 4]
    assert torch_scatter.scatter(src, index, 0, reduce="max").flatten().tolist() == [-2, 0, 0, 0, 0, 3, 0, 0, 0, 4]
    big = torch.randn(20000, 32, device=DEV)
    idx = torch.randint(0, 500, (20000,), device=DEV)
    a = torch_scatter.scatter(big, idx, 0, reduce="mean")
    b = torch_scatter.scatter(big, idx, 0, reduce="mean")
    assert torch.equal(a, b)


@pytest.mark.parametrize("reduce", ["sum", "mean"])
@pytest.mark.parametrize("C", [32, 20])
def test_scatter_many_small_segments_takes_the_narrow_kernel_bit_identically(monkeypatch, reduce, C):
    """N < 4 S with 16 < C <= 32: eight lanes per segment (segment_reduce_narrow_kernel) -- same values, bit for bit, as
    the wave-per-segment kernel (WSIS_SEGMENT_NARROW=0) and the oracle's scatter; empty segments and a segment of 11."""
    g = torch.Generator().manual_seed(3 + C)
    S, N = 6000, 9000
    index = torch.randint(0, S, (N,), generator=g)
    index[:11] = 17
    index[index == 4] = 5                     # an empty segment
    src = torch.randn(N, C, generator=g)
    xs, idx = src.to(DEV), index.to(DEV)
    got = torch_scatter.scatter(xs, idx, dim=0, reduce=reduce, dim_size=S)
    monkeypatch.setenv("WSIS_SEGMENT_NARROW", "0")
    wide = torch_scatter.scatter(xs, idx, dim=0, reduce=reduce, dim_size=S)
    assert torch.equal(got, wide)
    assert close(got, scatter_ref.scatter(src.double(), index, 0, S, reduce))
    assert float(got[4].abs().max()) == 0.0


# ---------------------------------------------------------------- edge affinity (a16)
def _graph(seed, S, deg):
    rng = np.random.default_rng(seed)
    edges = set()
    for u in range(S - 3):          # last nodes have no out-edges: res is shorter than S
        for v in rng.choice(S, size=rng.integers(1, deg + 1), replace=False):
            if u != v:
                edges.add((u, int(v)))
    e = np.array(sorted(edges), dtype=np.int64)
    return e[:, 0], e[:, 1]


@pytest.mark.parametrize("S,deg,D", [(64, 6, 64), (300, 12, 64), (40, 5, 32)])
def test_edge_affinity_fwd_bwd(S, deg, D):
    eu, ev = _graph(S, S, deg)
    g = torch.Generator().manual_seed(S)
    q, k, v = (torch.randn(S, D, generator=g) for _ in range(3))
    pos = torch.randn(len(eu), generator=g)
    tq, tk, tv, tp = (t.clone().to(DEV).requires_grad_(True) for t in (q, k, v, pos))
    graph = wsis_ops.EdgeGraph(torch.from_numpy(eu).to(DEV), torch.from_numpy(ev).to(DEV), S)
    aff, res = wsis_ops.edge_affinity(tq, tk, tv, tp, graph, 1.0 / np.sqrt(D))
    rq, rk, rv, rp = (t.clone().double().requires_grad_(True) for t in (q, k, v, pos))
    e_aff, e_res = affinity_ref.edge_affinity(rq, rk, rv, rp, torch.from_numpy(eu), torch.from_numpy(ev))
    assert res.shape == e_res.shape and res.shape[0] == eu.max() + 1 < S
    assert close(aff, e_aff) and close(res, e_res)
    ga, gr = torch.randn(e_aff.shape, generator=g), torch.randn(e_res.shape, generator=g)
    (aff * ga.to(DEV)).sum().add((res * gr.to(DEV)).sum()).backward()
    (e_aff * ga.double()).sum().add((e_res * gr.double()).sum()).backward()
    for a, b, name in ((tq, rq, "dq"), (tk, rk, "dk"), (tv, rv, "dv"), (tp, rp, "dpos")):
        assert close(a.grad, b.grad, rtol=1e-3, atol=1e-4), name


# ---------------------------------------------------------------- dense affinity + propagation (a17)
@pytest.mark.parametrize("M,N,K", [(64, 64, 64), (100, 70, 33), (257, 130, 5)])
def test_dgemm_f64_mfma(M, N, K):
    g = torch.Generator().manual_seed(M)
    A, B = torch.randn(M, K, generator=g, dtype=torch.float64), torch.randn(K, N, generator=g, dtype=torch.float64)
    C = wsis_ops.dgemm(A.to(DEV), B.to(DEV))
    assert torch.allclose(C.cpu(), A @ B, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("dense", [False, True])
@pytest.mark.parametrize("iterations", [0, 1, 2])
def test_label_propagation(iterations, dense):
    """sparse form (default) and dense f64-MFMA form against the numpy restatement of scannetv2_dataset.py:664-736"""
    S, classes = 180, 6
    eu, ev = _graph(5, S, 8)
    both = np.unique(np.concatenate([np.stack([eu, ev], 1), np.stack([ev, eu], 1)]), axis=0)
    eu, ev = both[:, 0], both[:, 1]
    rng = np.random.default_rng(7)
    aff = rng.random(len(eu)).astype(np.float32)
    adjacency = np.zeros((S, S), dtype=np.int64)
    adjacency[eu, ev] = 1
    pred = rng.integers(0, classes, S)
    conf = rng.random(S).astype(np.float32)
    label = np.full(S, -100)
    lab_ids = rng.choice(S, 20, replace=False)
    label[lab_ids] = pred[lab_ids] = rng.integers(0, classes - 1, 20)   # one class absent
    A_ref = affinity_ref.affinity_matrix(eu, ev, aff, S)
    e_final, e_scores, _ = affinity_ref.weak_label_propagation(A_ref, adjacency, conf, pred, label, iterations, classes)
    A = wsis_ops.affinity_matrix(torch.from_numpy(eu).to(DEV), torch.from_numpy(ev).to(DEV),
                                 torch.from_numpy(aff).to(DEV), S)
    assert np.array_equal(A.cpu().numpy(), A_ref)
    final, scores = wsis_ops.weak_label_propagation(A, adjacency, conf, pred, label, iterations, classes, dense=dense)
    assert np.allclose(scores, e_scores, rtol=1e-10, atol=1e-14)
    assert np.array_equal(final, e_final)


def test_label_propagation_sparse_equals_dense_on_a_scene_sized_graph():
    """S = 1,500 superpoints, 20 classes, multi-edges in the adjacency (igraph counts them) and a self edge: the sparse
    chain and the dense products agree to 1e-12 and pick the same labels"""
    S, classes = 1500, 20
    eu, ev = _graph(11, S, 9)
    both = np.unique(np.concatenate([np.stack([eu, ev], 1), np.stack([ev, eu], 1)]), axis=0)
    eu, ev = both[:, 0], both[:, 1]
    rng = np.random.default_rng(3)
    aff = rng.random(len(eu)).astype(np.float32)
    adjacency = np.zeros((S, S), dtype=np.int64)
    adjacency[eu, ev] = 1
    adjacency[eu[:40], ev[:40]] = 2                     # multi-edges
    pred = rng.integers(0, 5, S) * 3                    # classes 0, 3, 6, 9, 12 predicted: neighbours often agree
    conf = (0.4 + 0.6 * rng.random(S)).astype(np.float32)
    label = np.full(S, -100)
    lab_ids = rng.choice(S, 150, replace=False)
    label[lab_ids] = pred[lab_ids] = rng.integers(0, 4, 150) * 3          # class 12 predicted but never labelled
    A = wsis_ops.affinity_matrix(torch.from_numpy(eu).to(DEV), torch.from_numpy(ev).to(DEV),
                                 torch.from_numpy(aff).to(DEV), S)
    A[5, 5] = 0.25                                      # a self edge
    for it in (0, 1, 2):
        f_s, s_s = wsis_ops.weak_label_propagation(A, adjacency, conf, pred, label, it, classes, dense=False)
        f_d, s_d = wsis_ops.weak_label_propagation(A, adjacency, conf, pred, label, it, classes, dense=True)
        e_f, e_s, _ = affinity_ref.weak_label_propagation(A.cpu().numpy(), adjacency, conf, pred, label, it, classes)
        assert np.allclose(s_s, e_s, rtol=1e-10, atol=1e-14) and np.allclose(s_d, e_s, rtol=1e-10, atol=1e-14)
        assert np.array_equal(f_s, e_f) and np.array_equal(f_d, e_f)
        assert (f_s != -100).sum() > 20


# ---------------------------------------------------------------- ball query (a19) + clustering (a20)
@pytest.mark.parametrize("N,B,r", [(1, 1, 0.05), (700, 2, 0.05), (3000, 3, 0.03)])
def test_ballquery_and_bfs(N, B, r):
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(N)
    sizes = [N // B] * B
    sizes[-1] += N - sum(sizes)
    xyz = (rng.random((N, 3)) * 0.4).astype(np.float32)
    bi = np.repeat(np.arange(B), sizes).astype(np.int32)
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    idx, sl = pointgroup_ops.ballquery_batch_p(torch.from_numpy(xyz).to(DEV), torch.from_numpy(bi).to(DEV),
                                               torch.from_numpy(off).to(DEV), r, 50)
    e_idx, e_sl = pg_ops.ballquery_batch_p(xyz, bi, off, r)
    assert np.array_equal(sl.cpu().numpy(), e_sl) and np.array_equal(idx.cpu().numpy(), e_idx)
    # independent: cKDTree per batch item (strict radius: allow boundary-equal pairs to differ)
    for b in range(B):
        tree = cKDTree(xyz[off[b]:off[b + 1]].astype(np.float64))
        cnt = np.array([len(x) for x in tree.query_ball_point(xyz[off[b]:off[b + 1]].astype(np.float64), r)])
        assert np.abs(cnt - e_sl[off[b]:off[b + 1], 1]).max() <= 1
    sem = rng.integers(0, 2, N).astype(np.int32)
    ci, co = pointgroup_ops.bfs_cluster(torch.from_numpy(sem), idx.cpu(), sl.cpu(), 3)
    ri, ro = pg_ops.bfs_cluster(sem, e_idx, e_sl, 3)
    assert np.array_equal(ci.numpy(), ri) and np.array_equal(co.numpy(), ro)
    # device version (CUDA tensors): the same clusters, the same FIFO discovery order, element for element
    for thr in (3, 1, 40):
        di, do = pointgroup_ops.bfs_cluster(torch.from_numpy(sem).to(DEV), idx, sl, thr)
        ri, ro = pg_ops.bfs_cluster(sem, e_idx, e_sl, thr)
        assert di.is_cuda and di.dtype == torch.int32 and do.dtype == torch.int32
        assert np.array_equal(do.cpu().numpy(), ro), thr
        assert np.array_equal(di.cpu().numpy().reshape(-1, 2), np.asarray(ri).reshape(-1, 2)), thr


def test_bfs_cluster_device_on_long_thin_components_and_large_frontiers():
    """deep BFS (a 1-D chain: one point per level), a wide one (a dense blob: frontiers of several hundred points, more
    than one 256-thread chunk) and label boundaries, against the FIFO oracle"""
    rng = np.random.default_rng(5)
    chain = np.stack([np.arange(600) * 0.02, np.zeros(600), np.zeros(600)], 1)
    blob = rng.normal(0, 0.06, (4000, 3)) + np.array([5.0, 0, 0])
    sheet = np.concatenate([rng.random((3000, 2)) * 1.0, np.zeros((3000, 1))], 1) + np.array([0, 5.0, 0])
    xyz = np.concatenate([chain, blob, sheet]).astype(np.float32)
    perm = rng.permutation(len(xyz))
    xyz = xyz[perm]
    N = len(xyz)
    bi = np.zeros(N, dtype=np.int32)
    off = np.array([0, N], dtype=np.int32)
    sem = (xyz[:, 1] > 5.5).astype(np.int32)              # splits the sheet into two labels
    idx, sl = pointgroup_ops.ballquery_batch_p(torch.from_numpy(xyz).to(DEV), torch.from_numpy(bi).to(DEV),
                                               torch.from_numpy(off).to(DEV), 0.03, 50)
    assert int(sl[:, 1].max()) < 1000
    di, do = pointgroup_ops.bfs_cluster(torch.from_numpy(sem).to(DEV), idx, sl, 20)
    ri, ro = pg_ops.bfs_cluster(sem, idx.cpu().numpy(), sl.cpu().numpy(), 20)
    assert len(ro) - 1 >= 3
    assert np.array_equal(do.cpu().numpy(), ro)
    assert np.array_equal(di.cpu().numpy().reshape(-1, 2), np.asarray(ri).reshape(-1, 2))
    hi, ho = pointgroup_ops.bfs_cluster(torch.from_numpy(sem), idx.cpu(), sl.cpu(), 20)
    assert torch.equal(hi, di.cpu()) and torch.equal(ho, do.cpu())


def test_ballquery_cap_1000():
    N = 1500
    xyz = np.zeros((N, 3), dtype=np.float32)
    bi = np.zeros(N, dtype=np.int32)
    off = np.array([0, N], dtype=np.int32)
    idx, sl = pointgroup_ops.ballquery_batch_p(torch.from_numpy(xyz).to(DEV), torch.from_numpy(bi).to(DEV),
                                               torch.from_numpy(off).to(DEV), 0.01, 50)
    assert (sl[:, 1] == 1000).all() and idx.numel() == 1000 * N
    assert idx[:1000].cpu().tolist() == list(range(1000))


# ---------------------------------------------------------------- fused BatchNorm1d(+ReLU) (a12)
@pytest.mark.parametrize("M,C", [(5000, 32), (1300, 96), (1, 32), (777, 20), (4097, 160)])
@pytest.mark.parametrize("relu", [True, False])
@pytest.mark.parametrize("training", [True, False])
def test_fused_batchnorm_relu(M, C, relu, training):
    if M == 1 and training:
        pytest.skip("torch refuses a single value per channel in training mode")
    g = torch.Generator().manual_seed(M + C)
    x = torch.randn(M, C, generator=g) * 2 + 0.5
    bn = torch.nn.BatchNorm1d(C, eps=1e-4, momentum=0.1)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(C, generator=g) * 0.3)
        bn.running_mean.copy_(torch.randn(C, generator=g) * 0.1)
        bn.running_var.copy_(torch.rand(C, generator=g) + 0.5)
    import copy
    ref = copy.deepcopy(bn).double()
    bn = bn.to(DEV)
    bn.train(training)
    ref.train(training)
    xg = x.clone().to(DEV).requires_grad_(True)
    y = wsis_ops.batch_norm_relu(xg, bn, relu=relu)
    xr = x.clone().double().requires_grad_(True)
    yr = ref(xr)
    if relu:
        yr = torch.relu(yr)
    assert close(y, yr)
    go = torch.randn(M, C, generator=g)
    y.backward(go.to(DEV))
    yr.backward(go.double())
    assert close(xg.grad, xr.grad, rtol=1e-3, atol=1e-4)
    assert close(bn.weight.grad, ref.weight.grad, rtol=1e-3, atol=1e-4) and close(bn.bias.grad, ref.bias.grad, rtol=1e-3, atol=1e-4)
    assert close(bn.running_mean, ref.running_mean) and close(bn.running_var, ref.running_var)
    wsis_ops.flush_bn_counters(bn)
    assert int(bn.state_dict()["num_batches_tracked"]) == int(ref.num_batches_tracked)


# ---------------------------------------------------------------- ECC message passing (a21)
@pytest.mark.parametrize("S,deg", [(50, 4), (400, 9)])
def test_ecc_message_fwd_bwd(S, deg):
    from torch_scatter import SegmentCSR
    eu, ev = _graph(S + 1, S, deg)
    order = np.argsort(ev, kind="stable")          # GraphConvInfo order: sorted by target
    src, dst = torch.from_numpy(eu[order]), torch.from_numpy(ev[order])
    E, C = len(eu), 32
    g = torch.Generator().manual_seed(S)
    x = torch.randn(S, C, generator=g)
    w = torch.randn(E, C, C, generator=g) * 0.2
    xg, wg = x.clone().to(DEV).requires_grad_(True), w.clone().to(DEV).requires_grad_(True)
    csr_s, csr_d = SegmentCSR(src.to(DEV), S), SegmentCSR(dst.to(DEV), S)
    out = wsis_ops.ecc_message(xg, wg, src.to(DEV), dst.to(DEV), csr_s, csr_d)
    xr, wr = x.clone().double().requires_grad_(True), w.clone().double().requires_grad_(True)
    msg = torch.matmul(xr[dst].unsqueeze(1), wr).squeeze(1)
    ref_out = scatter_ref.scatter(msg, src, 0, S, "mean")
    assert close(out, ref_out)
    go = torch.randn(S, C, generator=g)
    out.backward(go.to(DEV))
    ref_out.backward(go.double())
    assert close(xg.grad, xr.grad, rtol=1e-3, atol=1e-4) and close(wg.grad, wr.grad, rtol=1e-3, atol=1e-4)


# ---------------------------------------------------------------- fused GRUCellEx (a21)
@pytest.mark.parametrize("S", [1, 7, 300, 1190])
def test_gru_cell_ex_fwd_bwd(S):
    import copy
    import graphnet
    torch.manual_seed(S)
    cell = graphnet.GRUCellEx(32, 32, bias=True, layernorm=True, ingate=True)
    ref = copy.deepcopy(cell).double()
    cell = cell.to(DEV)
    g = torch.Generator().manual_seed(S + 1)
    x, h = torch.randn(S, 32, generator=g), torch.randn(S, 32, generator=g)
    xg, hg = x.clone().to(DEV).requires_grad_(True), h.clone().to(DEV).requires_grad_(True)
    hy = cell(xg, hg)
    xr, hr = x.clone().double().requires_grad_(True), h.clone().double().requires_grad_(True)
    hyr = ref.forward_reference(xr, hr)
    assert close(hy, hyr)
    go = torch.randn(S, 32, generator=g)
    hy.backward(go.to(DEV))
    hyr.backward(go.double())
    assert close(xg.grad, xr.grad, rtol=1e-3, atol=1e-4) and close(hg.grad, hr.grad, rtol=1e-3, atol=1e-4)
    for (name, p), (_, q) in zip(cell.named_parameters(), ref.named_parameters()):
        assert close(p.grad, q.grad, rtol=1e-3, atol=1e-4), name


# ---------------------------------------------------------------- voxel -> point gather (a14)
def test_gather_rows_fwd_bwd_deterministic():
    g = torch.Generator().manual_seed(3)
    M, N, C = 700, 5000, 32
    src = torch.randn(M, C, generator=g)
    idx = torch.randint(0, M, (N,), generator=g).int()
    sg = src.clone().to(DEV).requires_grad_(True)
    out = wsis_ops.gather_rows(sg, idx.to(DEV))
    assert torch.equal(out.cpu(), src[idx.long()])
    go = torch.randn(N, C, generator=g)
    out.backward(go.to(DEV))
    ref = torch.zeros(M, C, dtype=torch.float64).index_add_(0, idx.long(), go.double())
    assert close(sg.grad, ref)
    g1 = sg.grad.clone()
    sg.grad = None
    wsis_ops.gather_rows(sg, idx.to(DEV)).backward(go.to(DEV))
    assert torch.equal(g1, sg.grad)


def test_ballquery_grid_large_properties_and_exactness():
    """uniform-grid ball query at PointGroup scale: exact against the brute-force oracle on a 20 k subset, and
    size-independent properties (symmetry, self inclusion, ascending lists, prefix-sum offsets) on 250 k points."""
    rng = np.random.default_rng(5)
    B, per = 4, 5000
    xyz = np.concatenate([np.stack([rng.random(per) * 3, rng.random(per) * 2.5, np.round(rng.random(per) * 3) * 0.4
                                    + rng.normal(0, 0.004, per)], 1) for _ in range(B)]).astype(np.float32)
    xyz[:40] = xyz[0]                                   # a dense cluster (many duplicates)
    bi = np.repeat(np.arange(B), per).astype(np.int32)
    off = (np.arange(B + 1) * per).astype(np.int32)
    idx, sl = pointgroup_ops.ballquery_batch_p(torch.from_numpy(xyz).to(DEV), torch.from_numpy(bi).to(DEV),
                                               torch.from_numpy(off).to(DEV), 0.03, 50)
    e_idx, e_sl = pg_ops.ballquery_batch_p(xyz, bi, off, 0.03)
    assert np.array_equal(sl.cpu().numpy(), e_sl) and np.array_equal(idx.cpu().numpy(), e_idx)
    # large
    per = 62500
    N = B * per
    xyz = np.concatenate([np.stack([rng.random(per) * 5, rng.random(per) * 4, np.round(rng.random(per) * 2) * 1.1
                                    + rng.normal(0, 0.003, per)], 1) for _ in range(B)]).astype(np.float32)
    bi = np.repeat(np.arange(B), per).astype(np.int32)
    off = (np.arange(B + 1) * per).astype(np.int32)
    idx, sl = pointgroup_ops.ballquery_batch_p(torch.from_numpy(xyz).to(DEV), torch.from_numpy(bi).to(DEV),
                                               torch.from_numpy(off).to(DEV), 0.03, 50)
    idx, sl = idx.cpu().numpy().astype(np.int64), sl.cpu().numpy().astype(np.int64)
    start, cnt = sl[:, 0], sl[:, 1]
    assert np.array_equal(start, np.concatenate([[0], np.cumsum(cnt)[:-1]])) and idx.size == cnt.sum()
    owner = np.repeat(np.arange(N), cnt)
    assert (cnt >= 1).all() and (bi[owner] == bi[idx]).all()
    d = xyz[owner].astype(np.float32) - xyz[idx].astype(np.float32)
    d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]).astype(np.float32) + (d[:, 2] * d[:, 2]).astype(np.float32)
    assert (d2 < np.float32(0.03) * np.float32(0.03)).all()
    inner = np.ones(idx.size, dtype=bool)
    inner[start[cnt > 0]] = False
    assert (np.diff(idx)[inner[1:]] > 0).all(), "ascending neighbour lists"
    pairs = set(zip(owner[:200000].tolist(), idx[:200000].tolist()))
    sample = list(pairs)[:5000]
    full = set(zip(owner.tolist(), idx.tolist()))
    assert all((b_, a_) in full for a_, b_ in sample), "symmetric relation"
    assert all((p, p) in full for p in range(0, N, 997)), "every point finds itself"
    from scipy.spatial import cKDTree
    sub = np.arange(0, per, 50)
    tree = cKDTree(xyz[:per].astype(np.float64))
    ref_cnt = np.array([len(v) for v in tree.query_ball_point(xyz[sub].astype(np.float64), 0.03)])
    assert np.abs(ref_cnt - cnt[sub]).max() <= 1


# ---------------------------------------------------------------- GPU voxelization_idx (8f-4)
@pytest.mark.parametrize("N", [1, 37, 5000, 300000])
def test_voxelization_idx_gpu_matches_host(N):
    rng = np.random.default_rng(N)
    ext = 6 if N < 100 else 70
    coords = np.concatenate([rng.integers(0, 3, (N, 1)), rng.integers(0, ext, (N, 3))], 1).astype(np.int64)
    coords[N // 2:, 1] += 2 ** 36                      # large values: the hash may not assume small coordinates
    h_locs, h_p2v, h_v2p = pointgroup_ops.voxelization_idx(torch.from_numpy(coords), 3, 4)
    d_locs, d_p2v, d_v2p = pointgroup_ops.voxelization_idx(torch.from_numpy(coords).to(DEV), 3, 4)
    assert d_locs.is_cuda and d_locs.dtype == torch.int64 and d_p2v.dtype == torch.int32
    assert torch.equal(d_p2v.cpu(), h_p2v) and torch.equal(d_locs.cpu(), h_locs) and torch.equal(d_v2p.cpu(), h_v2p)
    # both product paths (host library and device kernels) against the checkers, on the GPU box as well:
    # np.unique re-ordered by first occurrence at every size, the oracle's dict walk where it finishes in seconds
    u, first, inv = np.unique(coords, axis=0, return_index=True, return_inverse=True)
    order = np.argsort(first)
    rank = np.empty_like(order)
    rank[order] = np.arange(len(order))
    cnt = np.bincount(rank[inv.ravel()])
    for locs, p2v, v2p in ((h_locs, h_p2v, h_v2p), (d_locs.cpu(), d_p2v.cpu(), d_v2p.cpu())):
        assert np.array_equal(p2v.numpy(), rank[inv.ravel()]) and np.array_equal(locs.numpy(), u[order])
        assert np.array_equal(v2p[:, 0].numpy(), cnt) and v2p.shape[1] == 1 + cnt.max()
        if N <= 5000:
            rl, rp, rv = pg_ops.voxelization_idx(coords, 3, 4)
            assert np.array_equal(locs.numpy(), rl) and np.array_equal(p2v.numpy(), rp)
            assert np.array_equal(v2p.numpy(), rv)


def test_voxelization_idx_gpu_empty():
    locs, p2v, v2p = pointgroup_ops.voxelization_idx(torch.zeros((0, 4), dtype=torch.int64, device=DEV), 1, 4)
    assert locs.shape == (0, 4) and p2v.shape == (0,) and v2p.shape == (0, 1)


# ---- test-time grouping on the superpoint graph (test_scannetv2.py:281-455) ----------------------

def _cluster_case(tag):
    import os
    from oracle import cluster_ref
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cluster_golden.npz"))
    S = len(g[f"{tag}_sem"])
    return g, S, cluster_ref


class _Graph(object):          # the one igraph method the reference calls
    def __init__(self, lists):
        self.lists = lists

    def neighbors(self, vertex, mode="all"):
        return [int(v) for v in self.lists[int(vertex)]]


@pytest.mark.parametrize("tag", ["a", "b"])
@pytest.mark.parametrize("graph_kind", ["igraph_like", "edge_arrays"])
def test_clustering_in_graph_matches_reference_outputs(tag, graph_kind):
    """product (GPU segmented stages + host BFS) against the outputs of the reference's own function"""
    import inference
    g, S, cluster_ref = _cluster_case(tag)
    lists = cluster_ref.neighbour_lists(g[f"{tag}_edges"], S)
    graph = _Graph(lists) if graph_kind == "igraph_like" else (g[f"{tag}_edges"][:, 0], g[f"{tag}_edges"][:, 1])
    conf, label_id, masks = inference.clustering_in_graph("golden", g[f"{tag}_xyz"], g[f"{tag}_superpoint"], graph,
                                                          g[f"{tag}_sem"], g[f"{tag}_off"], g[f"{tag}_occ"],
                                                          g[f"{tag}_size"])
    assert masks.shape == (len(g[f"{tag}_conf"]), len(g[f"{tag}_xyz"])) and masks.dtype == np.int64
    inst = np.full(masks.shape[1], -1, dtype=np.int64)
    for i, m in enumerate(masks):
        assert (inst[m.astype(bool)] == -1).all(), "instance masks are disjoint"
        inst[m.astype(bool)] = i
    assert np.array_equal(label_id, g[f"{tag}_label_id"])
    assert np.array_equal(inst, g[f"{tag}_inst"]), "bit-exact instance membership"
    assert np.allclose(conf, g[f"{tag}_conf"], rtol=1e-5, atol=0)     # fp32 means, member order differs


def test_clustering_in_graph_matches_oracle_on_a_fresh_scene_and_handles_no_instances():
    import harness
    import inference
    from oracle import cluster_ref
    sc = harness.make_scene(23, room=(1.7, 1.2, 1.3), n_box=5)
    sem, off, occ, size = harness.synthetic_predictions(sc, 23, noise=0.03)
    sem = np.where(np.random.default_rng(5).random(sc["S"]) < 0.1, (sem + 7) % 20, sem)
    lists = cluster_ref.neighbour_lists(sc["edges"], sc["S"])
    xyz = sc["xyz"].astype(np.float32)
    e_conf, e_lab, e_masks = cluster_ref.clustering_in_graph(xyz, sc["superpoint"], lists, sem, off, occ, size)
    conf, lab, masks = inference.clustering_in_graph("s", xyz, sc["superpoint"], _Graph(lists), sem, off, occ, size)
    assert np.array_equal(lab, e_lab) and np.array_equal(masks, e_masks) and np.allclose(conf, e_conf, rtol=1e-5)
    # only wall / floor predictions: no instance at all
    conf, lab, masks = inference.clustering_in_graph("s", xyz, sc["superpoint"], _Graph(lists),
                                                     np.zeros(sc["S"], np.int64), off, occ, size)
    assert len(conf) == 0 and len(lab) == 0 and len(masks) == 0
    with pytest.raises(Exception):
        inference.clustering_in_graph("s", xyz, sc["superpoint"], _Graph(lists), sem, off, occ, size, device="cpu")


def test_clustering_in_graph_s3dis_variant_matches_reference_outputs():
    """test_s3dis.py:297-541 (13 classes, growth radius 0.8 * size, ceiling / floor reported as stuff instances)
    against the outputs of that function (its RANSAC wall split stubbed out when the vectors were made)"""
    import inference
    g, S, cluster_ref = _cluster_case("s")
    lists = cluster_ref.neighbour_lists(g["s_edges"], S)
    conf, label_id, masks = inference.clustering_in_graph(
        "golden", g["s_xyz"], g["s_superpoint"], _Graph(lists), g["s_sem"], g["s_off"], g["s_occ"], g["s_size"],
        semantic_ind2label=inference.S3DIS_LABEL_IDX, valid_labels=inference.S3DIS_VALID_LABELS, radius_factor=0.8,
        stuff_classes=(0, 1))
    want = np.unpackbits(g["s_masks"], axis=1)[:, :masks.shape[1]].astype(np.int64)
    assert np.array_equal(label_id, g["s_label_id"]) and 1 in label_id and 2 in label_id
    assert np.array_equal(masks, want)
    assert np.allclose(conf, g["s_conf"], rtol=1e-5, atol=0)


def test_superpoint_majority_label_matches_scipy_mode_loop():
    """test_scannetv2.py:216-224 restated with its own loop (np.where + scipy.stats.mode per superpoint)"""
    import inference
    from scipy import stats
    rng = np.random.default_rng(3)
    S, N, C = 300, 20000, 20
    sp = rng.integers(0, S, N)
    sp[:S] = np.arange(S)                                  # every superpoint occurs
    pred = rng.integers(0, C, N)
    pred[sp < 40] = (sp[sp < 40] % 3) * 2                  # exact ties inside small superpoints are likely as well
    want = np.zeros(N, dtype=np.int64)
    for s in np.unique(sp):
        m = np.where(sp == s)[0]
        want[m] = np.atleast_1d(stats.mode(pred[m], keepdims=True)[0])[0]
    got, per_sp = inference.superpoint_majority_label(pred, sp, C)
    assert np.array_equal(got.cpu().numpy(), want)
    assert np.array_equal(inference.broadcast_superpoint_label(per_sp, sp).cpu().numpy(), want)


# ---- arithmetic modes of spconv_fwd_kernel (WSIS_CONV_MATH) ---------------------------------------------------

@pytest.mark.parametrize("mode,tol", [("0", 1e-6), ("2", 1e-6), ("1", 2e-5)])
@pytest.mark.parametrize("cin,cout", [(32, 32), (64, 96), (160, 160), (24, 40)])
def test_conv_math_modes_against_fp64(monkeypatch, mode, tol, cin, cout):
    """0: exact fp32 MFMA (default).  2: three bf16 terms per operand, six products (fp32-equivalent: the tolerance
    is the SAME 1e-6 relative L2 as for mode 0).  1: two terms, three products (stated tolerance 2e-5 relative L2).
    Reference: fp64 gather-GEMM-scatter through the same table."""
    import harness
    from spconv import ops
    monkeypatch.setenv("WSIS_CONV_MATH", mode)
    b = harness.collate([harness.make_scene(41, room=(1.6, 1.3, 1.0), n_box=2)])
    idx = b["voxel_locs"].int().to(DEV).contiguous()
    shape = [int(s) for s in b["spatial_shape"]]
    rb = ops.build_subm_rulebook(idx, shape, [3] * 3, [1] * 3)
    M = idx.shape[0]
    g = torch.Generator(device="cpu").manual_seed(5)
    X = torch.randn(M, cin, generator=g).to(DEV)
    W = (torch.randn(27, cin, cout, generator=g) * 0.05).to(DEV)
    res = torch.randn(M, cout, generator=g).to(DEV)
    out = ops._conv(X, rb.nbr_p, rb.order, W, None, res, M)
    assert torch.equal(out, ops._conv(X, rb.nbr_p, rb.order, W, None, res, M)), "deterministic in every mode"
    ref = res.double().clone()
    for k in range(27):
        nb = rb.nbr[k].long()
        v = nb >= 0
        ref[v] += X[nb[v]].double() @ W[k].double()
    rel = float((out.double() - ref).norm() / ref.norm())
    assert rel < tol, (mode, rel)


@pytest.mark.gpu
def test_batched_order_and_pack_edge_cases():
    """wsis_tile_order_batch / wsis_rulebook_pack_batch: no tables, an empty table among others, a table without a
    mask, more than 16 tables (rejected), a batch index range that does not fit the key (rejected)."""
    import ctypes
    import wsis_native as _n
    lib = _n.hip()
    dev = "cuda"
    st = _n.stream_ptr()
    assert lib.wsis_tile_order_batch(0, None, None, None, 4, 1, None, None, 0, st) == 0
    g = torch.Generator().manual_seed(5)
    tabs = []
    for M in (700, 0, 300):
        c = torch.cat([torch.randint(0, 3, (M, 1), generator=g), torch.randint(0, 60, (M, 3), generator=g)], 1).int().to(dev)
        m = torch.randint(0, 1 << 27, (M,), generator=g, dtype=torch.int32).to(dev)
        tabs.append((c, m))
    tabs[2] = (tabs[2][0], None)                        # pure spatial order for the last table
    n = len(tabs)
    Ms = [int(c.shape[0]) for c, _ in tabs]
    N = sum(Ms)
    ws_bytes = lib.wsis_tile_order_batch_workspace_bytes(N)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    order_all = torch.full((N,), -7, dtype=torch.int32, device=dev)
    h_ind = (ctypes.c_void_p * n)(*[c.data_ptr() if c.shape[0] else None for c, _ in tabs])
    h_mask = (ctypes.c_void_p * n)(*[(m.data_ptr() if m is not None and m.shape[0] else None) for _, m in tabs])
    h_M = (ctypes.c_int64 * n)(*Ms)
    _n.check(lib.wsis_tile_order_batch(n, h_ind, h_mask, h_M, 4, 3, order_all.data_ptr(), ws.data_ptr(), ws_bytes, st),
             "tile_order_batch")
    off = 0
    for (c, m), M in zip(tabs, Ms):
        if M == 0:
            continue
        want = torch.empty(M, dtype=torch.int32, device=dev)
        wsb = lib.wsis_tile_order_workspace_bytes(M)
        w1 = torch.empty(wsb, dtype=torch.uint8, device=dev)
        _n.check(lib.wsis_tile_order(c.data_ptr(), None if m is None else m.data_ptr(), M, 4, want.data_ptr(),
                                     w1.data_ptr(), wsb, st), "tile_order")
        assert torch.equal(order_all[off:off + M], want)
        off += M
    # 17 tables / batch size 17: refused with an error status, nothing launched
    h17 = (ctypes.c_void_p * 17)(*[tabs[0][0].data_ptr()] * 17)
    m17 = (ctypes.c_void_p * 17)(*[None] * 17)
    M17 = (ctypes.c_int64 * 17)(*[4] * 17)
    assert lib.wsis_tile_order_batch(17, h17, m17, M17, 4, 1, order_all.data_ptr(), ws.data_ptr(), ws_bytes, st) != 0
    assert lib.wsis_tile_order_batch(n, h_ind, h_mask, h_M, 4, 17, order_all.data_ptr(), ws.data_ptr(), ws_bytes, st) != 0
    assert b"4 bits" in lib.wsis_last_error()
    # pack: one empty table among two real ones == the single-table entry point
    K = 8
    nbrs = [torch.randint(-1, 50, (K, M), generator=g, dtype=torch.int32).to(dev) for M in (700, 0, 300)]
    orders = [torch.randperm(M, generator=g).int().to(dev) for M in (700, 0, 300)]
    outs = [torch.full_like(x, -9) for x in nbrs]
    _n.check(lib.wsis_rulebook_pack_batch(
        3, (ctypes.c_void_p * 3)(*[x.data_ptr() if x.numel() else None for x in nbrs]),
        (ctypes.c_void_p * 3)(*[x.data_ptr() if x.numel() else None for x in orders]),
        (ctypes.c_void_p * 3)(*[x.data_ptr() if x.numel() else None for x in outs]),
        (ctypes.c_int64 * 3)(700, 0, 300), (ctypes.c_int32 * 3)(K, K, K), st), "pack_batch")
    for x, o, got in zip(nbrs, orders, outs):
        if x.numel():
            assert torch.equal(got, x[:, o.long()])
    assert lib.wsis_rulebook_pack_batch(17, None, None, None, None, None, st) != 0


@pytest.mark.gpu
@pytest.mark.parametrize("N,C,frac_ignored", [(5000, 20, 0.7), (257, 13, 0.0), (1, 20, 0.0), (40000, 32, 0.95)])
def test_fused_semantic_loss_matches_torch_formulation(N, C, frac_ignored):
    """wsis_semantic_loss_fwd/bwd (CE with ignore_index + per-class dice, csrc/loss.hip) against the torch
    evaluation of the reference's formulas (losses_3D_WSIS.py:52-67, dice :233-253) in fp64; tolerance: 2e-6
    relative on the loss, 1e-5 of the largest gradient entry on d(scores)."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(N + C)
    x = (torch.randn(N, C, generator=g) * 3).cuda().requires_grad_(True)
    y = torch.randint(0, C, (N,), generator=g)
    y[torch.rand(N, generator=g) < frac_ignored] = -100
    if (y == -100).all():
        y[0] = 1
    y = y.cuda()
    loss, n_kept = wsis_ops.semantic_point_loss(x, y, -100)
    (loss * 1.7).backward()
    xd = x.detach().double().requires_grad_(True)
    keep = y != -100
    ce = F.cross_entropy(xd, y, ignore_index=-100)
    p = F.softmax(xd[keep], -1)
    oh = F.one_hot(y[keep], C).double()
    dice = (2 * (p * oh).sum(0) + 1e-5) / ((p * p).sum(0) + (oh * oh).sum(0) + 1e-4 + 1e-5)
    want = ce + (1 - dice).mean()
    (want * 1.7).backward()
    assert int(n_kept) == int(keep.sum())
    assert abs(float(loss) - float(want)) < 2e-6 * abs(float(want)) + 1e-7
    gmax = float(xd.grad.abs().max())
    assert float((x.grad.double() - xd.grad).abs().max()) < 1e-5 * gmax + 1e-9
    assert float(x.grad[~keep].abs().sum()) == 0.0
    # deterministic
    x2 = x.detach().clone().requires_grad_(True)
    l2, _ = wsis_ops.semantic_point_loss(x2, y, -100)
    (l2 * 1.7).backward()
    assert torch.equal(l2, loss) and torch.equal(x2.grad, x.grad)


@pytest.mark.gpu
def test_flat_adamw_matches_torch_adamw_step_by_step():
    """optim.FlatAdamW (wsis_adamw_step: every tensor in one launch) against torch.optim.AdamW (for-loop
    implementation) on odd-sized tensors, a parameter without gradient, a non-16-byte-aligned gradient view, five
    steps; tolerance 2e-6 of the parameter scale per step (fp32 rounding of a different but equivalent expression)."""
    import wsis_optim as optim
    g = torch.Generator().manual_seed(3)
    shapes = [(3, 3, 3, 6, 32), (32,), (7, 64), (1,), (1025,), (96, 32), (20,)]
    a = [torch.randn(s, generator=g).cuda().requires_grad_(True) for s in shapes]
    b = [t.detach().clone().requires_grad_(True) for t in a]
    mine = optim.FlatAdamW(a, lr=1e-3, weight_decay=1e-4)
    ref = torch.optim.AdamW(b, lr=1e-3, weight_decay=1e-4, foreach=False, fused=False)
    flat = torch.empty(sum(t.numel() for t in a) + 3, device="cuda")
    for step in range(5):
        off = 1                                           # views that start 4 bytes into the buffer: unaligned
        for i, (p, q) in enumerate(zip(a, b)):
            if i == 3 and step % 2 == 0:
                p.grad, q.grad = None, None               # skipped by both
                continue
            gr = torch.randn(p.shape, generator=g).cuda() * (10.0 ** (i - 3))
            view = flat[off:off + p.numel()].view(p.shape)
            view.copy_(gr)
            off += p.numel()
            p.grad, q.grad = view, gr.clone()
        mine.step()
        ref.step()
        for p, q in zip(a, b):
            assert float((p - q).abs().max()) <= 2e-6 * float(q.abs().max()) + 1e-9, (step, p.shape)
    sd = mine.state_dict()
    rs = ref.state_dict()["state"]
    for i in range(len(a)):
        if i in rs:
            for name in ("exp_avg", "exp_avg_sq"):       # fp32 rounding (torch contracts to fma, this build does not)
                want = rs[i][name]
                assert float((sd["state"][i][name] - want).abs().max()) <= 1e-6 * float(want.abs().max()) + 1e-30
            assert float(sd["state"][i]["step"]) == float(rs[i]["step"])
    other = optim.FlatAdamW([t.detach().clone().requires_grad_(True) for t in a], lr=1e-3, weight_decay=1e-4)
    other.load_state_dict(sd)
    assert (other.steps == mine.steps).all() and torch.equal(other.exp_avg, mine.exp_avg)
    with pytest.raises(Exception):
        optim.FlatAdamW([torch.zeros(3, requires_grad=True)])        # CPU parameters are refused


@pytest.mark.gpu
def test_flat_adamw_gradient_clamp_inside_the_step():
    """FlatAdamW.set_grad_clamp: the clamp of train_scannetv2.py:247-249 (``p.grad.data.clamp_(-1, 1)`` over the ECC
    parameters) inside the optimizer's one launch -- parameters equal torch's AdamW on clamped gradients, the clamped
    gradients are written back, unclamped tensors untouched; odd sizes and an unaligned view included"""
    import wsis_optim as optim
    g = torch.Generator().manual_seed(5)
    shapes = [(96, 32), (33,), (1025,), (7, 5)]
    a = [torch.randn(s, generator=g).cuda().requires_grad_(True) for s in shapes]
    b = [t.detach().clone().requires_grad_(True) for t in a]
    mine = optim.FlatAdamW(a, lr=1e-3, weight_decay=1e-4)
    assert mine.set_grad_clamp([a[0], a[2], a[3]], 1.0) == 3
    ref = torch.optim.AdamW(b, lr=1e-3, weight_decay=1e-4, foreach=False, fused=False)
    flat = torch.empty(sum(t.numel() for t in a) + 3, device="cuda")
    for step in range(3):
        off = 1
        raw = []
        for i, (p, q) in enumerate(zip(a, b)):
            gr = torch.randn(p.shape, generator=g).cuda() * 3.0
            view = flat[off:off + p.numel()].view(p.shape)
            view.copy_(gr)
            off += p.numel()
            p.grad = view
            q.grad = gr.clamp(-1.0, 1.0) if i != 1 else gr.clone()
            raw.append(gr)
        mine.step()
        ref.step()
        for i, (p, q) in enumerate(zip(a, b)):
            assert float((p - q).abs().max()) <= 2e-6 * float(q.abs().max()) + 1e-9, (step, i)
            assert torch.equal(p.grad, q.grad), (step, i)          # written back clamped / left alone


@pytest.mark.gpu
def test_flat_adamw_gradient_clamp_keeps_nan():
    """``p.grad.data.clamp_(-1, 1)`` (train_scannetv2.py:247-249) propagates NaN: a diverged ECC gradient must surface
    in .grad and in the parameter, not turn into the bound (fminf / fmaxf drop it); vector path and scalar tail"""
    import wsis_optim as optim
    a = [torch.ones(64, 4).cuda().requires_grad_(True), torch.ones(7).cuda().requires_grad_(True)]
    mine = optim.FlatAdamW(a, lr=1e-3, weight_decay=0.0)
    assert mine.set_grad_clamp(a, 1.0) == 2
    for p in a:
        gr = torch.full(p.shape, 5.0).cuda()
        gr.view(-1)[1] = float("nan")
        gr.view(-1)[-1] = float("nan")
        p.grad = gr
    mine.step()
    for p in a:
        gv, pv = p.grad.view(-1), p.detach().view(-1)
        assert torch.isnan(gv[1]) and torch.isnan(gv[-1]) and torch.isnan(pv[1]) and torch.isnan(pv[-1])
        keep = torch.ones_like(gv, dtype=torch.bool)
        keep[1] = keep[-1] = False
        assert torch.equal(gv[keep], torch.ones_like(gv[keep])) and bool(torch.isfinite(pv[keep]).all())


@pytest.mark.gpu
@pytest.mark.parametrize("S", [1190, 5, 3000])
def test_fused_superpoint_regression_losses_match_torch_formulation(S):
    """wsis_sp_regression_loss_fwd/bwd against the torch evaluation of losses_3D_WSIS.py:79-96,113-127 in fp64
    (boolean-indexed, as the reference): offset L1 / cosine, occupancy and size L1, and the gradients of the three
    predictions; a zero prediction vector (norm backward = 0) and -inf targets on dropped rows included."""
    g = torch.Generator().manual_seed(S)
    pred_off = torch.randn(S, 3, generator=g)
    pred_off[1] = 0.0
    gt_off = torch.randn(S, 3, generator=g)
    pred_occ, gt_occ = torch.randn(S, generator=g), torch.randn(S, generator=g)
    pred_size, gt_size = torch.randn(S, generator=g), torch.rand(S, generator=g)
    sem = torch.randint(0, 20, (S,), generator=g)
    ins = torch.randint(0, 9, (S,), generator=g)
    sem[torch.rand(S, generator=g) < 0.3] = -100
    ins[torch.rand(S, generator=g) < 0.5] = -100
    sem[:2], ins[:2] = 3, 1                                   # the zero-vector row is a kept row
    valid = (sem != -100) & (ins != -100)
    gt_occ[~valid] = float("-inf")                            # log(0) of unlabelled superpoints
    leaves = [t.cuda().requires_grad_(True) for t in (pred_off, pred_occ, pred_size)]
    outs = wsis_ops.sp_regression_losses(leaves[0], gt_off.cuda(), leaves[1], gt_occ.cuda(), leaves[2],
                                         gt_size.cuda(), sem.cuda(), ins.cuda(), -100)
    w = [0.7, 1.3, 0.9, 1.1]
    sum(wi * o for wi, o in zip(w, outs[:4])).backward()
    ref = [t.double().requires_grad_(True) for t in (pred_off, pred_occ, pred_size)]
    n = valid.sum()
    po, go = ref[0][valid], gt_off.double()[valid]
    l_norm = (po - go).abs().sum(-1).sum() / (n + 1e-6)
    gd = go / (go.norm(p=2, dim=1).unsqueeze(-1) + 1e-8)
    pd = po / (po.norm(p=2, dim=1).unsqueeze(-1) + 1e-8)
    l_dir = (-(gd * pd).sum(-1)).sum() / (n + 1e-6)
    l_occ = torch.nn.functional.l1_loss(ref[1][valid], gt_occ.double()[valid])
    l_size = torch.nn.functional.l1_loss(ref[2][valid], gt_size.double()[valid])
    (w[0] * l_norm + w[1] * l_dir + w[2] * l_occ + w[3] * l_size).backward()
    assert float(outs[4]) == float(n)
    for got, want in zip(outs[:4], (l_norm, l_dir, l_occ, l_size)):
        assert abs(float(got) - float(want)) <= 2e-6 * abs(float(want)) + 1e-7
    for a, b in zip(leaves, ref):
        assert torch.isfinite(a.grad).all()
        assert float((a.grad.double().cpu() - b.grad).abs().max()) <= 1e-5 * float(b.grad.abs().max()) + 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("S,C,frac", [(2289, 20, 0.3), (7, 13, 0.0), (3000, 20, 0.95)])
def test_fused_superpoint_cross_entropy_matches_torch(S, C, frac):
    """wsis_sp_ce_loss_fwd/bwd against nn.CrossEntropyLoss(ignore_index) in fp64 (losses_3D_WSIS.py:72-74) and the logged
    scores.sum()"""
    g = torch.Generator().manual_seed(S + C)
    scores = torch.randn(S, C, generator=g) * 3
    labels = torch.randint(0, C, (S,), generator=g)
    labels[torch.rand(S, generator=g) < frac] = -100
    labels[0] = 1                                             # at least one kept row
    leaf = scores.cuda().requires_grad_(True)
    loss, total = wsis_ops.superpoint_cross_entropy(leaf, labels.cuda(), -100)
    (1.7 * loss).backward()
    ref = scores.double().requires_grad_(True)
    want = torch.nn.functional.cross_entropy(ref, labels, ignore_index=-100)
    (1.7 * want).backward()
    assert abs(float(loss) - float(want)) <= 2e-6 * abs(float(want))
    assert abs(float(total) - float(scores.double().sum())) <= 1e-5 * float(scores.double().abs().sum())
    assert float((leaf.grad.double().cpu() - ref.grad).abs().max()) <= 1e-5 * float(ref.grad.abs().max())
    assert bool((leaf.grad[labels.cuda() == -100] == 0).all())


@pytest.mark.gpu
def test_loss_sum_adds_in_the_reference_order():
    """wsis_loss_sum: ((((t0 + t1) + (t2 + t3)) + t4) + t5) + t6 bit for bit, gradient = upstream scalar for every term"""
    vals = [0.1234567, 3.7654321e-3, -0.91234, 0.3333333, 2.25e-5, 1.0101, 0.77]
    terms = [torch.tensor(v, dtype=torch.float32, device="cuda", requires_grad=True) for v in vals]
    got = wsis_ops.loss_sum(terms, paired=1 << 2)
    (2.0 * got).backward()
    t = [torch.tensor(v, dtype=torch.float32) for v in vals]
    want = ((((t[0] + t[1]) + (t[2] + t[3])) + t[4]) + t[5]) + t[6]
    assert float(got) == float(want)
    assert all(float(x.grad) == 2.0 for x in terms)
    # a term of shape [1] (e.g. a loss built with keepdim) or of another float type gets its gradient in its own shape / type
    a = torch.tensor([0.5], dtype=torch.float32, device="cuda", requires_grad=True)
    b = torch.tensor(0.25, dtype=torch.float64, device="cuda", requires_grad=True)
    (3.0 * wsis_ops.loss_sum([a, b])).backward()
    assert a.grad.shape == (1,) and float(a.grad) == 3.0 and b.grad.dtype == torch.float64 and float(b.grad) == 3.0


@pytest.mark.gpu
@pytest.mark.parametrize("S,I,keep", [(1190, 12, 0.6), (1536, 64, 0.9), (40, 1, 1.0), (9, 5, 0.5), (300, 33, 0.05)])
def test_fused_discriminative_loss_matches_torch_formulation(S, I, keep):
    """wsis_disc_loss_fwd/bwd against MultiTaskLoss.discriminative_loss_slots (itself pinned by the reference's
    golden vectors) evaluated in fp64: loss to 3e-6 relative, gradient to 2e-5 of its largest entry; single-member
    instances (zero distance), empty slots and ids beyond the bound included; deterministic."""
    import types
    import losses_3D_WSIS
    pl = types.SimpleNamespace(ignore_label=-100, supervise_instance_size=True, joint_training_epoch=0,
                               semantic_dice=True, supervise_sp_offset=True)
    crit = losses_3D_WSIS.MultiTaskLoss(None, pl, types.SimpleNamespace(classes=20))
    g = torch.Generator().manual_seed(S * 100 + I)
    x = torch.randn(S, 7, generator=g) * 0.7
    ins = torch.randint(0, max(I - 1, 1), (S,), generator=g)
    ins[0] = I - 1                                        # a single-member instance: distance to its own mean is 0
    if S > 20:
        ins[5] = I + 3                                    # beyond the bound: dropped by both
    sem = torch.randint(0, 20, (S,), generator=g)
    drop = torch.rand(S, generator=g) > keep
    drop[0] = False
    sem[drop & (torch.rand(S, generator=g) < 0.5)] = -100
    ins[drop & (sem != -100)] = -100
    valid = (sem != -100) & (ins != -100)
    xg = x.cuda().requires_grad_(True)
    loss = wsis_ops.discriminative_loss(xg, ins.cuda(), sem.cuda(), I, -100, crit.delta_v, crit.delta_d,
                                        crit.param_var, crit.param_dist, crit.param_reg)
    (loss * 1.3).backward()
    xd = x.double().requires_grad_(True)
    want = crit.discriminative_loss_slots(xd, ins, valid, I)
    (want * 1.3).backward()
    assert abs(float(loss) - float(want)) <= 3e-6 * abs(float(want)) + 1e-7, (float(loss), float(want))
    gmax = float(xd.grad.abs().max())
    assert float((xg.grad.double().cpu() - xd.grad).abs().max()) <= 2e-5 * gmax + 1e-10
    assert float(xg.grad[(~valid).cuda()].abs().sum()) == 0.0
    x2 = x.cuda().requires_grad_(True)
    l2 = wsis_ops.discriminative_loss(x2, ins.cuda(), sem.cuda(), I, -100, crit.delta_v, crit.delta_d,
                                      crit.param_var, crit.param_dist, crit.param_reg)
    (l2 * 1.3).backward()
    assert torch.equal(l2, loss) and torch.equal(x2.grad, xg.grad)


@pytest.mark.gpu
def test_flat_adamw_is_a_torch_optimizer_with_live_param_groups():
    """the reference's schedule (PolyLR, an _LRScheduler of power 0.9 stepped per epoch: train_scannetv2.py:93-115,269)
    drives FlatAdamW through ``param_groups``; the state dict survives a round trip into a USED optimizer without stale
    moments (parameters absent from the checkpoint are reset)."""
    import wsis_optim as optim
    g = torch.Generator().manual_seed(5)
    shapes = [(3, 3, 3, 6, 32), (32,), (7, 64)]
    a = [torch.randn(s, generator=g).cuda().requires_grad_(True) for s in shapes]
    b = [t.detach().clone().requires_grad_(True) for t in a]
    mine = optim.FlatAdamW(a, lr=1e-3, weight_decay=1e-4)
    ref = torch.optim.AdamW(b, lr=1e-3, weight_decay=1e-4, foreach=False, fused=False)
    assert isinstance(mine, torch.optim.Optimizer) and len(mine.param_groups) == 1
    poly = lambda e: (1 - e / 8) ** 0.9            # noqa: E731 -- utils/lr_scheduler.py PolyLR
    s_mine = torch.optim.lr_scheduler.LambdaLR(mine, poly)
    s_ref = torch.optim.lr_scheduler.LambdaLR(ref, poly)
    for epoch in range(4):
        for p, q in zip(a, b):
            gr = torch.randn(p.shape, generator=g).cuda()
            p.grad, q.grad = gr, gr.clone()
        mine.step()
        ref.step()
        s_mine.step()
        s_ref.step()
        assert abs(mine.param_groups[0]["lr"] - ref.param_groups[0]["lr"]) < 1e-12 and mine.lr < 1e-3
        for p, q in zip(a, b):
            assert float((p - q).abs().max()) <= 2e-6 * float(q.abs().max()) + 1e-9, epoch
    sd = mine.state_dict()
    assert "initial_lr" in sd["param_groups"][0] and abs(sd["param_groups"][0]["lr"] - mine.lr) < 1e-15
    # resume into a USED optimizer from a checkpoint that lacks parameter 1: its moments and step count are reset
    del sd["state"][1]
    other = optim.FlatAdamW(a, lr=5e-4, weight_decay=1e-4)
    for p in a:
        p.grad = torch.ones_like(p)
    other.step()
    other.load_state_dict(sd)
    assert abs(other.lr - mine.lr) < 1e-15 and other.steps.tolist() == [4, 0, 4]
    n0, n1 = a[0].numel(), a[1].numel()
    off1 = (n0 + 3) // 4 * 4
    assert float(other.exp_avg[off1:off1 + n1].abs().max()) == 0.0
    assert torch.equal(other.exp_avg[:n0], mine.exp_avg[:n0])


@pytest.mark.gpu
def test_rulebooks_wait_for_coordinates_produced_late_on_the_main_stream():
    """a drop-in caller builds ``SparseConvTensor(feats, coords.int(), ...)`` on the main stream, which may still hold
    a backlog; the side-stream rulebook build must be ordered behind the producer of the coordinates even without
    the harness's private ready event"""
    from spconv import ops as sp_ops
    rng = np.random.default_rng(9)
    shape = (40, 36, 30)
    occ = np.argwhere(rng.random(shape) < 0.08)
    idx_h = np.concatenate([np.zeros((len(occ), 1), np.int64), occ], 1)
    want = ref.subm_pairs_fast(idx_h.astype(np.int32), shape, 3, 1)
    for _ in range(3):
        big = torch.randn(4096, 4096, device=DEV)
        for _ in range(30):                               # tens of milliseconds of backlog on the main stream
            big = (big @ big) * 1e-3
        pinned = torch.from_numpy(idx_h).pin_memory()
        coords = torch.empty(idx_h.shape, dtype=torch.int64, device=DEV).fill_(-7)     # garbage until the copy lands
        coords.copy_(pinned, non_blocking=True)
        t = spconv.SparseConvTensor(torch.zeros(len(occ), 4, device=DEV), coords.int(), np.array(shape), 1)
        sp_ops.prebuild_unet_rulebooks(t, 2)
        rb = t.indice_dict["subm1"]
        torch.cuda.synchronize()
        pairs, num = rb.to_pairs()
        for k in range(27):
            got = set(map(tuple, pairs[k, :, :int(num[k])].t().cpu().numpy().tolist()))
            assert got == set(map(tuple, np.asarray(want[k]).T.tolist())), k


# ---------------------------------------------------------------- ECC without the [E, 1024] filter tensor (8f-1)
@pytest.mark.gpu
@pytest.mark.parametrize("S,deg", [(60, 4), (700, 9), (1500, 9)])
def test_ecc_contraction_matches_oracle_and_never_forms_the_filter_tensor(S, deg, monkeypatch):
    """graphnet.RNNGraphConvModule with the filter-free evaluation (m_e = h_e . U_t, U = x @ W') against the oracle's
    RefRNNGraphConv in fp64 (which forms W_e = fnet(f_e) like spg_modules.py:168-183): outputs and every gradient at
    1e-4 of the tensor scale; while it runs no per-edge tensor of E * 1024 elements is allocated (the per-node
    U / dU buffers [R*S, 65*32] are not per-edge).  The largest case has E >= 4096: the filter net's Linear layers
    take the row-split weight-gradient path."""
    import graphnet
    from oracle import network_ref
    eu, ev = _graph(S + 3, S, deg)
    order = np.argsort(ev, kind="stable")
    edge_indexes = torch.from_numpy(np.stack([eu[order], ev[order]]))
    E = edge_indexes.shape[1]
    g = torch.Generator().manual_seed(S)
    feats = torch.randn(E, 13, generator=g)
    x = torch.randn(S, 32, generator=g)
    torch.manual_seed(S)
    ref = network_ref.RefRNNGraphConv(32, 7).double().train()
    net = graphnet.GraphNetwork("gru_7_0", 32, [13, 32, 128, 64], fnet_orthoinit=True, fnet_llbias=True, fnet_bnidx=2)
    mod = net.gconvs[0]
    mod.load_state_dict({k: v.float() for k, v in ref.state_dict().items()}, strict=True)
    mod = mod.to(DEV).train()
    gi = graphnet.GraphConvInfo(edge_indexes.to(DEV), feats.to(DEV), S)
    mod.set_info(gi)
    assert mod._contract_ok(x.to(DEV))
    big = []

    class Spy(torch.utils._python_dispatch.TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            out = func(*args, **(kwargs or {}))
            for t in (out if isinstance(out, (tuple, list)) else (out,)):
                if (isinstance(t, torch.Tensor) and t.is_floating_point() and t.numel() >= E * 1024
                        and t.shape[0] % E == 0):
                    big.append((str(func), tuple(t.shape)))
            return out

    xg = x.clone().to(DEV).requires_grad_(True)
    go = torch.randn(S, 32 * 8, generator=g)
    with Spy():
        out = mod(xg)
        out.backward(go.to(DEV))
    assert not big, big
    assert E >= 4096 or S < 1500
    xr = x.clone().double().requires_grad_(True)
    want = ref(xr, edge_indexes, feats.double())
    want.backward(go.double())

    def ok(a, b, name):
        scale = max(float(b.abs().max()), 1e-6)
        err = float((a.detach().cpu().double() - b).abs().max()) / scale
        assert err <= 1e-4, (name, err)

    ok(out, want.detach(), "out")
    ok(xg.grad, xr.grad, "dx")
    refp = dict(ref.named_parameters())
    gmax = max(float(q.grad.abs().max()) for q in ref.parameters())
    for name, p in mod.named_parameters():
        # a bias that feeds a BatchNorm has a true gradient of zero (rounding noise only): measured against the
        # largest parameter gradient as well
        b = refp[name].grad
        err = float((p.grad.detach().cpu().double() - b).abs().max()) / max(float(b.abs().max()), 5e-2 * gmax)
        assert err <= 1e-4, (name, err)
    # the filter-materialising path gives the same numbers to fp32 rounding
    monkeypatch.setenv("WSIS_ECC_CONTRACT", "0")
    mod.zero_grad()
    x2 = x.clone().to(DEV).requires_grad_(True)
    out2 = mod(x2)
    out2.backward(go.to(DEV))
    ok(out2, want.detach(), "out (filters materialised)")
    ok(x2.grad, xr.grad, "dx (filters materialised)")


@pytest.mark.gpu
@pytest.mark.parametrize("C,rooms", [(32, 1), (64, 1), (128, 3)])
def test_din_epilogue_writes_the_batchnorm_backward_partials(C, rooms):
    """wsis_spconv_fwd_t_bn: the dIn product plus, per 32-row slice, (sum dz, sum dz*xhat) of the BatchNorm(+ReLU)
    behind it; wsis_bn_bwd_from_partials then gives the dgamma / dbeta / dx of wsis_bn_bwd (reduction re-associated:
    1e-5 relative), on every epilogue variant (one wave per slice, four waves, z-slabs + reduce kernel)."""
    import wsis_native as _n
    from spconv import ops
    lib = _n.hip()
    g = torch.Generator(device=DEV).manual_seed(5)
    import harness
    room = (1.6, 1.3, 1.0) if rooms == 1 else (1.2, 1.0, 0.8)       # the small room: fewer slices, more z-slabs
    bt = harness.collate([harness.make_scene(41, room=room, n_box=2 if rooms == 1 else 1)])
    idx = bt["voxel_locs"].int().to(DEV).contiguous()
    shape = [int(v) for v in bt["spatial_shape"]]
    rb = ops.build_subm_rulebook(idx, shape, [3] * 3, [1] * 3)
    M = idx.shape[0]
    dY = torch.randn(M, C, device=DEV, generator=g)
    W = torch.randn(27, C, C, device=DEV, generator=g) * 0.05
    x = torch.randn(M, C, device=DEV, generator=g)
    gamma = torch.randn(C, device=DEV, generator=g)
    beta = torch.randn(C, device=DEV, generator=g) * 0.3
    addend = torch.randn(M, C, device=DEV, generator=g)
    mean, var = x.mean(0).contiguous(), x.var(0, unbiased=False).contiguous()
    eps = 1e-4
    want_dx_in = ops._conv_t(dY, rb.nbr_p, rb.order, W, 1, None, None, M)
    n_part = (M + 31) // 32
    part = torch.full((n_part, 2, C), float("nan"), device=DEV)
    out = torch.empty(M, C, device=DEV)
    wsb = lib.wsis_spconv_fwd_t_workspace_bytes(M, 27, C, C)
    ws = torch.empty(max(wsb, 256) + lib.wsis_bn_stats_finalize_workspace_bytes(n_part, C), dtype=torch.uint8, device=DEV)
    for relu in (1, 0):
        _n.check(lib.wsis_spconv_fwd_t_bn(_n.ptr(dY), _n.ptr(rb.nbr_p), _n.ptr(rb.order), _n.ptr(W), 1, _n.ptr(out),
                                          _n.ptr(part), _n.ptr(x), _n.ptr(mean), _n.ptr(var), _n.ptr(gamma), _n.ptr(beta),
                                          eps, relu, M, M, 27, C, C, _n.ptr(ws), ws.numel(), _n.ptr(_n.sync_block()),
                                          _n.stream_ptr()), "fwd_t_bn")
        assert torch.equal(out, want_dx_in)
        xh = (x - mean) * torch.rsqrt(var + eps)
        dz = torch.where((xh * gamma + beta <= 0) if relu else torch.zeros_like(xh, dtype=torch.bool), torch.zeros_like(out), out)
        # which rows form a slice differs between the epilogue variants (tile order / row order): the totals count
        a, b = dz.double().sum(0), (dz * xh).double().sum(0)
        sc = max(float(a.abs().max()), float(b.abs().max()), 1.0)
        assert not torch.isnan(part).any()
        assert float((part[:, 0].double().sum(0) - a).abs().max()) < 1e-5 * sc
        assert float((part[:, 1].double().sum(0) - b).abs().max()) < 1e-5 * sc
        dx1, dg1, db1 = torch.empty_like(x), torch.empty(C, device=DEV), torch.empty(C, device=DEV)
        dx2, dg2, db2 = torch.empty_like(x), torch.empty(C, device=DEV), torch.empty(C, device=DEV)
        _n.check(lib.wsis_bn_bwd_from_partials(_n.ptr(part), n_part, _n.ptr(x), _n.ptr(out), _n.ptr(mean), _n.ptr(var),
                                               _n.ptr(gamma), _n.ptr(beta), eps, relu, _n.ptr(dx1), _n.ptr(dg1), _n.ptr(db1),
                                               _n.ptr(addend), M, C, _n.ptr(ws), ws.numel(), _n.ptr(_n.sync_block()),
                                               _n.stream_ptr()), "bn_bwd_from_partials")
        wb = torch.empty(lib.wsis_bn_workspace_bytes(M, C), dtype=torch.uint8, device=DEV)
        _n.check(lib.wsis_bn_bwd(_n.ptr(x), _n.ptr(out), _n.ptr(mean), _n.ptr(var), _n.ptr(gamma), _n.ptr(beta), eps, relu, 1,
                                 _n.ptr(dx2), _n.ptr(dg2), _n.ptr(db2), _n.ptr(addend), M, C, _n.ptr(wb), wb.numel(),
                                 _n.stream_ptr()), "bn_bwd")
        for u, v in ((dg1, dg2), (db1, db2), (dx1, dx2)):
            assert float((u - v).abs().max()) <= 1e-5 * max(float(v.abs().max()), 1.0)


@pytest.mark.gpu
@pytest.mark.parametrize("rows,cin,cout", [(199790, 32, 20), (30011, 64, 32), (5000, 128, 64), (777, 32, 4)])
def test_dense_weight_gradient_on_the_wave_autonomous_kernel(rows, cin, cout):
    """wsis_spconv_dw without a table (K = 1: row t pairs with itself) is X^T dY -- the 1x1 projections of the decoder
    blocks and the point-level Linear layers (tall_linear); a partial last output block (Cout % 4 == 0) is allowed.
    Against fp64, and bit-identical when repeated."""
    from spconv import ops
    g = torch.Generator(device=DEV).manual_seed(rows)
    X = torch.randn(rows, cin, device=DEV, generator=g)
    dY = torch.randn(rows, cout, device=DEV, generator=g)
    dW = ops._dw(X, None, None, dY, 1, cin, cout)
    want = (X.double().t() @ dY.double())[None]
    assert dW.shape == (1, cin, cout)
    assert float((dW.double() - want).abs().max()) <= 2e-6 * float(want.abs().max())
    assert torch.equal(dW, ops._dw(X, None, None, dY, 1, cin, cout))


# ---------------------------------------------------------------- column sums (Linear bias gradient)
@pytest.mark.gpu
@pytest.mark.parametrize("M,C", [(1, 4), (255, 20), (2289, 64), (4097, 128), (199790, 32), (300000, 1024), (0, 8)])
def test_colsum_matches_fp64_and_is_run_to_run_identical(M, C):
    """wsis_colsum: out[c] = sum_r x[r, c] against an fp64 sum (1e-6 of sum |x|), bit-identical between two launches
    (fixed summation order), shapes it does not take fall back to torch.sum."""
    g = torch.Generator(device=DEV).manual_seed(M + C)
    x = torch.randn(M, C, device=DEV, generator=g)
    a = wsis_ops.colsum(x)
    b = wsis_ops.colsum(x)
    assert a.shape == (C,) and torch.equal(a, b)
    want = x.double().sum(0)
    bound = 1e-6 * max(float(x.double().abs().sum(0).max()) if M else 0.0, 1e-30)
    assert float((a.double() - want).abs().max()) <= bound if M else float(a.abs().max()) == 0.0
    odd = torch.randn(17, 7, device=DEV, generator=g)
    assert torch.allclose(wsis_ops.colsum(odd), odd.sum(0))


@pytest.mark.gpu
def test_edge_graph_row_count_from_the_loader_equals_the_device_read_back():
    """EdgeGraph(num_src=...) (the loader's host-side edge_u.max() + 1, no device read-back) gives the tensors of the
    default constructor, which mirrors the reference's scatter(..., edge_u) row count."""
    S, D = 300, 64
    eu, ev = _graph(S, S, 12)
    g = torch.Generator().manual_seed(3)
    q, k, v = (torch.randn(S, D, generator=g).to(DEV) for _ in range(3))
    pos = torch.randn(len(eu), generator=g).to(DEV)
    tu, tv = torch.from_numpy(eu).to(DEV), torch.from_numpy(ev).to(DEV)
    a = wsis_ops.EdgeGraph(tu, tv, S)
    b = wsis_ops.EdgeGraph(tu, tv, S, num_src=int(eu.max()) + 1)
    assert a.Su == b.Su == int(eu.max()) + 1 < S
    ra, rb = (wsis_ops.edge_affinity(q, k, v, pos, gr, 0.125) for gr in (a, b))
    assert torch.equal(ra[0], rb[0]) and torch.equal(ra[1], rb[1])


@pytest.mark.gpu
def test_batch_graphs_built_on_the_side_stream_equal_the_in_stream_build(monkeypatch):
    """harness.build_batch_graphs: the CSRs of a batch built on the side stream (default) are the tensors of the
    in-stream build, and a step that uses them right away sees them complete (stream join)."""
    import harness
    bt = harness.to_device(harness.collate([harness.make_scene(11, room=(1.6, 1.3, 1.0), n_box=2)]), DEV)

    def snap():
        cs = [bt["superpoint_csr"], bt["p2v_csr"], bt["edge_graph"].csr_u, bt["edge_graph"].csr_v,
              bt["GIs"][0].csr(), bt["GIs"][0].csr_dst()]
        return [(c.perm.clone(), c.offsets.clone()) for c in cs]

    harness.build_batch_graphs(bt, side_stream=True)
    side = snap()
    torch.cuda.synchronize()
    harness.build_batch_graphs(bt, side_stream=False)
    main = snap()
    for (p0, o0), (p1, o1) in zip(side, main):
        assert torch.equal(p0, p1) and torch.equal(o0, o1)


# ---------------------------------------------------------------- BatchNorm: statistics finish + apply in one launch
@pytest.mark.gpu
@pytest.mark.experimental
@pytest.mark.parametrize("M,C", [(153685, 32), (40003, 64), (12011, 96), (3300, 128), (1100, 160), (70, 32), (200000, 256)])
def test_bn_finalize_apply_in_one_launch_equals_the_two_calls(M, C, monkeypatch):
    monkeypatch.setenv("WSIS_BN_FUSED_APPLY", "1")      # (the one-launch form is opt-in since round 3)
    """wsis_bn_stats_finalize_apply (chunk stage + tickets, epoch flag, apply by the waiting workgroups) against
    wsis_bn_stats_finalize followed by wsis_bn_apply on the same slice partials: mean, var, running statistics and y bit
    for bit, twice in a row (the flag / ticket / done counters are ready for the next launch), and y against torch."""
    import wsis_native as _n
    lib = _n.hip()
    g = torch.Generator(device=DEV).manual_seed(M + C)
    x = torch.randn(M, C, device=DEV, generator=g) * 1.7 + 0.3
    gamma, beta = torch.rand(C, device=DEV, generator=g) + 0.5, torch.randn(C, device=DEV, generator=g)
    n_part = (M + 31) // 32
    pad = n_part * 32 - M
    xp = torch.cat([x, torch.zeros(pad, C, device=DEV)]) if pad else x
    blocks = xp.view(n_part, 32, C)
    cnt = torch.full((n_part, 1), 32.0, device=DEV)
    cnt[-1] = 32 - pad
    s = blocks.sum(1)
    mask = (torch.arange(n_part * 32, device=DEV) < M).view(n_part, 32, 1).float()
    q = (((blocks - (s / cnt).unsqueeze(1)) * mask) ** 2).sum(1)
    partial = torch.stack([s, q], 1).contiguous()                     # [n_part, 2, C]
    ws_bytes = lib.wsis_bn_stats_finalize_workspace_bytes(n_part, C)
    st = _n.stream_ptr()

    def run(fused):
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=DEV)
        mean, var = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
        rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
        y = torch.empty_like(x)
        if fused:
            _n.check(lib.wsis_bn_stats_finalize_apply(_n.ptr(partial), n_part, M, C, _n.ptr(mean), _n.ptr(var), _n.ptr(rm),
                                                      _n.ptr(rv), 0.1, _n.ptr(x), _n.ptr(gamma), _n.ptr(beta), 1e-4, 1,
                                                      _n.ptr(y), _n.ptr(ws), ws_bytes, _n.ptr(_n.sync_block()), st), "fused")
        else:
            _n.check(lib.wsis_bn_stats_finalize(_n.ptr(partial), n_part, M, C, _n.ptr(mean), _n.ptr(var), _n.ptr(rm),
                                                _n.ptr(rv), 0.1, _n.ptr(ws), ws_bytes, None, st), "finalize")
            _n.check(lib.wsis_bn_apply(_n.ptr(x), _n.ptr(mean), _n.ptr(var), _n.ptr(gamma), _n.ptr(beta), 1e-4, 1,
                                       _n.ptr(y), M, C, st), "apply")
        torch.cuda.synchronize()
        return mean, var, rm, rv, y

    ref = run(False)
    for _ in range(3):
        got = run(True)
        for a, b in zip(ref, got):
            assert torch.equal(a, b)
    want = torch.relu(torch.nn.functional.batch_norm(x, None, None, gamma, beta, True, 0.1, 1e-4))
    assert torch.allclose(ref[4], want, rtol=1e-4, atol=1e-4)
    assert not _n.sync_block()[:64 * 4096].any(), "every launch must leave its sync slot zero"
    assert _n.sync_errors() == []


@pytest.mark.gpu
@pytest.mark.experimental
def test_fused_batchnorm_launches_on_two_streams_concurrently_do_not_interfere(monkeypatch):
    monkeypatch.setenv("WSIS_BN_FUSED_APPLY", "1")
    """the sync words of the one-launch BatchNorm forms live in the caller's slots (no device globals): 1,000 launches on
    each of two streams at the same time, each stream with its own sync block, every result bit-identical to the
    two-launch form; no bounded wait runs out (SURVEY 8b: re-entrant, no global mutable state)"""
    import wsis_native as _n
    lib = _n.hip()
    cases = []
    for M, C in ((12011, 96), (40003, 64)):
        g = torch.Generator(device=DEV).manual_seed(M)
        x = torch.randn(M, C, device=DEV, generator=g)
        gamma, beta = torch.rand(C, device=DEV, generator=g) + 0.5, torch.randn(C, device=DEV, generator=g)
        n_part = (M + 31) // 32
        pad = n_part * 32 - M
        blocks = torch.cat([x, torch.zeros(pad, C, device=DEV)]).view(n_part, 32, C)
        cnt = torch.full((n_part, 1), 32.0, device=DEV)
        cnt[-1] = 32 - pad
        s = blocks.sum(1)
        mask = (torch.arange(n_part * 32, device=DEV) < M).view(n_part, 32, 1).float()
        q = (((blocks - (s / cnt).unsqueeze(1)) * mask) ** 2).sum(1)
        partial = torch.stack([s, q], 1).contiguous()
        wsb = lib.wsis_bn_stats_finalize_workspace_bytes(n_part, C)
        ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
        mean, var, y = torch.empty(C, device=DEV), torch.empty(C, device=DEV), torch.empty_like(x)
        _n.check(lib.wsis_bn_stats_finalize(_n.ptr(partial), n_part, M, C, _n.ptr(mean), _n.ptr(var), None, None, 0.1,
                                            _n.ptr(ws), wsb, None, _n.stream_ptr()), "finalize")
        _n.check(lib.wsis_bn_apply(_n.ptr(x), _n.ptr(mean), _n.ptr(var), _n.ptr(gamma), _n.ptr(beta), 1e-4, 1, _n.ptr(y),
                                   M, C, _n.stream_ptr()), "apply")
        cases.append(dict(M=M, C=C, x=x, gamma=gamma, beta=beta, partial=partial, n_part=n_part, wsb=wsb, want=y,
                          ws=torch.empty(wsb, dtype=torch.uint8, device=DEV), mean=torch.empty(C, device=DEV),
                          var=torch.empty(C, device=DEV), y=torch.empty_like(x), stream=torch.cuda.Stream()))
    torch.cuda.synchronize()
    for c in cases:
        with torch.cuda.stream(c["stream"]):
            c["sync"] = _n.sync_block()            # one block per (device, stream)
    assert cases[0]["sync"].data_ptr() != cases[1]["sync"].data_ptr()
    for it in range(1000):
        for c in cases:                            # alternate: both streams always have launches queued
            with torch.cuda.stream(c["stream"]):
                _n.check(lib.wsis_bn_stats_finalize_apply(
                    _n.ptr(c["partial"]), c["n_part"], c["M"], c["C"], _n.ptr(c["mean"]), _n.ptr(c["var"]), None, None, 0.1,
                    _n.ptr(c["x"]), _n.ptr(c["gamma"]), _n.ptr(c["beta"]), 1e-4, 1, _n.ptr(c["y"]), _n.ptr(c["ws"]),
                    c["wsb"], _n.ptr(c["sync"]), _n.stream_ptr()), "fused")
        if it % 250 == 249:
            torch.cuda.synchronize()
            for c in cases:
                assert torch.equal(c["y"], c["want"])
    torch.cuda.synchronize()
    assert _n.sync_errors() == []
    for c in cases:
        assert not c["sync"].any()


@pytest.mark.gpu
@pytest.mark.parametrize("M,C", [(3300, 128), (344, 160), (70, 32), (4100, 32), (6572, 96), (9000, 128), (10007, 64), (10240, 32)])
def test_small_level_batchnorm_forward_finish_and_apply_in_one_launch(M, C, monkeypatch):
    """wsis_bn_stats_finalize_apply on levels of up to four chunks of slice partials (bn_small_finish_apply_kernel: every
    workgroup redoes the chunked finish of its channel group, then applies) against the two launches
    (WSIS_BN_SMALL_FUSED=0: ticketed chunk stage + apply pass): mean, var, running statistics and y bit for bit, and
    y against torch; 10,240 rows = five chunks: the two launches either way"""
    import wsis_native as _n
    lib = _n.hip()
    g = torch.Generator(device=DEV).manual_seed(M + 3 * C)
    x = torch.randn(M, C, device=DEV, generator=g) * 1.7 + 0.3
    gamma, beta = torch.rand(C, device=DEV, generator=g) + 0.5, torch.randn(C, device=DEV, generator=g)
    n_part = (M + 31) // 32
    pad = n_part * 32 - M
    xp = torch.cat([x, torch.zeros(pad, C, device=DEV)]) if pad else x
    blocks = xp.view(n_part, 32, C)
    cnt = torch.full((n_part, 1), 32.0, device=DEV)
    cnt[-1] = 32 - pad
    sm = blocks.sum(1)
    mask = (torch.arange(n_part * 32, device=DEV) < M).view(n_part, 32, 1).float()
    q = (((blocks - (sm / cnt).unsqueeze(1)) * mask) ** 2).sum(1)
    partial = torch.stack([sm, q], 1).contiguous()
    ws_bytes = lib.wsis_bn_stats_finalize_workspace_bytes(n_part, C)

    def run():
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=DEV)
        mean, var = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
        rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
        y = torch.empty_like(x)
        _n.check(lib.wsis_bn_stats_finalize_apply(_n.ptr(partial), n_part, M, C, _n.ptr(mean), _n.ptr(var), _n.ptr(rm),
                                                  _n.ptr(rv), 0.1, _n.ptr(x), _n.ptr(gamma), _n.ptr(beta), 1e-4, 1,
                                                  _n.ptr(y), _n.ptr(ws), ws_bytes, _n.ptr(_n.sync_block()), _n.stream_ptr()),
                 "finalize_apply")
        torch.cuda.synchronize()
        return mean, var, rm, rv, y

    one = run()
    again = run()
    monkeypatch.setenv("WSIS_BN_SMALL_FUSED", "0")
    two = run()
    for a, b, c in zip(one, again, two):
        assert torch.equal(a, b) and torch.equal(a, c)
    want = torch.relu(torch.nn.functional.batch_norm(x, None, None, gamma, beta, True, 0.1, 1e-4))
    assert torch.allclose(one[4], want, rtol=1e-4, atol=1e-4)
    assert not _n.sync_block()[:64 * 4096].any(), "every launch must leave its sync slot zero"


@pytest.mark.gpu
@pytest.mark.parametrize("M,C,addend", [(3300, 128, True), (1100, 160, False), (70, 32, True), (344, 160, True),
                                        (6572, 96, True), (9000, 128, False), (10007, 64, True), (4100, 32, False)])
def test_small_level_batchnorm_backward_finish_and_apply_in_one_launch(M, C, addend, monkeypatch):
    """up to four chunks of slice partials (fewer than ~10,240 rows; round 5: one chunk, 4,096 rows): wsis_bn_bwd_from_partials
    runs the reduction finish and the apply pass as ONE launch without any hand-off between workgroups
    (bn_small_bwd_finish_apply_kernel: every workgroup redoes the chunked finish of its channel group) -- dx, dgamma, dbeta
    bit for bit equal to the two launches (WSIS_BN_SMALL_FUSED=0), and dx equal to autograd through BatchNorm+ReLU"""
    import wsis_native as _n
    lib = _n.hip()
    g = torch.Generator(device=DEV).manual_seed(M * 7 + C)
    x = torch.randn(M, C, device=DEV, generator=g) * 1.3 + 0.2
    dy = torch.randn(M, C, device=DEV, generator=g)
    add = torch.randn(M, C, device=DEV, generator=g) if addend else None
    gamma, beta = torch.rand(C, device=DEV, generator=g) + 0.5, torch.randn(C, device=DEV, generator=g) * 0.3
    eps = 1e-4
    mean, var = x.mean(0), x.var(0, unbiased=False)
    xh = (x - mean) * torch.rsqrt(var + eps)
    dz = torch.where(xh * gamma + beta > 0, dy, torch.zeros_like(dy))
    n_part = (M + 31) // 32
    pad = n_part * 32 - M
    def slices(t):
        tp = torch.cat([t, torch.zeros(pad, C, device=DEV)]) if pad else t
        return tp.view(n_part, 32, C).sum(1)
    partial = torch.stack([slices(dz), slices(dz * xh)], 1).contiguous()          # [n_part, 2, C]
    ws_bytes = lib.wsis_bn_stats_finalize_workspace_bytes(n_part, C)

    def run():
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=DEV)
        dx = torch.empty_like(x)
        dg, db = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
        _n.check(lib.wsis_bn_bwd_from_partials(_n.ptr(partial), n_part, _n.ptr(x), _n.ptr(dy), _n.ptr(mean), _n.ptr(var),
                                               _n.ptr(gamma), _n.ptr(beta), eps, 1, _n.ptr(dx), _n.ptr(dg), _n.ptr(db),
                                               _n.ptr(add) if addend else None, M, C, _n.ptr(ws), ws_bytes,
                                               _n.ptr(_n.sync_block()), _n.stream_ptr()), "bn_bwd_from_partials")
        torch.cuda.synchronize()
        return dx, dg, db

    one = run()
    monkeypatch.setenv("WSIS_BN_SMALL_FUSED", "0")
    two = run()
    for a, b in zip(one, two):
        assert torch.equal(a, b)
    xr = x.double().requires_grad_(True)
    yr = torch.relu(torch.nn.functional.batch_norm(xr, None, None, gamma.double(), beta.double(), True, 0.1, eps))
    yr.backward(dy.double())
    want = xr.grad + (add.double() if addend else 0)
    assert float((one[0].double() - want).abs().max()) <= 2e-4 * float(want.abs().max())
