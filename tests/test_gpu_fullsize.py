"""GPU: BASELINE.json's full sizes (C2 ~150 k voxels, C3 batch of 4, C4 ~1 M points) through size-independent
properties -- the CPU oracle cannot finish these sizes in seconds, so the checks are algebraic:

  rulebook    offset-flip symmetry of the SubM table, transposition of the coupled down/up tables, strictly
              ascending linear index of strided-conv outputs, every fine voxel maps to its parent
  conv        linearity in the input and in the weights, dIn = adjoint of the forward (<conv(x), y> = <x, dIn(y)>),
              dW consistent with the same bilinear form, bit-exact determinism
  scatter     mean * count == sum, max >= mean, gather(scatter_sum) adjoint identity
  voxelize    p2v/v2p consistency, mean of a constant is the constant
"""
import numpy as np
import pytest
import torch

import harness
import pointgroup_ops
import spconv
import torch_scatter
from spconv import ops

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def c2_batch():
    scene = harness.make_scene(1)
    return harness.collate([scene])


@pytest.fixture(scope="module")
def c3_batch():
    # BASELINE configs[2] as bench.py runs it (c3_batch4_train): four C2-sized scenes, seeds 1-4, ~625 k voxels
    scenes = [harness.bench_scene(s) for s in (1, 2, 3, 4)]
    return harness.collate(scenes)


def _check_subm(indices, shape):
    rb = ops.build_subm_rulebook(indices, shape, [3, 3, 3], [1, 1, 1])
    nbr = rb.nbr
    K, M = nbr.shape
    ar = torch.arange(M, device=DEV, dtype=torch.int32)
    assert torch.equal(nbr[13], ar), "centre offset pairs every voxel with itself"
    for k in range(K):
        o = torch.nonzero(nbr[k] >= 0).flatten()
        i = nbr[k][o].long()
        assert torch.equal(nbr[K - 1 - k][i].long(), o), f"offset {k}: table must be symmetric under the flip"
        # geometry: coord[i] - coord[o] == kappa - 1, same batch
        kappa = torch.tensor([k // 9 - 1, (k // 3) % 3 - 1, k % 3 - 1], device=DEV)
        d = indices[i].long() - indices[o].long()
        assert int(d[:, 0].abs().max() if len(o) else 0) == 0
        assert len(o) == 0 or bool((d[:, 1:] == kappa).all())
    assert sorted(rb.order.cpu().tolist()) == list(range(M))
    assert torch.equal(rb.nbr_p, nbr[:, rb.order.long()])
    return rb


def _check_down(indices, shape):
    rb = ops.build_down_rulebook(indices, shape, [2] * 3, [2] * 3, [0] * 3)
    out = rb.out_indices.long()
    S = rb.out_shape
    lin = ((out[:, 0] * S[0] + out[:, 1]) * S[1] + out[:, 2]) * S[2] + out[:, 3]
    assert bool((lin[1:] > lin[:-1]).all()), "strided-conv outputs in strictly ascending linear index"
    up, down = rb.nbr_up, rb.nbr
    per_fine = (up >= 0).sum(0)
    assert int(per_fine.max()) <= 1, "k2 s2: every fine voxel has at most one parent"
    inside = ((indices[:, 1:].long() // 2) < torch.tensor(S, device=DEV)).all(1)
    assert torch.equal(per_fine == 1, inside), "exactly the voxels inside the floor-divided extent have a parent"
    for k in range(8):
        f = torch.nonzero(up[k] >= 0).flatten()
        o = up[k][f].long()
        assert torch.equal(down[k][o].long(), f), "down table is the transpose of the up table"
        assert torch.equal(indices[f, 1:].long() // 2, out[o, 1:]) and torch.equal(indices[f, 0].long(), out[o, 0])
        kappa = torch.tensor([k // 4, (k // 2) % 2, k % 2], device=DEV)
        assert len(f) == 0 or bool((indices[f, 1:].long() % 2 == kappa).all())
    assert int((down >= 0).sum()) == int(inside.sum())
    return rb


def _dot(a, b):
    return float((a.double() * b.double()).sum())


def _check_conv_algebra(rb_nbr, rb_order, M_in, M_out, K, cin, cout, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    x1, x2 = (torch.randn(M_in, cin, generator=g).to(DEV) for _ in range(2))
    W1, W2 = (torch.randn(K, cin, cout, generator=g).to(DEV) * 0.1 for _ in range(2))
    y = torch.randn(M_out, cout, generator=g).to(DEV)
    f = lambda x, W: ops._conv(x, rb_nbr, rb_order, W, None, None, M_out)
    o11 = f(x1, W1)
    assert torch.equal(o11, f(x1, W1)), "bit-exact run-to-run"
    scale = float(o11.abs().max()) + 1e-6
    assert float((f(x1 + 2 * x2, W1) - (o11 + 2 * f(x2, W1))).abs().max()) < 2e-4 * scale, "linear in x"
    assert float((f(x1, W1 - 3 * W2) - (o11 - 3 * f(x1, W2))).abs().max()) < 2e-4 * scale, "linear in W"
    return x1, W1, y, o11


def test_c2_full_scene_rulebooks_and_conv_properties(c2_batch):
    b = c2_batch
    idx = b["voxel_locs"].int().to(DEV).contiguous()
    shape = [int(s) for s in b["spatial_shape"]]
    M = idx.shape[0]
    assert 140_000 < M < 175_000, M
    rb = _check_subm(idx, shape)
    x1, W1, y, o = _check_conv_algebra(rb.nbr_p, rb.order, M, M, 27, 32, 32, 0)
    # adjoint: <conv(x,W), y> == <x, dIn(y)> with dIn = conv(y, same table, flipped transposed weights)
    WT = ops._weight_t(W1, 1)
    dx = ops._conv(y, rb.nbr_p, rb.order, WT, None, None, M)
    lhs, rhs = _dot(o, y), _dot(x1, dx)
    assert abs(lhs - rhs) < 1e-4 * max(abs(lhs), 1.0)
    dW = ops._dw(x1, rb.nbr_p, rb.order, y, 27, 32, 32)
    assert abs(_dot(dW, W1) - lhs) < 1e-4 * max(abs(lhs), 1.0), "<dW, W> equals the same bilinear form"
    assert torch.equal(dW, ops._dw(x1, rb.nbr_p, rb.order, y, 27, 32, 32)), "dW bit-exact run-to-run (no atomics)"
    # strided level
    rd = _check_down(idx, shape)
    Mo = rd.out_indices.shape[0]
    xd, Wd, yd, od = _check_conv_algebra(rd.nbr_p, rd.order, M, Mo, 8, 32, 64, 1)
    WTd = ops._weight_t(Wd, 0)
    dxd = ops._conv(yd, rd.nbr_up_p, rd.order_up, WTd, None, None, M)
    lhs, rhs = _dot(od, yd), _dot(xd, dxd)
    assert abs(lhs - rhs) < 1e-4 * max(abs(lhs), 1.0)


def test_c3_batch_of_four_never_mixes_scenes(c3_batch):
    b = c3_batch
    idx = b["voxel_locs"].int().to(DEV).contiguous()
    shape = [int(s) for s in b["spatial_shape"]]
    rb = ops.build_subm_rulebook(idx, shape, [3, 3, 3], [1, 1, 1])
    for k in (0, 5, 13, 20, 26):
        o = torch.nonzero(rb.nbr[k] >= 0).flatten()
        assert torch.equal(idx[rb.nbr[k][o].long(), 0], idx[o, 0]), "pairs stay inside a batch item"
    rd = ops.build_down_rulebook(idx, shape, [2] * 3, [2] * 3, [0] * 3)
    assert set(rd.out_indices[:, 0].unique().cpu().tolist()) == {0, 1, 2, 3}


def test_c2_voxelization_and_scatter_properties(c2_batch):
    b = c2_batch
    N, M = b["locs"].shape[0], b["voxel_locs"].shape[0]
    p2v, v2p = b["p2v_map"], b["v2p_map"]
    assert int(v2p[:, 0].sum()) == N and int(p2v.max()) == M - 1
    cnt = torch.bincount(p2v.long(), minlength=M)
    assert torch.equal(cnt.int(), v2p[:, 0])
    first = v2p[:, 1].long()
    assert torch.equal(b["voxel_locs"], b["locs"][first]), "voxel coords = coords of the voxel's first point"
    const = torch.full((N, 6), 0.37, device=DEV)
    out = pointgroup_ops.voxelization(const, v2p.to(DEV), 4)
    assert float((out - 0.37).abs().max()) < 1e-6
    feats = torch.randn(N, 32, device=DEV)
    sp = b["superpoint"].to(DEV)
    csr = torch_scatter.SegmentCSR(sp)
    s = torch_scatter.scatter(feats, sp, 0, reduce="sum", csr=csr)
    m = torch_scatter.scatter(feats, sp, 0, reduce="mean", csr=csr)
    mx = torch_scatter.scatter(feats, sp, 0, reduce="max", csr=csr)
    counts = torch.bincount(sp, minlength=s.shape[0]).float().unsqueeze(1)
    assert float((m * counts - s).abs().max()) < 1e-3 and bool((mx >= m - 1e-6).all())
    y = torch.randn_like(s)
    assert abs(_dot(s, y) - _dot(feats, y[sp])) < 1e-4 * abs(_dot(s, y)), "gather is the adjoint of scatter-sum"


def test_c4_one_million_points_inference_slice():
    """S3DIS-room sized input (~1 M points): host voxelization, all rulebooks of the pyramid, one forward of the
    first two layers; checks shapes, table symmetry on samples and determinism, and that nothing overflows."""
    scene = harness.make_scene(5, room=(13.0, 10.0, 3.0), n_box=36)
    b = harness.collate([scene])
    N, M = b["locs"].shape[0], b["voxel_locs"].shape[0]
    assert N > 900_000 and M > 600_000, (N, M)
    idx = b["voxel_locs"].int().to(DEV).contiguous()
    shape = [int(s) for s in b["spatial_shape"]]
    t = spconv.SparseConvTensor(torch.randn(M, 6, device=DEV), idx, b["spatial_shape"], 1)
    ops.prebuild_unet_rulebooks(t, 5)
    sizes = []
    for lvl in range(1, 6):
        rb = t.indice_dict[f"subm{lvl}"]
        K, Ml = rb.nbr.shape
        sizes.append(Ml)
        for k in (0, 13, 26):
            o = torch.nonzero(rb.nbr[k] >= 0).flatten()[:50000]
            assert torch.equal(rb.nbr[K - 1 - k][rb.nbr[k][o].long()].long(), o)
    assert sizes[0] == M and all(a > b_ for a, b_ in zip(sizes, sizes[1:]))
    conv1 = spconv.SubMConv3d(6, 32, 3, padding=1, bias=False, indice_key="subm1").to(DEV)
    conv2 = spconv.SubMConv3d(32, 32, 3, padding=1, bias=False, indice_key="subm1").to(DEV)
    with torch.no_grad():
        o1 = conv2(conv1(t)).features
        o2 = conv2(conv1(t)).features
    assert o1.shape == (M, 32) and torch.isfinite(o1).all() and torch.equal(o1, o2)


# ---------------------------------------------------------------- full-size VALUE checks against the oracle's tables
def _oracle_rows_check(indices_host, shape, cin, cout, n_rows, seed, kind="subm"):
    """sparse conv forward / dIn / two offsets of dW of a full-size level against an fp64 gather-GEMM whose pair lists
    come from the ORACLE (oracle/spconv_ref.subm_pairs_fast / down_pairs_fast: numpy, independent of csrc/rulebook.hip),
    on ``n_rows`` random output rows; tolerance 1e-5 of the tensor scale (fp32 fma chains of <= 27 * cin terms)."""
    from oracle import spconv_ref as ref
    g = torch.Generator().manual_seed(seed)
    idx_d = torch.from_numpy(indices_host).to(DEV)
    M = indices_host.shape[0]
    if kind == "subm":
        pairs = ref.subm_pairs_fast(indices_host, shape, 3, 1)
        mod = spconv.SubMConv3d(cin, cout, 3, padding=1, bias=False, indice_key="v").to(DEV)
        M_out = M
    else:
        out_idx, out_shape, pairs = ref.down_pairs_fast(indices_host, shape, 2, 2, 0)
        mod = spconv.SparseConv3d(cin, cout, 2, stride=2, bias=False, indice_key="v").to(DEV)
        M_out = out_idx.shape[0]
    K = len(pairs)
    x = torch.randn(M, cin, generator=g).to(DEV).requires_grad_(True)
    t = spconv.SparseConvTensor(x, idx_d, np.array(shape), int(indices_host[:, 0].max()) + 1)
    out = mod(t)
    y = out.features
    assert y.shape == (M_out, cout)
    if kind != "subm":      # strided conv: output rows in ascending linear index = the oracle's order
        assert np.array_equal(out.indices.cpu().numpy(), np.asarray(out_idx, dtype=np.int32))
    gy = torch.randn(M_out, cout, generator=g).to(DEV)
    y.backward(gy)
    W = mod.weight.detach().view(K, cin, cout).double()
    xd, gyd = x.detach().double(), gy.double()
    rows_o = torch.randint(0, M_out, (n_rows,), generator=g).to(DEV)
    rows_i = torch.randint(0, M, (n_rows,), generator=g).to(DEV)
    want_y = torch.zeros(n_rows, cout, dtype=torch.float64, device=DEV)
    want_dx = torch.zeros(n_rows, cin, dtype=torch.float64, device=DEV)
    pos_o = torch.full((M_out,), -1, dtype=torch.long, device=DEV)
    pos_o[rows_o] = torch.arange(n_rows, device=DEV)           # duplicates: the last position wins; handled below
    pos_i = torch.full((M,), -1, dtype=torch.long, device=DEV)
    pos_i[rows_i] = torch.arange(n_rows, device=DEV)
    dw_checked = 0
    for k, (pi, po) in enumerate(pairs):
        pi_d, po_d = torch.from_numpy(np.asarray(pi)).to(DEV), torch.from_numpy(np.asarray(po)).to(DEV)
        sel = pos_o[po_d] >= 0
        want_y.index_add_(0, pos_o[po_d[sel]], xd[pi_d[sel]] @ W[k])
        sel = pos_i[pi_d] >= 0
        want_dx.index_add_(0, pos_i[pi_d[sel]], gyd[po_d[sel]] @ W[k].t())
        if k in (0, K // 2):
            dw = xd[pi_d].t() @ gyd[po_d]
            got = mod.weight.grad.view(K, cin, cout)[k].double()
            assert float((got - dw).abs().max()) <= 1e-5 * max(float(dw.abs().max()), 1.0), f"dW offset {k}"
            dw_checked += 1
    assert dw_checked == 2
    uo, ui = pos_o[rows_o], pos_i[rows_i]                          # the surviving position of every sampled row
    sy, sx = float(want_y.abs().max()), float(want_dx.abs().max())
    assert float((y.detach()[rows_o].double() - want_y[uo]).abs().max()) <= 1e-5 * max(sy, 1.0), "forward values"
    assert float((x.grad[rows_i].double() - want_dx[ui]).abs().max()) <= 1e-5 * max(sx, 1.0), "dIn values"


def _level_indices(batch, level):
    idx = batch["voxel_locs"].int().to(DEV).contiguous()
    shape = [int(s) for s in batch["spatial_shape"]]
    for _ in range(level):
        rd = ops.build_down_rulebook(idx, shape, [2] * 3, [2] * 3, [0] * 3)
        idx, shape = rd.out_indices, rd.out_shape
    return idx.cpu().numpy().astype(np.int32), shape


@pytest.mark.parametrize("level,cin,cout", [(0, 32, 32), (1, 64, 64), (2, 96, 96), (3, 128, 128), (4, 160, 160), (0, 64, 32)])
def test_c2_conv_values_against_oracle_tables(c2_batch, level, cin, cout):
    idx, shape = _level_indices(c2_batch, level)
    _oracle_rows_check(idx, shape, cin, cout, 4096, 100 + level)


@pytest.mark.parametrize("level,cin,cout", [(0, 32, 64), (2, 96, 128)])
def test_c2_strided_conv_values_against_oracle_tables(c2_batch, level, cin, cout):
    idx, shape = _level_indices(c2_batch, level)
    _oracle_rows_check(idx, shape, cin, cout, 4096, 200 + level, kind="down")


@pytest.mark.parametrize("level,cin,cout", [(0, 32, 32), (1, 64, 64)])
def test_c3_conv_values_against_oracle_tables(c3_batch, level, cin, cout):
    idx, shape = _level_indices(c3_batch, level)
    _oracle_rows_check(idx, shape, cin, cout, 4096, 300 + level)


def test_c4_conv_values_against_oracle_tables():
    scene = harness.make_scene(5, room=(13.0, 10.0, 3.0), n_box=36)
    b = harness.collate([scene])
    idx, shape = _level_indices(b, 0)
    assert idx.shape[0] > 600000
    _oracle_rows_check(idx, shape, 32, 32, 4096, 400)
    idx1, shape1 = _level_indices(b, 1)
    _oracle_rows_check(idx1, shape1, 64, 64, 4096, 401)
