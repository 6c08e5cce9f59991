"""CPU: the spconv oracle (oracle/spconv_ref.py) against an INDEPENDENT dense construction:
F.conv3d / F.conv_transpose3d in fp64 on the densified grid, sampled at the active sites."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import spconv_ref as ref
from tests.util import dense_from_sparse, random_sparse_coords


def _w(seed, k, cin, cout):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(list(k) + [cin, cout], generator=g, dtype=torch.float64)


@pytest.mark.parametrize("ksize,pad", [((3, 3, 3), (1, 1, 1)), ((1, 3, 3), (0, 1, 1)), ((3, 1, 3), (1, 0, 1))])
def test_subm_matches_dense_conv3d(ksize, pad):
    B, shape = 2, (9, 8, 7)
    idx = random_sparse_coords(1, B, shape, 0.3)
    M = idx.shape[0]
    x = torch.randn(M, 5, dtype=torch.float64, generator=torch.Generator().manual_seed(2))
    w = _w(3, ksize, 5, 4)
    pairs = ref.subm_pairs(idx, shape, ksize, pad)
    out = ref.pairs_conv(x, w, pairs, M)
    dense = F.conv3d(dense_from_sparse(idx, x, B, shape), w.permute(4, 3, 0, 1, 2), padding=pad)
    ii = torch.as_tensor(idx).long()
    expect = dense[ii[:, 0], :, ii[:, 1], ii[:, 2], ii[:, 3]]
    assert torch.allclose(out, expect, atol=1e-12)


@pytest.mark.parametrize("shape", [(8, 8, 8), (9, 7, 8)])
def test_down_and_inverse_match_dense(shape):
    """k2 s2: output set = sites with >= 1 active input, ascending linear index; odd extents drop the
    last plane (SURVEY App. A.1 'edge effect')."""
    B = 2
    idx = random_sparse_coords(4, B, shape, 0.25)
    M = idx.shape[0]
    x = torch.randn(M, 3, dtype=torch.float64, generator=torch.Generator().manual_seed(5))
    w = _w(6, (2, 2, 2), 3, 6)
    out_idx, out_shape, pairs = ref.down_pairs(idx, shape, 2, 2, 0)
    assert out_shape == [(s - 2) // 2 + 1 for s in shape]
    out = ref.pairs_conv(x, w, pairs, out_idx.shape[0])
    dense = F.conv3d(dense_from_sparse(idx, x, B, shape), w.permute(4, 3, 0, 1, 2), stride=2)
    oi = torch.as_tensor(out_idx)
    assert torch.allclose(out, dense[oi[:, 0], :, oi[:, 1], oi[:, 2], oi[:, 3]], atol=1e-12)
    # active set == sites reachable from an active input, sorted by linear index
    occ = F.conv3d(dense_from_sparse(idx, torch.ones(M, 1), B, shape), torch.ones(1, 1, 2, 2, 2, dtype=torch.float64),
                   stride=2)[:, 0]
    assert int((occ > 0).sum()) == out_idx.shape[0]
    lin = ((out_idx[:, 0] * out_shape[0] + out_idx[:, 1]) * out_shape[1] + out_idx[:, 2]) * out_shape[2] + out_idx[:, 3]
    assert np.all(np.diff(lin) > 0)
    # dense conv is exactly zero outside the sparse output set
    mask = torch.zeros_like(dense[:, 0], dtype=torch.bool)
    mask[oi[:, 0], oi[:, 1], oi[:, 2], oi[:, 3]] = True
    assert float(dense.abs().sum(1)[~mask].max()) == 0.0
    # inverse conv: transposed conv sampled at the encoder's active sites
    y = torch.randn(out_idx.shape[0], 6, dtype=torch.float64, generator=torch.Generator().manual_seed(7))
    wi = _w(8, (2, 2, 2), 6, 3)
    back = ref.pairs_conv(y, wi, ref.inverse_pairs(pairs), M)
    dense_t = F.conv_transpose3d(dense_from_sparse(out_idx, y, B, out_shape), wi.permute(3, 4, 0, 1, 2), stride=2)
    ii = torch.as_tensor(idx).long()
    # inputs on a dropped last plane lie outside the transposed conv's extent and get zero rows
    expect = torch.zeros(M, 3, dtype=torch.float64)
    inside = (ii[:, 1] < dense_t.shape[2]) & (ii[:, 2] < dense_t.shape[3]) & (ii[:, 3] < dense_t.shape[4])
    expect[inside] = dense_t[ii[inside, 0], :, ii[inside, 1], ii[inside, 2], ii[inside, 3]]
    assert torch.allclose(back, expect, atol=1e-12)
    if any(s % 2 for s in shape):
        assert int((~inside).sum()) > 0 and float(back[~inside].abs().max()) == 0.0


def test_general_strided_conv_k3_s2_p1():
    B, shape = 1, (7, 6, 8)
    idx = random_sparse_coords(9, B, shape, 0.3)
    M = idx.shape[0]
    x = torch.randn(M, 2, dtype=torch.float64, generator=torch.Generator().manual_seed(1))
    w = _w(2, (3, 3, 3), 2, 3)
    out_idx, out_shape, pairs = ref.down_pairs(idx, shape, 3, 2, 1)
    out = ref.pairs_conv(x, w, pairs, out_idx.shape[0])
    dense = F.conv3d(dense_from_sparse(idx, x, B, shape), w.permute(4, 3, 0, 1, 2), stride=2, padding=1)
    oi = torch.as_tensor(out_idx)
    assert list(dense.shape[2:]) == out_shape
    assert torch.allclose(out, dense[oi[:, 0], :, oi[:, 1], oi[:, 2], oi[:, 3]], atol=1e-12)


def test_pair_table_roundtrip():
    idx = random_sparse_coords(3, 2, (6, 6, 6), 0.3)
    pairs = ref.subm_pairs(idx, (6, 6, 6), 3, 1)
    nbr = ref.pairs_to_table(pairs, idx.shape[0])
    # centre offset pairs i <-> i; table symmetric under offset flip
    assert np.array_equal(nbr[13], np.arange(idx.shape[0]))
    K = 27
    for k in range(K):
        o = np.nonzero(nbr[k] >= 0)[0]
        assert np.array_equal(nbr[K - 1 - k][nbr[k][o]], o)


# ---------------------------------------------------------------- host-side voxel counts of the strided levels
@pytest.mark.parametrize("shape,seed", [((9, 8, 7), 0), ((16, 16, 16), 1), ((13, 21, 6), 2), ((33, 5, 17), 3)])
def test_level_voxel_counts_equal_the_oracle_pyramid(shape, seed):
    """spconv.ops.level_voxel_counts (what a loader puts in the batch so that the device rulebook build does not read
    the counts back) against the oracle's SparseConv3d k2 s2 p0 chain, incl. odd extents (the last plane has no
    output voxel) and several batch items."""
    import importlib
    importlib.import_module("3d-wsis_amd")
    import spconv
    idx = random_sparse_coords(seed, batch=3, shape=shape, density=0.25)
    levels = 4
    got = spconv.ops.level_voxel_counts(idx, shape, levels)
    want, cur, cur_shape = [], idx.astype(np.int64), list(shape)
    for _ in range(levels - 1):
        out_idx, out_shape, _ = ref.down_pairs_fast(cur, cur_shape, 2, 2, 0)
        want.append(len(out_idx))
        cur, cur_shape = out_idx, list(out_shape)
    assert got == want and len(got) == levels - 1
