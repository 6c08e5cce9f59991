"""CPU: host operators of libwsis_host.so (the product path for voxelization_idx / bfs_cluster) against the
oracle and against independent constructions (np.unique, scipy csgraph)."""
import numpy as np
import pytest
import torch

import pointgroup_ops
from oracle import pg_ops as ref


def _coords(seed, N, B=2, extent=12, dup=0.3):
    rng = np.random.default_rng(seed)
    c = np.concatenate([rng.integers(0, B, (N, 1)), rng.integers(0, extent, (N, 3))], 1).astype(np.int64)
    return c


@pytest.mark.parametrize("N", [0, 1, 7, 2000])
def test_voxelization_idx_matches_oracle_and_unique(N):
    coords = _coords(0, N)
    locs, p2v, v2p = pointgroup_ops.voxelization_idx(torch.from_numpy(coords), 2, 4)
    assert locs.dtype == torch.int64 and p2v.dtype == torch.int32 and v2p.dtype == torch.int32
    rl, rp, rv = ref.voxelization_idx(coords, 2, 4)
    assert np.array_equal(locs.numpy(), rl) and np.array_equal(p2v.numpy(), rp) and np.array_equal(v2p.numpy(), rv)
    if N:
        # independent: np.unique re-ordered by first occurrence
        u, first, inv = np.unique(coords, axis=0, return_index=True, return_inverse=True)
        order = np.argsort(first)
        rank = np.empty_like(order)
        rank[order] = np.arange(len(order))
        assert np.array_equal(p2v.numpy(), rank[inv.ravel()])
        assert np.array_equal(locs.numpy(), u[order])
        cnt = np.bincount(p2v.numpy())
        assert np.array_equal(v2p[:, 0].numpy(), cnt) and v2p.shape[1] == 1 + cnt.max()


def test_voxelization_idx_ragged_max_active_and_large_values():
    coords = np.array([[0, 5, 5, 5]] * 9 + [[1, 5, 5, 5]] + [[0, 2**40, -3, 7]] * 2, dtype=np.int64)
    locs, p2v, v2p = pointgroup_ops.voxelization_idx(torch.from_numpy(coords), 2, 4)
    assert locs.shape == (3, 4) and v2p.shape == (3, 10)
    assert v2p[0].tolist() == [9] + list(range(9))
    assert v2p[1].tolist() == [1, 9] + [0] * 8
    assert v2p[2].tolist() == [2, 10, 11] + [0] * 7
    assert locs[2].tolist() == [0, 2**40, -3, 7]


def test_voxelization_idx_rejects_cuda_mode():
    with pytest.raises(NotImplementedError):
        pointgroup_ops.voxelization_idx(torch.zeros((3, 4), dtype=torch.int64), 1, 0)


def _ball_lists(seed, N, B=2):
    rng = np.random.default_rng(seed)
    sizes = [N // B] * B
    sizes[-1] += N - sum(sizes)
    xyz = rng.random((N, 3)).astype(np.float32) * 0.5
    batch_idx = np.repeat(np.arange(B), sizes).astype(np.int32)
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    idx, sl = ref.ballquery_batch_p(xyz, batch_idx, off, 0.06)
    sem = rng.integers(0, 3, N).astype(np.int32)
    return xyz, batch_idx, off, idx, sl, sem


@pytest.mark.parametrize("threshold", [1, 5, 50])
def test_bfs_cluster_matches_oracle_and_csgraph(threshold):
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components
    N = 1500
    xyz, bi, off, idx, sl, sem = _ball_lists(3, N)
    ci, co = pointgroup_ops.bfs_cluster(torch.from_numpy(sem), torch.from_numpy(idx), torch.from_numpy(sl), threshold)
    ri, ro = ref.bfs_cluster(sem, idx, sl, threshold)
    assert np.array_equal(ci.numpy(), ri) and np.array_equal(co.numpy(), ro)
    # independent membership check: components of the same-label ball graph
    rows = np.repeat(np.arange(N), sl[:, 1])
    keep = sem[rows] == sem[idx]
    g = coo_matrix((np.ones(keep.sum()), (rows[keep], idx[keep])), shape=(N, N))
    ncomp, lab = connected_components(g, directed=False)
    sizes = np.bincount(lab, minlength=ncomp)
    assert (sizes >= threshold).sum() == len(ro) - 1
    for c in range(len(ro) - 1):
        members = ci[ro[c]:ro[c + 1], 1].numpy()
        assert len(set(lab[members])) == 1 and sizes[lab[members[0]]] == len(members)
        assert (ci[ro[c]:ro[c + 1], 0] == c).all()
    # clusters ordered by smallest member (seed order)
    firsts = [int(ci[ro[c], 1]) for c in range(len(ro) - 1)]
    assert firsts == sorted(firsts)


def test_bfs_cluster_empty():
    ci, co = pointgroup_ops.bfs_cluster(torch.zeros(0, dtype=torch.int32), torch.zeros(0, dtype=torch.int32),
                                        torch.zeros((0, 2), dtype=torch.int32), 5)
    assert ci.shape == (0, 2) and co.tolist() == [0]
