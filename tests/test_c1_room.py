"""BASELINE.json configs[0] (SURVEY 8d "C1"): a single 10 k-pt synthetic room through voxelize -> superpoint
scatter-mean -> edge affinity -> dense affinity matrix.

* CPU (not gpu): the host operator of libwsis_host.so (the product path of ``voxelization_idx`` inside DataLoader
  workers) against the oracle and against ``np.unique``, on exactly the C1 room bench.py times on the host.
* GPU: the HIP operators on the same room against the oracle (integer outputs bit-exact, fp32 at 1e-5 / 1e-4).
"""
import numpy as np
import pytest
import torch

import harness
import pointgroup_ops
from oracle import affinity_ref, pg_ops, scatter_ref

C1_ROOM, C1_POINTS = (3.0, 3.0, 2.4), 10000


def _room():
    sc = harness.make_scene(0, room=C1_ROOM, n_box=2, max_points=C1_POINTS)
    return harness.collate([sc])


def _unique_first_occurrence(coords):
    u, first, inv = np.unique(coords, axis=0, return_index=True, return_inverse=True)
    order = np.argsort(first)
    rank = np.empty_like(order)
    rank[order] = np.arange(len(order))
    return u[order], rank[inv.ravel()]


def test_c1_host_voxelizer_matches_oracle_and_unique():
    b = _room()
    assert b["locs"].shape[0] == C1_POINTS
    coords = b["locs"].numpy()
    rl, rp, rv = pg_ops.voxelization_idx(coords, 1, 4)
    # collate() already ran the host operator: its outputs ARE the batch dict entries
    assert np.array_equal(b["voxel_locs"].numpy(), rl)
    assert np.array_equal(b["p2v_map"].numpy(), rp)
    assert np.array_equal(b["v2p_map"].numpy(), rv)
    ul, up = _unique_first_occurrence(coords)
    assert np.array_equal(rl, ul) and np.array_equal(rp, up)
    # the rest of the C1 plumbing on the host (oracle functions): shapes and basic invariants
    feats = torch.cat([b["feats"], b["locs_float"]], 1)
    vf = pg_ops.voxelization(feats.numpy(), rv, 4)
    assert vf.shape == (rl.shape[0], 6) and np.isfinite(vf).all()
    pooled = scatter_ref.scatter(torch.from_numpy(vf)[torch.from_numpy(rp).long()], b["superpoint"], reduce="mean")
    S = int(b["sp_batch_offsets"][-1])
    assert pooled.shape == (S, 6)


@pytest.mark.gpu
def test_c1_room_on_the_gpu_matches_oracle():
    import torch_scatter
    import wsis_ops
    dev = "cuda"
    b = _room()
    coords = b["locs"]
    rl, rp, rv = pg_ops.voxelization_idx(coords.numpy(), 1, 4)
    # device voxelizer: bit-exact against the oracle and np.unique
    d_locs, d_p2v, d_v2p = pointgroup_ops.voxelization_idx(coords.to(dev), 1, 4)
    assert np.array_equal(d_locs.cpu().numpy(), rl) and np.array_equal(d_p2v.cpu().numpy(), rp)
    assert np.array_equal(d_v2p.cpu().numpy(), rv)
    ul, up = _unique_first_occurrence(coords.numpy())
    assert np.array_equal(d_locs.cpu().numpy(), ul) and np.array_equal(d_p2v.cpu().numpy(), up)
    # voxelization (sequential fp32 mean): bit-exact
    feats = torch.cat([b["feats"], b["locs_float"]], 1)
    vf = pointgroup_ops.voxelization(feats.to(dev), d_v2p, 4)
    ref_vf = pg_ops.voxelization(feats.numpy(), rv, 4)
    assert np.array_equal(vf.cpu().numpy(), ref_vf)
    # superpoint scatter-mean of the point features
    S = int(b["sp_batch_offsets"][-1])
    pf = vf[d_p2v.long()]
    pooled = torch_scatter.scatter(pf, b["superpoint"].to(dev), dim=0, reduce="mean")
    ref_pooled = scatter_ref.scatter(torch.from_numpy(ref_vf)[torch.from_numpy(rp).long()], b["superpoint"],
                                     reduce="mean")
    assert pooled.shape == (S, 6)
    assert torch.allclose(pooled.cpu(), ref_pooled, rtol=1e-5, atol=1e-6)
    # edge affinity attention + dense matrix
    rng = np.random.default_rng(0)
    q, k, v = (torch.from_numpy(rng.standard_normal((S, 64)).astype("float32")) for _ in range(3))
    E = int(b["edge_u_list"].shape[0])
    pos = torch.from_numpy(rng.standard_normal(E).astype("float32"))
    ref_aff, ref_res = affinity_ref.edge_affinity(q, k, v, pos, b["edge_u_list"], b["edge_v_list"])
    graph = wsis_ops.EdgeGraph(b["edge_u_list"].to(dev), b["edge_v_list"].to(dev), S)
    aff, res = wsis_ops.edge_affinity(q.to(dev), k.to(dev), v.to(dev), pos.to(dev), graph, 1.0 / np.sqrt(64))
    assert torch.allclose(aff.cpu(), ref_aff, rtol=1e-4, atol=1e-6)
    assert torch.allclose(res.cpu()[:ref_res.shape[0]], ref_res, rtol=1e-4, atol=1e-5)
    A = wsis_ops.affinity_matrix(b["edge_u_list"].to(dev), b["edge_v_list"].to(dev), aff.double(), S)
    ref_A = affinity_ref.affinity_matrix(b["edge_u_list"].numpy(), b["edge_v_list"].numpy(),
                                         aff.double().cpu().numpy(), S)
    assert np.array_equal(A.cpu().numpy(), ref_A)
