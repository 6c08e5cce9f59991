"""Host-side input contract (3d-wsis_amd/datasets.py) against golden vectors produced by the reference's own methods
(tests/golden/make_dataset_golden.py runs scannetv2_dataset.py's ``__getitem__``/``data_aug``/``crop``/... here), and
batch assembly against the schema of ``collate_fn``.  CPU only."""
import importlib
import os

import numpy as np
import pytest
import torch

importlib.import_module("3d-wsis_amd")
import wsis_datasets as datasets  # noqa: E402
import harness  # noqa: E402

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dataset_golden.npz"))


def test_augmentation_matrix_draws_match_reference():
    p = datasets.ScenePrep(seed=21)
    np.testing.assert_array_equal(p.data_aug(G["aug_in"], True, True, True), G["aug_out"])
    p = datasets.ScenePrep(seed=22)
    np.testing.assert_array_equal(p.data_aug(G["aug_in"], False, False, True), G["aug_rot_only"])


def test_crop_relabel_and_instance_info_match_reference():
    p = datasets.ScenePrep(max_npoint=3000, seed=23)
    xyz, valid = p.crop(G["crop_in"])
    np.testing.assert_array_equal(valid, G["crop_valid"])
    np.testing.assert_array_equal(xyz, G["crop_xyz"])
    assert valid.sum() <= 3000 < len(valid)
    cropped = p.get_cropped_inst_label(G["inst_in"].copy(), valid)
    np.testing.assert_array_equal(cropped, G["inst_cropped"])
    n, info = p.get_instance_info(G["crop_in"][valid], cropped.astype(np.int32))
    assert n == int(G["info_n"])
    np.testing.assert_array_equal(info["instance_info"], G["info"])
    np.testing.assert_array_equal(np.asarray(info["instance_pointnum"]), G["info_pointnum"])


def test_crop_rejects_negative_coordinates():
    with pytest.raises(ValueError):
        datasets.ScenePrep().crop(np.array([[1.0, -1.0, 0.0]]))


def test_elastic_matches_reference():
    p = datasets.ScenePrep(seed=24)
    np.testing.assert_allclose(p.elastic(G["elastic_in"], 6, 40.0), G["elastic_out"], rtol=0, atol=1e-9)


@pytest.mark.parametrize("tag", ["t", "c", "e"])
def test_scene_preparation_matches_reference_getitem(tag):
    aug, test_mode, max_npoint, seed = [int(x) for x in G[tag + "_cfg"]]
    sc = harness.make_scene(5, room=(1.0, 0.9, 0.8), n_box=2)
    tup, graph = datasets.synthetic_scene_to_reference_format(sc)
    prep = datasets.ScenePrep(max_npoint=max_npoint, aug=bool(aug), test_mode=bool(test_mode), seed=seed)
    scene, loc, loc_offset, loc_float, feat, sem, ins, sp, sub, inst_num, inst_info, inst_pointnum = prep(tup, graph)
    np.testing.assert_array_equal(loc.numpy(), G[tag + "_loc"])
    np.testing.assert_array_equal(loc_offset.numpy(), G[tag + "_loc_offset"])
    np.testing.assert_array_equal(loc_float.numpy(), G[tag + "_loc_float"])
    np.testing.assert_array_equal(sem.numpy(), G[tag + "_sem"])
    np.testing.assert_array_equal(ins.numpy(), G[tag + "_ins"])
    np.testing.assert_array_equal(sp.numpy(), G[tag + "_sp"])
    assert inst_num == int(G[tag + "_inst_num"])
    np.testing.assert_array_equal(inst_info.numpy(), G[tag + "_inst_info"])
    np.testing.assert_array_equal(np.asarray(inst_pointnum), G[tag + "_inst_pointnum"])
    np.testing.assert_array_equal(sub.vs["superpoint_offset_vector"], G[tag + "_g_off"])
    np.testing.assert_array_equal(sub.vs["v"], G[tag + "_g_v"])
    np.testing.assert_array_equal(sub.edges, G[tag + "_g_edges"])
    # colour jitter: same distribution, different stream (torch global RNG in the reference, a Generator here)
    assert feat.shape == G[tag + "_feat"].shape and feat.dtype == torch.float32
    if not aug:
        np.testing.assert_array_equal(feat.numpy(), G[tag + "_feat"])
    # the original graph is untouched (the reference deep-copies it, :140)
    np.testing.assert_array_equal(graph.vs["superpoint_offset_vector"], sc["sp_offset"].astype(np.float64))


def test_collate_fn_equals_harness_collate_without_augmentation():
    scenes = [harness.make_scene(s, room=(1.0, 0.9, 0.8), n_box=2) for s in (5, 6)]
    want = harness.collate(scenes)
    prep = datasets.ScenePrep(aug=False, test_mode=True)
    got = datasets.collate_fn([prep(*datasets.synthetic_scene_to_reference_format(sc)) for sc in scenes])
    # voxel coordinates: the reference truncates (xyz*scale - min(xyz*scale)) (:149-152,176), the synthetic-workload
    # collate floors xyz*scale first, so a point on a cell border may land one cell apart
    assert got["locs"].shape == want["locs"].shape and int((got["locs"] - want["locs"]).abs().max()) <= 1
    assert torch.equal(got["locs"][:, 0], want["locs"][:, 0])
    assert got["p2v_map"].shape == want["p2v_map"].shape and got["v2p_map"].dtype == want["v2p_map"].dtype
    for k in ("locs_float", "feats", "semantic_labels",
              "instance_labels", "offsets", "superpoint", "sp_batch_offsets", "edge_u_list", "edge_v_list",
              "superpoint_semantic_labels", "superpoint_offset_vector",
              "superpoint_instance_voxel_num", "superpoint_instance_size"):
        assert got[k].dtype == want[k].dtype, k
        assert torch.equal(got[k], want[k]), k
    # the reference batches the graph's per-superpoint instance ids as they are (:407) -- only the point-level ids
    # are shifted per scene (:389-391); the loss only ever compares them inside one scene (losses_3D_WSIS.py:100-111)
    raw = torch.from_numpy(np.concatenate([sc["sp_ins"] for sc in scenes])).long()
    assert torch.equal(got["superpoint_instance_labels"], raw)
    assert got["spatial_shape"].shape == (3,) and (got["spatial_shape"] >= 128).all()
    gi, wi = got["GIs"][0], want["GIs"][0]
    assert torch.equal(gi._edge_indexes, wi._edge_indexes) and torch.equal(gi._edgefeats, wi._edgefeats)
    # keys of the reference's batch dict (scannetv2_dataset.py:460-474)
    for k in ("locs_offset", "instance_info", "instance_pointnum", "is1ins_labels", "scene_list"):
        assert k in got
    assert got["locs_offset"].shape == (2, 3) and got["instance_info"].shape == (got["locs"].shape[0], 9)


def test_cropped_batch_keeps_the_contract():
    """after a crop: superpoint ids dense per scene, graph restricted to the surviving superpoints, batch offsets
    consistent, edges inside each scene's id range"""
    scenes = [harness.make_scene(s, room=(1.0, 0.9, 0.8), n_box=2) for s in (5, 6)]
    prep = datasets.ScenePrep(max_npoint=11000, aug=True, seed=4)
    b = datasets.collate_fn([prep(*datasets.synthetic_scene_to_reference_format(sc)) for sc in scenes])
    S = int(b["sp_batch_offsets"][-1])
    assert len(torch.unique(b["superpoint"])) == S == b["superpoint_semantic_labels"].shape[0]
    for i in range(2):
        lo, hi = int(b["offsets"][i]), int(b["offsets"][i + 1])
        assert hi - lo <= 11000
        s_lo, s_hi = int(b["sp_batch_offsets"][i]), int(b["sp_batch_offsets"][i + 1])
        sp = b["superpoint"][lo:hi]
        assert int(sp.min()) == s_lo and int(sp.max()) == s_hi - 1
    u, v = b["edge_u_list"], b["edge_v_list"]
    scene_of = torch.bucketize(u, b["sp_batch_offsets"][1:].long(), right=True)
    assert torch.equal(scene_of, torch.bucketize(v, b["sp_batch_offsets"][1:].long(), right=True))
    ids = b["instance_labels"][b["instance_labels"] != -100]
    assert ids.numel() == 0 or int(ids.min()) >= 0


def test_plain_graph_roundtrip_and_scene_file(tmp_path):
    sc = harness.make_scene(5, room=(1.0, 0.9, 0.8), n_box=2)
    tup, g = datasets.synthetic_scene_to_reference_format(sc)
    g.save(tmp_path / "scene_spg.npz")
    h = datasets.PlainGraph.load(tmp_path / "scene_spg.npz")
    assert sorted(h.vs) == sorted(g.vs)
    for k in g.vs:
        np.testing.assert_array_equal(h.vs[k], g.vs[k])
    np.testing.assert_array_equal(h.edges, g.edges); np.testing.assert_array_equal(h.f, g.f)
    torch.save(tup, tmp_path / "scene_inst_nostuff.pth")
    back = datasets.load_scene_file(tmp_path / "scene_inst_nostuff.pth")
    assert back[5] == "synthetic"
    for a, b in zip(back[:5], tup[:5]):
        np.testing.assert_array_equal(a, b)
    torch.save((1, 2, 3), tmp_path / "bad.pth")
    with pytest.raises(ValueError):
        datasets.load_scene_file(tmp_path / "bad.pth")


@pytest.mark.parametrize("tag", ["w1", "w2"])
def test_acquire_weak_label_matches_reference(tag):
    """datasets.acquire_weak_label against the reference's own method (scannetv2_dataset.py:970-1036) run on the same
    scene with numpy's global state seeded: same drawn superpoints, same labels kept, same offset vectors."""
    k, seed = [int(x) for x in G[tag + "_cfg"]]
    sc = harness.make_scene(9, room=(1.0, 0.9, 0.8), n_box=3)
    _, graph = datasets.synthetic_scene_to_reference_format(sc)
    graph.vs["semantic_label"] = G[tag + "_in_sem"].copy()
    graph.vs["instance_label"] = G[tag + "_in_ins"].copy()
    chosen = datasets.acquire_weak_label(sc["xyz"], G["w_sem_gt"], G["w_ins_gt"], sc["superpoint"], graph, k,
                                         rng=np.random.RandomState(seed))
    np.testing.assert_array_equal(graph.vs["semantic_label"], G[tag + "_sem"])
    np.testing.assert_array_equal(graph.vs["instance_label"], G[tag + "_ins"])
    np.testing.assert_array_equal(graph.vs["superpoint_offset_vector"], G[tag + "_off"])
    kept = np.flatnonzero(G[tag + "_ins"] != -100)
    assert sorted(chosen) == kept.tolist() and len(chosen) == len(set(chosen))
    assert len(chosen) == k * 7                      # 7 instances (no superpoint has -100 as its majority label)


def test_segment_mode_ties_take_the_smallest_value():
    seg = np.array([0, 0, 0, 0, 1, 1, 2])
    val = np.array([5.0, 3.0, 5.0, 3.0, -100.0, 7.0, 2.0])
    np.testing.assert_array_equal(datasets._segment_mode(seg, val, 4), [3.0, -100.0, 2.0, -100.0])


@pytest.mark.parametrize("tag", ["s3a", "s3b", "s3c"])
def test_s3dis_item_matches_reference_getitem(tag):
    """the S3DIS variant (s3dis_dataset.py: random quarter of the points per training item :135-144, block crop
    ``crop_v2`` :285-319) against the reference's own ``__getitem__`` run with numpy's global state seeded"""
    sub, max_npoint, seed = [int(x) for x in G[tag + "_cfg"]]
    sc = harness.make_scene(5, room=(1.0, 0.9, 0.8), n_box=2)
    tup, graph = datasets.synthetic_scene_to_reference_format(sc)
    prep = datasets.ScenePrep(max_npoint=max_npoint, aug=True, test_mode=False, seed=seed, crop_version=2,
                              subsample_train=bool(sub))
    scene, loc, loc_offset, loc_float, feat, sem, ins, sp, g, inst_num, inst_info, inst_pointnum = prep(tup, graph)
    np.testing.assert_array_equal(loc.numpy(), G[tag + "_loc"])
    np.testing.assert_array_equal(loc_offset.numpy(), G[tag + "_loc_offset"])
    np.testing.assert_array_equal(loc_float.numpy(), G[tag + "_loc_float"])
    np.testing.assert_array_equal(sem.numpy(), G[tag + "_sem"])
    np.testing.assert_array_equal(ins.numpy(), G[tag + "_ins"])
    np.testing.assert_array_equal(sp.numpy(), G[tag + "_sp"])
    assert inst_num == int(G[tag + "_inst_num"])
    np.testing.assert_array_equal(inst_info.numpy(), G[tag + "_inst_info"])
    np.testing.assert_array_equal(g.vs["superpoint_offset_vector"], G[tag + "_g_off"])
    np.testing.assert_array_equal(g.vs["v"], G[tag + "_g_v"])
    np.testing.assert_array_equal(g.edges, G[tag + "_g_edges"])
    assert loc.shape[0] <= max_npoint and (loc.numpy() >= 0).all()
