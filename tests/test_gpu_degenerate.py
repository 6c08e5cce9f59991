"""GPU: degenerate scenes through the whole path against the oracle -- fewer voxels than one 32-row MFMA slice at every
UNet level (2 voxels at the deepest one), superpoints of a single point, a batch that mixes such a scene with a normal
one, and a scene whose points all fall into ONE voxel (eval mode: BatchNorm over one row has no batch statistics, the
reference's torch.nn.BatchNorm1d raises there in training mode too)."""
import numpy as np
import pytest
import torch

import harness
from oracle import network_ref

pytestmark = pytest.mark.gpu

KEYS = ("semantic_scores", "sp_semantic_scores", "pred_sp_offset_vectors", "pred_sp_occupancy", "pred_sp_ins_size",
        "edge_affinity", "sp_discriminative_feats")


def _rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).norm() / (b.norm() + 1e-12))


def _scene(seed, centres, n_per=14, spread=0.05):
    """a few points around each centre; every cluster is one instance made of two superpoints (one of them may hold a
    single point), superpoints of one cluster are connected, clusters are chained"""
    rng = np.random.default_rng(seed)
    xyz, sp, inst = [], [], []
    for ci, c in enumerate(centres):
        p = np.asarray(c, np.float64) + (rng.random((n_per, 3)) - 0.5) * spread
        xyz.append(p)
        s = np.full(n_per, 2 * ci, dtype=np.int64)
        s[-1] = 2 * ci + 1                      # a superpoint of ONE point
        sp.append(s)
        inst.append(np.full(n_per, ci, dtype=np.int64))
    xyz = np.concatenate(xyz).astype(np.float32)
    xyz -= xyz.min(0)
    sp, inst = np.concatenate(sp), np.concatenate(inst)
    S, n_inst = int(sp.max()) + 1, len(centres)
    und = [(2 * i, 2 * i + 1) for i in range(n_inst)] + [(2 * i, 2 * i + 2) for i in range(n_inst - 1)]
    edges = np.array(sorted(und + [(b, a) for a, b in und]), dtype=np.int64)
    cnt = np.bincount(sp, minlength=S).astype(np.float64)
    centre = np.stack([np.bincount(sp, xyz[:, j], S) / cnt for j in range(3)], 1)
    sp_inst = np.array([i // 2 for i in range(S)])
    sp_sem = np.full(S, -100, dtype=np.int64)
    sp_ins = np.full(S, -100, dtype=np.int64)
    sem_of = rng.integers(0, 20, n_inst)
    for i in range(n_inst):                       # weak supervision: one labelled superpoint per instance
        sp_sem[2 * i], sp_ins[2 * i] = sem_of[i], i
    inst_centre = np.stack([xyz[inst == i].mean(0) for i in range(n_inst)])
    inst_vox = np.array([max(int((inst == i).sum()), 1) for i in range(n_inst)])
    inst_size = np.array([np.linalg.norm(xyz[inst == i].max(0) - xyz[inst == i].min(0)) for i in range(n_inst)])
    return dict(xyz=xyz, rgb=rng.uniform(-1, 1, (len(xyz), 3)).astype(np.float32), superpoint=sp, edges=edges,
                edge_feats=rng.standard_normal((len(edges), 13)).astype(np.float32), sem_label=sp_sem[sp],
                ins_label=sp_ins[sp], sp_sem=sp_sem, sp_ins=sp_ins,
                sp_offset=(inst_centre[sp_inst] - centre).astype(np.float32),
                sp_voxnum=inst_vox[sp_inst].astype(np.float32), sp_size=inst_size[sp_inst].astype(np.float32),
                n_inst=n_inst, S=S)


def _models(cfg):
    model, crit, opt = harness.build_model(cfg, "cuda")
    ref = network_ref.RefNetwork()
    ref.load_state_dict({k: v.detach().cpu() for k, v in model.state_dict().items()}, strict=True)
    return model, crit, ref


def _compare_train(scenes, tol_fwd=2e-3, tol_grad=3e-2):
    cfg = harness.default_cfg()
    cfg.batch_size = len(scenes)
    host = harness.collate(scenes)
    model, crit, ref = _models(cfg)
    batch = harness.to_device(host, "cuda")
    model.train()
    ref.train()
    loss, ret = harness.forward_loss(model, crit, batch, cfg)
    loss.backward()
    r_loss, r_ret = network_ref.forward_loss_cpu(ref, crit, host)
    r_loss.backward()
    for k in KEYS:
        assert ret[k].shape == r_ret[k].shape, k
        assert _rel(ret[k], r_ret[k]) < tol_fwd, (k, _rel(ret[k], r_ret[k]))
    assert torch.isfinite(loss).item() and abs(float(loss.detach()) - float(r_loss.detach())) < 2e-3 * abs(float(r_loss.detach()))
    ref_params = dict(ref.named_parameters())
    gmax = max(float(p.grad.norm()) for p in ref.parameters() if p.grad is not None)
    worst = ("", 0.0)
    for name, p in model.named_parameters():
        rp = ref_params[name]
        if rp.grad is None:
            continue
        assert p.grad is not None and torch.isfinite(p.grad).all(), name
        e = float((p.grad.detach().cpu().double() - rp.grad.double()).norm()) / (float(rp.grad.norm()) + 1e-5 * gmax)
        worst = max(worst, (name, e), key=lambda t: t[1])
    assert worst[1] < tol_grad, worst
    return host


def test_scene_smaller_than_one_mfma_slice_at_every_level():
    # two clusters 0.8 m apart: ~25 voxels at level 0 (one partial 32-row slice), 2 voxels at level 4
    host = _compare_train([_scene(3, [(0.1, 0.1, 0.1), (0.9, 0.2, 0.1)])])
    assert 2 <= host["voxel_locs"].shape[0] < 32
    assert len(set(map(tuple, (host["voxel_locs"][:, 1:] // 16).tolist()))) >= 2        # >= 2 voxels at level 4


def test_batch_mixing_a_tiny_scene_with_a_normal_one():
    big = harness.make_scene(21, room=(1.3, 1.1, 0.9), n_box=1)
    tiny = _scene(4, [(0.1, 0.1, 0.1), (0.9, 0.2, 0.1), (0.2, 0.9, 0.5)], n_per=9)
    _compare_train([tiny, big])
    _compare_train([big, tiny])


def test_all_points_in_one_voxel_eval_forward():
    rng = np.random.default_rng(0)
    sc = _scene(5, [(0.1, 0.1, 0.1)], n_per=12, spread=0.004)
    sc["xyz"] = (np.array([0.107, 0.107, 0.107]) + (rng.random((12, 3)) - 0.5) * 0.004).astype(np.float32)
    cfg = harness.default_cfg()
    host = harness.collate([sc])
    assert host["voxel_locs"].shape[0] == 1
    model, crit, ref = _models(cfg)
    model.eval()
    ref.eval()
    batch = harness.to_device(host, "cuda")
    with torch.no_grad():
        loss, ret = harness.forward_loss(model, crit, batch, cfg)
        r_loss, r_ret = network_ref.forward_loss_cpu(ref, crit, host)
    for k in KEYS:
        assert ret[k].shape == r_ret[k].shape and torch.isfinite(ret[k]).all(), k
        assert _rel(ret[k], r_ret[k]) < 2e-3, (k, _rel(ret[k], r_ret[k]))


def test_s3dis_class_count_through_the_whole_path():
    """13 semantic classes (config/S3DIS_3D_WSIS.yaml): the point- and superpoint-level heads are 32 -> 13 / 64 -> 13
    Linear layers (widths the row-split weight-gradient kernel only takes zero-padded), the point-level loss kernels run
    with 13 classes"""
    cfg = harness.default_cfg()
    cfg.model.classes = 13
    scene = harness.make_scene(31, room=(1.3, 1.1, 0.9), n_box=1, classes=13)
    host = harness.collate([scene])
    model, crit, opt = harness.build_model(cfg, "cuda")
    ref = network_ref.RefNetwork(classes=13)
    ref.load_state_dict({k: v.detach().cpu() for k, v in model.state_dict().items()}, strict=True)
    batch = harness.to_device(host, "cuda")
    model.train()
    ref.train()
    loss, ret = harness.forward_loss(model, crit, batch, cfg)
    loss.backward()
    r_loss, r_ret = network_ref.forward_loss_cpu(ref, crit, host)
    r_loss.backward()
    assert ret["semantic_scores"].shape[1] == 13 and ret["sp_semantic_scores"].shape[1] == 13
    for k in KEYS:
        assert _rel(ret[k], r_ret[k]) < 2e-3, (k, _rel(ret[k], r_ret[k]))
    assert abs(float(loss.detach()) - float(r_loss.detach())) < 2e-3 * abs(float(r_loss.detach()))
    ref_params = dict(ref.named_parameters())
    gmax = max(float(p.grad.norm()) for p in ref.parameters() if p.grad is not None)
    for name in ("linear.3.weight", "linear.3.bias", "sp_sem_seg.3.weight", "sp_sem_seg.3.bias", "linear.0.weight"):
        p, rp = dict(model.named_parameters())[name], ref_params[name]
        e = float((p.grad.detach().cpu().double() - rp.grad.double()).norm()) / (float(rp.grad.norm()) + 1e-5 * gmax)
        assert e < 2e-2, (name, e)
