"""tests/golden/network_golden.npz holds what the REFERENCE's own model file computes (``Network.forward`` of
/root/reference/modules/model/backbone_3D_WSIS.py with sparse_unet3d.py / graphnet.py / spg_modules.py, run by
tests/golden/make_network_golden.py with oracle-backed stand-ins for the absent spconv / torch_scatter / PyG) on a
seeded two-scene batch with name-seeded weights.  It pins the model glue -- wiring, residual / skip order, BatchNorm
placement, message-passing direction, GRUCellEx, heads, affinity block -- of

  * oracle/network_ref.py (here, CPU): the arithmetic is the same restatement, so the agreement is to fp32 rounding;
  * the HIP path (-m gpu): ``backbone_3D_WSIS.Network`` on the device, within the network tolerance."""
import os

import numpy as np
import pytest
import torch

from tests.util import seeded_state_dict

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "network_golden.npz")
KEYS = ("semantic_scores", "sp_semantic_scores", "pred_sp_offset_vectors", "pred_sp_occupancy", "pred_sp_ins_size",
        "edge_affinity", "sp_discriminative_feats")


def _rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def _check(ret, g, mode, tol):
    worst = {}
    for k in KEYS:
        got = ret[k].detach().cpu().numpy()
        if k == "semantic_scores":
            worst[k + "_colsum"] = _rel(got.astype(np.float64).sum(0), g[f"{mode}_{k}_colsum"])
            got = got[::4]
        want = g[f"{mode}_{k}"]
        assert got.shape == want.shape, (k, got.shape, want.shape)
        worst[k] = _rel(got, want)
    bad = {k: v for k, v in worst.items() if not v <= tol}
    assert not bad, (mode, bad)
    return worst


def test_oracle_network_matches_the_reference_model_file():
    from oracle import network_ref
    g = np.load(GOLD)
    ref = network_ref.RefNetwork()
    sd = ref.state_dict()
    assert sorted(sd.keys()) == list(g["state_names"]), "the oracle's state dict must carry the reference's names"
    ref.load_state_dict(seeded_state_dict({k: v.shape for k, v in sd.items()}), strict=True)
    t = lambda k: torch.from_numpy(g["in_" + k])
    for mode in ("train", "eval"):
        ref.train(mode == "train")
        with torch.no_grad():
            ret = ref(t("voxel_feats"), g["in_voxel_locs"], g["in_spatial_shape"], t("p2v_map"), t("superpoint"),
                      t("centre"), t("edge_indexes"), t("edgefeats"), t("edge_u"), t("edge_v"))
        _check(ret, g, mode, 2e-5)
    for k, v in ref.state_dict().items():          # the training pass updated the running statistics the same way
        if k.endswith("running_mean") or k.endswith("running_var"):
            assert np.allclose(v.numpy(), g["stat_" + k], rtol=1e-5, atol=1e-6), k


@pytest.mark.gpu
def test_hip_network_matches_the_reference_model_file():
    import spconv
    import harness
    import backbone_3D_WSIS
    from graphnet import GraphConvInfo
    g = np.load(GOLD)
    dev = "cuda"
    cfg = harness.default_cfg()
    net = backbone_3D_WSIS.Network(cfg.model)
    sd = net.state_dict()
    assert sorted(sd.keys()) == list(g["state_names"])
    net.load_state_dict(seeded_state_dict({k: v.shape for k, v in sd.items()}), strict=True)
    net = net.to(dev)
    t = lambda k: torch.from_numpy(g["in_" + k]).to(dev)
    S = int(g["in_superpoint"].max()) + 1
    for mode in ("train", "eval"):
        net.train(mode == "train")
        inp = spconv.SparseConvTensor(t("voxel_feats"), t("voxel_locs").int().contiguous(), g["in_spatial_shape"],
                                      int(g["in_batch_size"]))
        extra = {"superpoint": t("superpoint"), "GIs": [GraphConvInfo(t("edge_indexes"), t("edgefeats"), S)],
                 "superpoint_cenetr_xyz": t("centre"), "edge_u_list": t("edge_u"), "edge_v_list": t("edge_v")}
        with torch.no_grad():
            ret = net(inp, t("p2v_map"), extra)
        worst = _check(ret, g, mode, 2e-3)
        print(mode, {k: "%.1e" % v for k, v in worst.items()})
    for k, v in net.state_dict().items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            assert np.allclose(v.cpu().numpy(), g["stat_" + k], rtol=2e-3, atol=2e-4), k


@pytest.mark.gpu
def test_reference_layout_checkpoint_loads_on_the_gpu_and_reproduces_the_reference_forward(tmp_path):
    """utils/checkpoint.py:105-135 of the reference: the state dict sits under "model" (here with the "module." prefix a
    DataParallel wrapper leaves), is stripped and loaded strictly; the network then reproduces the outputs the
    reference's own model file computed for these weights (network_golden.npz).  save_checkpoint -> load_checkpoint
    round-trips the container (utils/checkpoint.py:205-262: "meta", "model", "optimizer")."""
    import spconv
    import harness
    from graphnet import GraphConvInfo
    g = np.load(GOLD)
    dev = "cuda"
    cfg = harness.default_cfg()
    model, crit, opt = harness.build_model(cfg, dev)
    sd = seeded_state_dict({k: v.shape for k, v in model.state_dict().items()})
    f1 = str(tmp_path / "epoch_00042.pth")
    torch.save({"meta": {"epoch": 42, "iter": 5040}, "model": {"module." + k: v for k, v in sd.items()}}, f1)
    ck = harness.load_checkpoint(model, f1, map_location="cpu", strict=True)
    assert ck["meta"]["epoch"] == 42
    assert all(p.is_cuda for p in model.parameters())
    t = lambda k: torch.from_numpy(g["in_" + k]).to(dev)
    S = int(g["in_superpoint"].max()) + 1
    for mode in ("train", "eval"):          # the fixture's order: the training pass moves the running statistics first
        model.train(mode == "train")
        inp = spconv.SparseConvTensor(t("voxel_feats"), t("voxel_locs").int().contiguous(), g["in_spatial_shape"],
                                      int(g["in_batch_size"]))
        extra = {"superpoint": t("superpoint"), "GIs": [GraphConvInfo(t("edge_indexes"), t("edgefeats"), S)],
                 "superpoint_cenetr_xyz": t("centre"), "edge_u_list": t("edge_u"), "edge_v_list": t("edge_v")}
        with torch.no_grad():
            ret = model(inp, t("p2v_map"), extra)
        _check(ret, g, mode, 2e-3)
    # container round trip, optimizer state included
    f2 = str(tmp_path / "out" / "latest.pth")
    harness.save_checkpoint(model, f2, optimizer=opt, meta={"epoch": 43})
    model2, _, opt2 = harness.build_model(cfg, dev, seed=7)
    ck2 = harness.load_checkpoint(model2, f2, optimizer=opt2)
    assert ck2["meta"]["epoch"] == 43 and set(ck2) >= {"meta", "model", "optimizer"}
    for (k, a), (_, b) in zip(model.state_dict().items(), model2.state_dict().items()):
        assert torch.equal(a, b), k
