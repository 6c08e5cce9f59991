"""CPU: golden vectors produced by the IMPORTED reference (tests/golden/make_golden.py) pin
 (a) the MultiTaskLoss mirror (3d-wsis_amd/model/losses_3D_WSIS.py) and
 (b) the label-propagation oracle (oracle/affinity_ref.py), which in turn checks the HIP path on the GPU."""
import os
import types

import numpy as np
import pytest
import torch

import losses_3D_WSIS
from oracle import affinity_ref

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _loss_inputs(z, device="cpu"):
    t = lambda k: torch.from_numpy(z[k]).to(device)
    leaves = {k: t("in_" + k).clone().requires_grad_(True)
              for k in ("semantic_scores", "sp_semantic", "pred_off", "disc", "pred_occ", "pred_size")}
    loss_inp = {
        "point_labels": (t("in_sem_lab"), t("in_ins_lab")), "semantic_scores": leaves["semantic_scores"],
        "superpoint_labels": (t("in_sp_sem"), t("in_sp_ins")), "sp_semantic": leaves["sp_semantic"],
        "sp_offset_vector": (leaves["pred_off"], t("in_gt_off")),
        "sp_occupancy": (leaves["pred_occ"], t("in_gt_occ")),
        "sp_instance_size": (leaves["pred_size"], t("in_gt_size")),
        "sp_discriminative_features": (leaves["disc"], t("in_sp_off")),
    }
    return leaves, loss_inp


def _check_loss(device, slots=False):
    z = np.load(os.path.join(G, "loss_golden.npz"))
    pl = types.SimpleNamespace(ignore_label=-100, supervise_instance_size=True, joint_training_epoch=0,
                               semantic_dice=True, supervise_sp_offset=True)
    crit = losses_3D_WSIS.MultiTaskLoss(None, pl, types.SimpleNamespace(classes=20))
    for epoch, tag in ((0, "sem"), (5, "joint")):
        leaves, loss_inp = _loss_inputs(z, device)
        if slots:       # host-known bound of the instance ids per scene -> slot formulation of the push/pull terms
            ins, off = z["in_sp_ins"], z["in_sp_off"]
            loss_inp["sp_instance_slots"] = [max(int(ins[off[i]:off[i + 1]].max()) + 1, 1) + 3 * i
                                             for i in range(len(off) - 1)]      # a loose bound is as good
        loss, loss_out = crit(loss_inp, epoch)
        loss.backward()
        assert np.allclose(loss.item(), z[f"{tag}_loss"], rtol=1e-5, atol=1e-6)
        keys = [k[len(tag) + 1:] for k in z.files if k.startswith(tag + "_") and "grad" not in k and k != f"{tag}_loss"]
        assert set(keys) == set(loss_out.keys())
        for k in keys:
            assert np.allclose(loss_out[k][0].item(), z[f"{tag}_{k}"], rtol=1e-5, atol=1e-6), k
        for k, v in leaves.items():
            gk = f"{tag}_grad_{k}"
            if gk in z.files:
                assert np.allclose(v.grad.cpu().numpy(), z[gk], rtol=1e-4, atol=1e-7), gk
            else:
                assert v.grad is None or float(v.grad.abs().max()) == 0.0


def test_loss_matches_reference_golden_cpu():
    _check_loss("cpu")


def test_loss_slot_formulation_matches_reference_golden_cpu():
    _check_loss("cpu", slots=True)


@pytest.mark.gpu
def test_loss_matches_reference_golden_gpu():
    _check_loss("cuda")
    _check_loss("cuda", slots=True)


def _expected_labels(z, final):
    sem, ins = z["sem_label"].copy(), z["ins_label"].copy()
    for i, ind in enumerate(final):
        if ind != -100:
            sem[i], ins[i] = z["sem_label"][int(ind)], z["ins_label"][int(ind)]
    return sem, ins


def _adjacency(z):
    S = int(z["S"])
    adj = np.zeros((S, S), dtype=np.int64)
    np.add.at(adj, (z["edge_u"], z["edge_v"]), 1)
    return adj


@pytest.mark.parametrize("it", [0, 1, 2])
def test_propagation_oracle_matches_reference_golden(it):
    z = np.load(os.path.join(G, "propagation_golden.npz"))
    S = int(z["S"])
    A = affinity_ref.affinity_matrix(z["edge_u"], z["edge_v"], z["affinity"], S)
    final, scores, _ = affinity_ref.weak_label_propagation(A, _adjacency(z), z["conf"], z["pred"], z["sem_label"],
                                                           it, int(z["classes"]))
    sem, ins = _expected_labels(z, final)
    assert np.array_equal(sem, z[f"it{it}_semantic"])
    assert np.array_equal(ins, z[f"it{it}_instance"])
    assert int((ins != z["ins_label"]).sum()) > 10, "fixture must actually propagate labels"


@pytest.mark.gpu
@pytest.mark.parametrize("dense", [False, True])
@pytest.mark.parametrize("it", [0, 1, 2])
def test_propagation_hip_matches_reference_golden(it, dense):
    import wsis_ops
    z = np.load(os.path.join(G, "propagation_golden.npz"))
    S = int(z["S"])
    dev = "cuda"
    A = wsis_ops.affinity_matrix(torch.from_numpy(z["edge_u"]).to(dev), torch.from_numpy(z["edge_v"]).to(dev),
                                 torch.from_numpy(z["affinity"]).to(dev), S)
    final, scores = wsis_ops.weak_label_propagation(A, _adjacency(z), z["conf"], z["pred"], z["sem_label"], it,
                                                    int(z["classes"]), dense=dense)
    sem, ins = _expected_labels(z, final)
    assert np.array_equal(sem, z[f"it{it}_semantic"])
    assert np.array_equal(ins, z[f"it{it}_instance"])
