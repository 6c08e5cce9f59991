"""Generates the committed golden fixtures by IMPORTING the reference (read-only, this container only):

  loss_golden.npz         inputs + outputs (+ input gradients) of the reference
                          modules/model/losses_3D_WSIS.py MultiTaskLoss, stage-1 and stage-3 switches
  propagation_golden.npz  inputs + resulting labels of the reference
                          ScanNetV2Inst_spg.weak_label_propagation (modules/datasets/scannetv2_dataset.py:664-778)

Run:  python tests/golden/make_golden.py        (needs /root/reference; never runs on the GPU box)
Only data (inputs / expected outputs) is written -- no reference source text.
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


class _Logger:
    def info(self, *a, **k):
        pass


def make_loss():
    _stub("pointgroup_ops")
    sys.path.insert(0, os.path.join(REF, "modules", "model"))
    import losses_3D_WSIS as ref_loss
    assert ref_loss.__file__.startswith(REF)
    g = torch.Generator().manual_seed(1234)
    N, S, B, C = 4000, 300, 3, 20
    pl = types.SimpleNamespace(ignore_label=-100, supervise_instance_size=True, joint_training_epoch=0,
                               semantic_dice=True, supervise_sp_offset=True)
    pm = types.SimpleNamespace(classes=C)
    crit = ref_loss.MultiTaskLoss(_Logger(), pl, pm)
    crit.device = "cpu"
    sem_lab = torch.randint(0, C, (N,), generator=g)
    sem_lab[torch.rand(N, generator=g) < 0.7] = -100
    ins_lab = torch.randint(0, 30, (N,), generator=g)
    ins_lab[sem_lab == -100] = -100
    sp_off = [0, 90, 200, S]
    sp_sem = torch.randint(0, C, (S,), generator=g)
    sp_ins = torch.randint(0, 8, (S,), generator=g)
    for b in range(B):
        sp_ins[sp_off[b]:sp_off[b + 1]] += 10 * b
    unl = torch.rand(S, generator=g) < 0.5
    sp_sem[unl] = -100
    sp_ins[unl] = -100
    tensors = dict(
        semantic_scores=torch.randn(N, C, generator=g),
        sp_semantic=torch.randn(S, C, generator=g),
        pred_off=torch.randn(S, 3, generator=g), gt_off=torch.randn(S, 3, generator=g),
        disc=torch.randn(S, 7, generator=g),
        pred_occ=torch.randn(S, generator=g), gt_occ=torch.randn(S, generator=g).abs(),
        pred_size=torch.randn(S, generator=g), gt_size=torch.randn(S, generator=g).abs())
    leaves = {k: v.clone().requires_grad_(True) for k, v in tensors.items() if k in
              ("semantic_scores", "sp_semantic", "pred_off", "disc", "pred_occ", "pred_size")}
    loss_inp = {
        "point_labels": (sem_lab, ins_lab), "semantic_scores": leaves["semantic_scores"],
        "superpoint_labels": (sp_sem, sp_ins), "sp_semantic": leaves["sp_semantic"],
        "sp_offset_vector": (leaves["pred_off"], tensors["gt_off"]),
        "sp_occupancy": (leaves["pred_occ"], tensors["gt_occ"]),
        "sp_instance_size": (leaves["pred_size"], tensors["gt_size"]),
        "sp_discriminative_features": (leaves["disc"], torch.tensor(sp_off, dtype=torch.int32)),
    }
    out = {}
    for epoch, tag in ((0, "sem"), (5, "joint")):
        for v in leaves.values():
            v.grad = None
        loss, loss_out = crit(loss_inp, epoch)
        loss.backward()
        out[f"{tag}_loss"] = loss.detach().numpy()
        for k, (val, _) in loss_out.items():
            out[f"{tag}_{k}"] = val.detach().numpy()
        for k, v in leaves.items():
            if v.grad is not None:
                out[f"{tag}_grad_{k}"] = v.grad.numpy().copy()
    inputs = {f"in_{k}": v.numpy() for k, v in tensors.items()}
    inputs.update(in_sem_lab=sem_lab.numpy(), in_ins_lab=ins_lab.numpy(), in_sp_sem=sp_sem.numpy(),
                  in_sp_ins=sp_ins.numpy(), in_sp_off=np.array(sp_off, dtype=np.int32))
    np.savez_compressed(os.path.join(OUT, "loss_golden.npz"), **inputs, **out)
    print("loss_golden.npz:", {k: float(v) for k, v in out.items() if v.ndim == 0})


# ---- minimal stand-ins for the igraph objects the reference method touches (test doubles, not reference code)
class _VS:
    def __init__(self, attrs):
        self.attrs = attrs  # dict name -> list

    def __getitem__(self, key):
        if isinstance(key, str):
            return self.attrs[key]
        return _V(self, key)

    def __len__(self):
        return len(next(iter(self.attrs.values())))


class _V:
    def __init__(self, vs, i):
        self.vs, self.i = vs, i

    def __getitem__(self, name):
        return self.vs.attrs[name][self.i]

    def __setitem__(self, name, val):
        self.vs.attrs[name][self.i] = val


class _E:
    def __init__(self, g, j):
        self.g, self.j = g, j
        self.source, self.target = g.edges[j]

    def __setitem__(self, name, val):
        self.g.eattrs[name][self.j] = val


class _Adj:
    def __init__(self, data):
        self.data = data


class _Graph:
    def __init__(self, n, edges, vattrs):
        self.n, self.edges = n, edges
        self.vs = _VS(vattrs)
        self.eattrs = {"is1ins": [0] * len(edges)}

    def vcount(self):
        return self.n

    def get_adjacency(self):
        a = [[0] * self.n for _ in range(self.n)]
        for s, t in self.edges:
            a[s][t] += 1
        return _Adj(a)

    @property
    def es(self):
        return [_E(self, j) for j in range(len(self.edges))]


def make_propagation():
    for name in ("igraph", "plyfile", "utils", "pointgroup_ops"):
        _stub(name)
    sys.modules["plyfile"].PlyData = sys.modules["plyfile"].PlyElement = object
    sys.modules["utils"].derive_logger = lambda *a, **k: _Logger()
    pkg = _stub("modules")
    pkg.__path__ = [os.path.join(REF, "modules")]
    mm = _stub("modules.model")
    mm.__path__ = []
    _stub("modules.model.ecc")
    mm.ecc = sys.modules["modules.model.ecc"]
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_scannet_ds",
                                                  os.path.join(REF, "modules", "datasets", "scannetv2_dataset.py"))
    ds = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ds)
    rng = np.random.default_rng(77)
    S, classes, N = 160, 20, 4000
    # undirected neighbourhood graph stored as both directions (prepare_data_inst_ScanNetV2.py:213-231)
    und = set()
    for u in range(S):
        for v in rng.choice(S, 4, replace=False):
            if u != v:
                und.add((min(u, int(v)), max(u, int(v))))
    edges = sorted(list(und) + [(b, a) for a, b in und])
    eu = np.array([e[0] for e in edges])
    ev = np.array([e[1] for e in edges])
    aff = rng.random(len(edges)).astype(np.float32)
    pred = rng.integers(0, classes, S)
    conf = rng.random(S).astype(np.float32)
    conf[rng.random(S) < 0.6] = 0.95
    sem_label = np.full(S, -100)
    ins_label = np.full(S, -100)
    labelled = rng.choice(S, 24, replace=False)
    for j, sp in enumerate(labelled):
        sem_label[sp] = pred[sp] = int(rng.integers(0, 6))
        ins_label[sp] = 1000 + j                     # unique instance id per labelled superpoint
    # make neighbours of labelled superpoints agree with them often, so labels actually propagate
    for sp in labelled:
        for (a, b) in edges:
            if a == sp and rng.random() < 0.8 and sem_label[b] == -100:
                pred[b] = sem_label[sp]
    superpoint = rng.integers(0, S, N)
    superpoint[:S] = np.arange(S)
    xyz = rng.random((N, 3)).astype(np.float32)
    golden = dict(S=S, classes=classes, edge_u=eu, edge_v=ev, affinity=aff, pred=pred, conf=conf,
                  sem_label=sem_label, ins_label=ins_label)
    for it in (0, 1, 2):
        graph = _Graph(S, edges, {"v": list(range(S)), "semantic_label": sem_label.tolist(),
                                  "instance_label": ins_label.tolist(),
                                  "superpoint_offset_vector": [np.zeros(3) for _ in range(S)]})
        fake = types.SimpleNamespace(superpoints_graph={"scene": graph}, CLASS_NUM=classes, weak_label_spg={},
                                     scene2files={"scene": (xyz, None, None, None, superpoint, "scene")})
        A = np.zeros((S, S))
        for u, v, a in zip(eu, ev, aff):           # train_scannetv2.py:567-570
            A[u][v] = a
        ds.ScanNetV2Inst_spg.weak_label_propagation(fake, "scene", conf, pred, A, it)
        out = fake.weak_label_spg["scene"]
        golden[f"it{it}_semantic"] = np.array(out.vs["semantic_label"])
        golden[f"it{it}_instance"] = np.array(out.vs["instance_label"])
        golden[f"it{it}_is1ins"] = np.array(out.eattrs["is1ins"])
        print("propagation it", it, "newly labelled:", int((golden[f'it{it}_instance'] != ins_label).sum()))
    np.savez_compressed(os.path.join(OUT, "propagation_golden.npz"), **golden)


if __name__ == "__main__":
    make_loss()
    make_propagation()
