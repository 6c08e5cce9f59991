"""Generates tests/golden/dataset_golden.npz by running the REFERENCE's own per-scene preparation
(/root/reference/modules/datasets/scannetv2_dataset.py: ``__getitem__`` :96-190, ``data_aug`` :211-222, ``elastic``
:225-250, ``crop`` :252-273, ``get_instance_info`` :275-309, ``get_cropped_inst_label`` :311-330) in this container.

The dataset module cannot be imported (igraph, pointgroup_ops, ecc, utils are absent), so the methods' source is read
from the reference checkout AT GENERATION TIME, compiled and bound to a bare object carrying the attributes they read
(full_scale, scale, max_npoint, aug_flag, test_mode, task, files, ...); nothing of it is stored here.  The igraph
graph is replaced by an object with what ``__getitem__`` touches: ``vs`` (per-vertex dicts) and ``subgraph``.
``np.bool`` (removed in numpy 2) is aliased to ``bool`` for the run.

    python tests/golden/make_dataset_golden.py
"""
import ast
import copy
import importlib
import math
import os
import sys
import types

import numpy as np
import scipy
import scipy.interpolate
import scipy.ndimage
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
importlib.import_module("3d-wsis_amd")
import wsis_datasets as datasets                     # noqa: E402
import harness                      # noqa: E402

REF = "/root/reference/modules/datasets/scannetv2_dataset.py"
METHODS = ("__getitem__", "data_aug_with_graph", "data_aug", "elastic", "crop", "get_instance_info",
           "get_cropped_inst_label", "acquire_weak_label")


class GraphStub(object):
    """what ``__getitem__`` uses of igraph.Graph: iteration over ``vs`` with item access, and ``subgraph``."""

    def __init__(self, plain):
        self.plain = plain
        self.vs = [{k: plain.vs[k][i] for k in plain.vs} for i in range(plain.vcount)]

    def subgraph(self, subset):
        vs = {k: np.asarray([v[k] for v in self.vs]) for k in self.plain.vs}
        return datasets.PlainGraph(vs, self.plain.edges, self.plain.f, self.plain.is1ins).subgraph(subset)


REF_S3DIS = "/root/reference/modules/datasets/s3dis_dataset.py"


def reference_object(_path=REF, _cls="ScanNetV2Inst_spg", **attrs):
    tree = ast.parse(open(_path).read())
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == _cls][0]
    fns = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in METHODS + ("crop_v2",)]
    for f in fns:
        f.returns = None
        for a in f.args.args:
            a.annotation = None
    code = compile(ast.Module(body=fns, type_ignores=[]), _path, "exec")
    if not hasattr(np, "bool"):
        np.bool = bool
    if not hasattr(scipy.ndimage, "filters"):
        scipy.ndimage.filters = scipy.ndimage
    import collections

    class _Stats(object):          # scipy < 1.9 API the reference was written for: mode() returns arrays
        @staticmethod
        def mode(x):
            import scipy.stats
            r = scipy.stats.mode(x, keepdims=True)
            return r.mode, r.count

    ns = {"np": np, "math": math, "scipy": scipy, "torch": torch, "copy": copy, "collections": collections,
          "stats": _Stats}
    exec(code, ns)
    obj = types.SimpleNamespace(**attrs)
    for name in METHODS + ("crop_v2",):
        if name in ns:
            setattr(obj, name if name != "__getitem__" else "getitem", types.MethodType(ns[name], obj))
    return obj


def main():
    out = {}
    # --- single methods -------------------------------------------------------------------------------------------
    ref = reference_object(full_scale=[128, 512], scale=50, max_npoint=3000)
    rs = np.random.RandomState(3)
    xyz = rs.rand(500, 3) * np.array([4.0, 3.0, 2.0])
    np.random.seed(21)
    out["aug_in"] = xyz
    out["aug_out"] = ref.data_aug(xyz, True, True, True)
    np.random.seed(22)
    out["aug_rot_only"] = ref.data_aug(xyz, False, False, True)
    np.random.seed(23)
    big = rs.rand(6000, 3) * np.array([700.0, 650.0, 120.0])            # voxel units, larger than 512 -> crop loops
    out["crop_in"] = big
    c_xyz, c_valid = ref.crop(big)
    out["crop_xyz"], out["crop_valid"] = c_xyz, c_valid
    lab = rs.randint(0, 12, 6000).astype(np.float64)
    lab[rs.rand(6000) < 0.2] = -100
    out["inst_in"] = lab
    cropped = ref.get_cropped_inst_label(lab.copy(), c_valid)
    out["inst_cropped"] = cropped
    n_inst, info = ref.get_instance_info(big[c_valid], cropped.astype(np.int32))
    out["info_n"] = np.int64(n_inst)
    out["info"] = info["instance_info"]
    out["info_pointnum"] = np.asarray(info["instance_pointnum"])
    np.random.seed(24)
    el_in = rs.rand(400, 3) * np.array([200.0, 150.0, 100.0])
    out["elastic_in"] = el_in
    out["elastic_out"] = ref.elastic(el_in, 6 * 50 // 50, 40 * 50 / 50)

    # --- the whole __getitem__ (val task: GT labels; augmentation on; crop forced by a small max_npoint) ------------
    sc = harness.make_scene(5, room=(1.0, 0.9, 0.8), n_box=2)
    tup, plain = datasets.synthetic_scene_to_reference_format(sc)
    for tag, aug, test_mode, max_npoint, seed in (("t", True, False, 250000, 31), ("c", True, False, 11000, 32),
                                                  ("e", False, True, 250000, 33)):
        ref = reference_object(full_scale=[128, 512], scale=50, max_npoint=max_npoint, aug_flag=aug,
                               test_mode=test_mode, task="val", files=[tup],
                               superpoints_graph={"synthetic": None}, weak_label_spg={"synthetic": GraphStub(plain)},
                               superpoints={"synthetic": tup[4]})
        np.random.seed(seed)
        torch.manual_seed(seed)
        item = ref.getitem(0)
        scene, loc, loc_offset, loc_float, feat, sem, ins, sp, G, inst_num, inst_info, inst_pointnum = item
        out[tag + "_loc"] = loc.numpy(); out[tag + "_loc_offset"] = loc_offset.numpy()
        out[tag + "_loc_float"] = loc_float.numpy(); out[tag + "_feat"] = feat.numpy()
        out[tag + "_sem"] = sem.numpy(); out[tag + "_ins"] = ins.numpy(); out[tag + "_sp"] = sp.numpy()
        out[tag + "_inst_num"] = np.int64(inst_num); out[tag + "_inst_info"] = inst_info.numpy()
        out[tag + "_inst_pointnum"] = np.asarray(inst_pointnum)
        out[tag + "_g_off"] = G.vs["superpoint_offset_vector"]; out[tag + "_g_v"] = G.vs["v"]
        out[tag + "_g_edges"] = G.edges
        out[tag + "_cfg"] = np.asarray([int(aug), int(test_mode), max_npoint, seed])
        print(tag, "points", loc.shape[0], "of", len(tup[0]), "superpoints", len(G.vs["v"]), "edges", len(G.edges))
    # --- S3DIS variant (s3dis_dataset.py): train item with the random quarter of the points and the block crop ------
    sc = harness.make_scene(5, room=(1.0, 0.9, 0.8), n_box=2)
    tup, plain = datasets.synthetic_scene_to_reference_format(sc)
    for tag, sub, max_npoint, seed in (("s3a", True, 250000, 51), ("s3b", False, 6000, 52), ("s3c", True, 2000, 53)):
        ref = reference_object(REF_S3DIS, "S3DIS_Inst_spg", full_scale=[128, 512], scale=50, max_npoint=max_npoint,
                               aug_flag=True, test_mode=False, task="train", files=[tup], subsample_train=sub,
                               scene_point_level_weak_label={"synthetic": (tup[2], tup[3])},
                               weak_label_spg={"synthetic": GraphStub(plain)}, superpoints={"synthetic": tup[4]})
        np.random.seed(seed)
        torch.manual_seed(seed)
        item = ref.getitem(0)
        scene, loc, loc_offset, loc_float, feat, sem, ins, sp, G, inst_num, inst_info, inst_pointnum = item
        out[tag + "_loc"] = loc.numpy(); out[tag + "_loc_offset"] = loc_offset.numpy()
        out[tag + "_loc_float"] = loc_float.numpy()
        out[tag + "_sem"] = sem.numpy(); out[tag + "_ins"] = ins.numpy(); out[tag + "_sp"] = sp.numpy()
        out[tag + "_inst_num"] = np.int64(inst_num); out[tag + "_inst_info"] = inst_info.numpy()
        out[tag + "_g_off"] = G.vs["superpoint_offset_vector"]; out[tag + "_g_v"] = G.vs["v"]
        out[tag + "_g_edges"] = G.edges
        out[tag + "_cfg"] = np.asarray([int(sub), max_npoint, seed])
        print(tag, "points", loc.shape[0], "of", len(tup[0]), "superpoints", len(G.vs["v"]))

    # --- acquire_weak_label (:970-1036): GT-labelled synthetic scene, 1 and 2 annotated superpoints per instance ------
    sc = harness.make_scene(9, room=(1.0, 0.9, 0.8), n_box=3)
    rs = np.random.RandomState(9)
    sem_gt = rs.randint(0, 20, sc["S"])[sc["superpoint"]].astype(np.float64)
    ins_gt = (sc["superpoint"] % 7).astype(np.float64)            # 7 instances made of many superpoints each
    ins_gt[rs.rand(len(ins_gt)) < 0.05] = -100                    # some unlabelled points
    sem_gt[rs.rand(len(sem_gt)) < 0.1] = rs.randint(0, 20)        # label noise inside superpoints -> real modes
    for tag, k, seed in (("w1", 1, 41), ("w2", 2, 42)):
        _, plain = datasets.synthetic_scene_to_reference_format(sc)
        plain.vs["semantic_label"] = rs.randint(0, 20, sc["S"])
        plain.vs["instance_label"] = np.arange(sc["S"]) % 7
        stub = GraphStub(plain)
        ref = reference_object()
        np.random.seed(seed)
        ref.acquire_weak_label(sc["xyz"], sem_gt, ins_gt, sc["superpoint"], stub, k)
        out[tag + "_sem"] = np.asarray([v["semantic_label"] for v in stub.vs])
        out[tag + "_ins"] = np.asarray([v["instance_label"] for v in stub.vs])
        out[tag + "_off"] = np.asarray([np.asarray(v["superpoint_offset_vector"], dtype=np.float64) for v in stub.vs])
        out[tag + "_in_sem"] = plain.vs["semantic_label"]
        out[tag + "_in_ins"] = plain.vs["instance_label"]
        out[tag + "_cfg"] = np.asarray([k, seed])
        print(tag, "annotated superpoints", int((out[tag + "_ins"] != -100).sum()), "of", sc["S"])
    out["w_sem_gt"], out["w_ins_gt"] = sem_gt, ins_gt
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dataset_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
