"""Host side of the input contract: the reference's per-scene preparation and batch assembly for the hot path.

Mirrors ``modules/datasets/scannetv2_dataset.py`` -- ``__getitem__`` (:96-190), ``data_aug_with_graph`` (:194-209),
``elastic`` (:225-250), ``crop`` (:252-273), ``get_instance_info`` (:275-309), ``get_cropped_inst_label`` (:311-330)
and ``collate_fn`` (:343-474, schema in SURVEY App. C) -- on plain numpy arrays.  Pure host code, safe in DataLoader
workers (``pointgroup_ops.voxelization_idx`` runs in ``libwsis_host.so``).

Two deliberate differences:

* randomness comes from an explicit ``numpy.random.RandomState`` (the reference uses numpy's global state; with
  ``RandomState(seed)`` the draws equal the reference's after ``np.random.seed(seed)``), and the colour jitter from
  a ``torch.Generator``;
* the superpoint graph is a :class:`PlainGraph` (arrays) instead of an ``igraph.Graph``: igraph is not in the
  image, so the ``*_spg.dat`` pickles cannot be opened here.  ``PlainGraph.from_igraph`` is the converter a
  maintainer runs where igraph exists.
"""
import math

import numpy as np
import torch

VERTEX_ATTRS = ("v", "semantic_label", "instance_label", "superpoint_offset_vector", "instance_voxel_num",
                "instance_size")


class PlainGraph(object):
    """Superpoint graph as arrays: per-vertex attributes (``prepare_data_inst_ScanNetV2.py:268-271``) and a directed
    edge list with 13 edge features ``f`` and the ``is1ins`` flag."""

    def __init__(self, vs, edges, f=None, is1ins=None):
        self.vs = {k: np.asarray(a) for k, a in vs.items()}
        self.edges = np.asarray(edges, dtype=np.int64).reshape(-1, 2)
        n_e = self.edges.shape[0]
        self.f = np.zeros((n_e, 13), np.float32) if f is None else np.asarray(f, dtype=np.float32).reshape(n_e, -1)
        self.is1ins = np.zeros(n_e, np.int64) if is1ins is None else np.asarray(is1ins, dtype=np.int64)

    @property
    def vcount(self):
        return len(next(iter(self.vs.values())))

    def copy(self):
        return PlainGraph({k: a.copy() for k, a in self.vs.items()}, self.edges.copy(), self.f.copy(),
                          self.is1ins.copy())

    def subgraph(self, subset):
        """Induced subgraph on the ascending vertex list ``subset`` (``superpoint_graph.subgraph(subset)``, :169):
        kept vertices are renumbered 0..len-1 in that order, edges with both ends kept stay in their order."""
        subset = np.asarray(subset, dtype=np.int64)
        new_id = np.full(self.vcount, -1, np.int64)
        new_id[subset] = np.arange(len(subset))
        e = new_id[self.edges] if len(self.edges) else self.edges
        keep = (e >= 0).all(1) if len(e) else np.zeros(0, bool)
        return PlainGraph({k: a[subset] for k, a in self.vs.items()}, e[keep], self.f[keep], self.is1ins[keep])

    @staticmethod
    def from_igraph(g):
        """Converter for a machine that has igraph (not this image): ``igraph.Graph.Read_Pickle(..._spg.dat)``."""
        vs = {k: np.asarray(g.vs[k]) for k in g.vs.attributes()}
        edges = np.asarray([e.tuple for e in g.es], dtype=np.int64).reshape(-1, 2)
        f = np.asarray(g.es["f"], dtype=np.float32) if "f" in g.es.attributes() else None
        one = np.asarray(g.es["is1ins"]) if "is1ins" in g.es.attributes() else None
        return PlainGraph(vs, edges, f, one)

    def save(self, path):
        np.savez_compressed(path, edges=self.edges, f=self.f, is1ins=self.is1ins,
                            **{"vs_" + k: a for k, a in self.vs.items()})

    @staticmethod
    def load(path):
        z = np.load(path)
        return PlainGraph({k[3:]: z[k] for k in z.files if k.startswith("vs_")}, z["edges"], z["f"], z["is1ins"])


class _PickledIGraph(object):
    """What ``igraph.Graph.__reduce__`` stores: the constructor arguments ``(n, edges, directed, graph_attrs,
    vertex_attrs, edge_attrs)`` (python-igraph 0.8 - 0.10: ``Graph.__reduce__`` returns ``(cls, (vcount, edgelist,
    is_directed, gattrs, vattrs, eattrs), __dict__)``; ``write_pickle`` is ``pickle.dump(graph)``).  Stands in for the
    class while a ``_spg.dat`` file is unpickled on a machine without igraph."""

    def __init__(self, n=0, edges=None, directed=False, graph_attrs=None, vertex_attrs=None, edge_attrs=None, *rest):
        self.n, self.edges, self.directed = int(n), list(edges or []), bool(directed)
        self.graph_attrs, self.vertex_attrs, self.edge_attrs = dict(graph_attrs or {}), dict(vertex_attrs or {}), dict(edge_attrs or {})

    def __setstate__(self, state):       # the instance __dict__ igraph appends (empty for a plain Graph)
        pass


class _SpgUnpickler(__import__("pickle").Unpickler):
    """resolves ``igraph.Graph`` (whatever sub-module the installed version defined it in) to the stand-in and lets exactly
    the globals a pickled graph with numpy attributes needs through -- an allow-list of (module, name) PAIRS, not of
    modules: ``builtins.eval`` / ``numpy.testing...runstring`` reachable through REDUCE would be code execution.
    Anything else in the stream raises ``UnpicklingError``."""

    _OK = frozenset(
        [(m, "_reconstruct") for m in ("numpy.core.multiarray", "numpy._core.multiarray")] +
        [(m, "scalar") for m in ("numpy.core.multiarray", "numpy._core.multiarray")] +
        [(m, "_frombuffer") for m in ("numpy.core.numeric", "numpy._core.numeric")] +      # (pickle protocol 5 arrays)
        [("numpy", "ndarray"), ("numpy", "dtype"), ("_codecs", "encode"), ("collections", "OrderedDict")] +
        [("builtins", n) for n in ("list", "dict", "tuple", "set", "frozenset", "int", "float", "complex", "str", "bytes",
                                   "bytearray", "bool", "slice", "range")])

    def find_class(self, module, name):
        if module.split(".")[0] == "igraph" and name == "Graph":
            return _PickledIGraph
        if (module, name) in self._OK:
            return super().find_class(module, name)
        raise __import__("pickle").UnpicklingError(f"_spg.dat: unexpected global {module}.{name}")


def read_spg_pickle(path):
    """``igraph.Graph.Read_Pickle(scene + '_spg.dat')`` (``scannetv2_dataset.py:79``) WITHOUT igraph -> PlainGraph.
    The file is the pickle ``graph.write_pickle`` wrote (``prepare_data_inst_ScanNetV2.py:88,163``; gzip-compressed
    streams are accepted like Read_Pickle does): vertex attributes ``v, semantic_label, instance_label,
    superpoint_feature, superpoint_offset_vector``, directed edge list in the prep's sorted-tuple order with edge
    attributes ``f`` (13 standardised features) and ``is1ins`` (``:268-280``)."""
    import gzip
    with open(path, "rb") as fh:
        head = fh.read(2)
    opener = gzip.open if head == b"\x1f\x8b" else open
    with opener(path, "rb") as fh:
        g = _SpgUnpickler(fh).load()
    if not isinstance(g, _PickledIGraph):
        raise ValueError(f"{path}: not an igraph.Graph pickle")
    vs = {k: np.asarray(v) for k, v in g.vertex_attrs.items()}
    for k, a in vs.items():
        if len(a) != g.n:
            raise ValueError(f"{path}: vertex attribute {k} has {len(a)} entries for {g.n} vertices")
    if not vs:
        vs = {"v": np.arange(g.n)}
    edges = np.asarray(g.edges, dtype=np.int64).reshape(-1, 2)
    f = np.asarray(g.edge_attrs["f"], dtype=np.float32).reshape(len(edges), -1) if "f" in g.edge_attrs else None
    one = np.asarray(g.edge_attrs["is1ins"]).astype(np.int64) if "is1ins" in g.edge_attrs else None
    return PlainGraph(vs, edges, f, one)


def _numpy_safe_globals():
    """The globals a ``torch.save`` of plain numpy arrays references, as (callable, path-in-the-file) pairs for
    ``torch.serialization.safe_globals``: the array reconstructor under both of its module names (files written with
    numpy 1.x say ``numpy.core.multiarray``, numpy 2 says ``numpy._core.multiarray``), ``ndarray``, ``dtype`` and the
    per-type dtype classes newer numpy pickles.  Nothing here can run code: ``_reconstruct`` / ``scalar`` build an
    array / a scalar from bytes (object dtypes are refused below)."""
    try:
        from numpy._core.multiarray import _reconstruct, scalar
    except ImportError:                                            # numpy 1.x
        from numpy.core.multiarray import _reconstruct, scalar
    out = [np.ndarray, np.dtype]
    for mod in ("numpy.core.multiarray", "numpy._core.multiarray"):
        out += [(_reconstruct, mod + "._reconstruct"), (scalar, mod + ".scalar")]
    for code in ("f2", "f4", "f8", "i1", "i2", "i4", "i8", "u1", "u2", "u4", "u8", "b1"):
        cls = type(np.dtype(code))
        if cls is not np.dtype:
            out.append(cls)
    return out


def safe_torch_load(path, map_location="cpu"):
    """``torch.load(..., weights_only=True)`` with the numpy reconstructors allow-listed: tensors, numpy arrays of
    numeric dtype, containers, strings and numbers load; every other global in the stream (a ``REDUCE`` of
    ``os.system``, a pickled class) raises ``pickle.UnpicklingError`` before anything of the file runs.  The loader of
    the reference's own files: per-scene ``.pth`` (``scannetv2_dataset.py:62-73``) and checkpoints
    (``utils/checkpoint.py:105-135``) are files of somebody else's making."""
    with torch.serialization.safe_globals(_numpy_safe_globals()):
        return torch.load(path, map_location=map_location, weights_only=True)


def load_scene_file(path):
    """The reference's per-scene ``.pth``: ``(coords, colors, sem, inst, superpoint, scene_name)``
    (``prepare_data_inst_ScanNetV2.py:166``, read at ``scannetv2_dataset.py:62,72``).  Read with ``safe_torch_load``:
    nothing of the file is executed."""
    t = safe_torch_load(path)
    if not (isinstance(t, (tuple, list)) and len(t) == 6):
        raise ValueError(f"{path}: expected the 6-tuple (coords, colors, sem, inst, superpoint, scene)")
    coords, colors, sem, inst, superpoint, scene = t
    arrays = [np.asarray(a) for a in (coords, colors, sem, inst, superpoint)]
    for a in arrays:
        if a.dtype.hasobject:
            raise ValueError(f"{path}: object arrays are not scene data")
    return (*arrays, str(scene))


class ScenePrep(object):
    """Per-scene transform of ``ScanNetV2Inst_spg.__getitem__``.

    ``full_scale`` [128, 512], ``scale`` 50, ``max_npoint`` 250000 are the values of
    ``config/ScanNet_v2_3D_WSIS.yaml`` (read at ``scannetv2_dataset.py:36-38``)."""

    def __init__(self, full_scale=(128, 512), scale=50, max_npoint=250000, aug=True, test_mode=False, seed=None,
                 crop_version=1, subsample_train=False):
        """``crop_version=2`` / ``subsample_train=True``: the S3DIS variant (``s3dis_dataset.py``: block crop around a
        random point, :285-319, and a random quarter of the points per training item, :135-144)."""
        self.crop_version = int(crop_version)
        self.subsample_train = bool(subsample_train)
        self.full_scale = [int(full_scale[0]), int(full_scale[1])]
        self.scale = scale
        self.max_npoint = max_npoint
        self.aug_flag = aug
        self.test_mode = test_mode
        self.rng = np.random.RandomState(seed)
        self.gen = torch.Generator()
        if seed is not None:
            self.gen.manual_seed(int(seed))

    # -- :211-222 / :194-209 -------------------------------------------------------------------------------------
    def aug_matrix(self, jitter=False, flip=False, rot=False):
        m = np.eye(3)
        if jitter:
            m += self.rng.randn(3, 3) * 0.1
        if flip:
            m[0][0] *= self.rng.randint(0, 2) * 2 - 1
        if rot:
            theta = self.rng.rand() * 2 * math.pi
            m = np.matmul(m, [[math.cos(theta), math.sin(theta), 0], [-math.sin(theta), math.cos(theta), 0],
                              [0, 0, 1]])
        return m

    def data_aug(self, xyz, jitter=False, flip=False, rot=False):
        return np.matmul(xyz, self.aug_matrix(jitter, flip, rot))

    def data_aug_with_graph(self, xyz, graph, jitter=False, flip=False, rot=False):
        """The same matrix also rotates every superpoint's offset vector (the reference loops over ``graph.vs``)."""
        m = self.aug_matrix(jitter, flip, rot)
        graph.vs["superpoint_offset_vector"] = np.matmul(graph.vs["superpoint_offset_vector"], m)
        return np.matmul(xyz, m)

    # -- :225-250 ------------------------------------------------------------------------------------------------
    def elastic(self, xyz, gran, mag):
        import scipy.interpolate
        import scipy.ndimage
        bb = np.abs(xyz).max(0).astype(np.int32) // gran + 3
        noise = [self.rng.randn(bb[0], bb[1], bb[2]).astype("float32") for _ in range(3)]
        for _ in range(2):
            for axis in range(3):
                shape = [1, 1, 1]
                shape[axis] = 3
                blur = np.ones(shape, "float32") / 3
                noise = [scipy.ndimage.convolve(n, blur, mode="constant", cval=0) for n in noise]
        ax = [np.linspace(-(b - 1) * gran, (b - 1) * gran, b) for b in bb]
        interp = [scipy.interpolate.RegularGridInterpolator(ax, n, bounds_error=0, fill_value=0) for n in noise]
        return xyz + np.hstack([i(xyz)[:, None] for i in interp]) * mag

    # -- :252-273 ------------------------------------------------------------------------------------------------
    def crop(self, xyz):
        xyz_offset = xyz.copy()
        valid = xyz_offset.min(1) >= 0
        if valid.sum() != xyz.shape[0]:
            raise ValueError("crop expects coordinates already shifted to be non-negative (:151-152)")
        full_scale = np.array([self.full_scale[1]] * 3)
        room_range = xyz.max(0) - xyz.min(0)
        while valid.sum() > self.max_npoint:
            offset = np.clip(full_scale - room_range + 0.001, None, 0) * self.rng.rand(3)
            xyz_offset = xyz + offset
            valid = (xyz_offset.min(1) >= 0) * ((xyz_offset < full_scale).sum(1) == 3)
            full_scale[:2] -= 32
        return xyz_offset, valid

    # -- s3dis_dataset.py:285-319 ------------------------------------------------------------------------------------
    def crop_v2(self, xyz):
        """S3DIS rooms: an x/y block around a random point, its half-widths the largest of 20 scale steps (binary
        search) that keeps at most ``max_npoint`` points; coordinates are shifted to the block's minimum."""
        out = xyz.copy()
        if (out.min(1) >= 0).sum() != xyz.shape[0]:
            raise ValueError("crop_v2 expects coordinates already shifted to be non-negative")
        room_max = xyz.max(0)
        center = xyz[self.rng.choice(len(xyz))][:3]
        half_x, half_y = max(room_max[0] - center[0], center[0]), max(room_max[1] - center[1], center[1])
        scale = np.arange(0, 1, 0.05)

        def inside(s):
            dx, dy = half_x * s, half_y * s
            lo, hi = center - [dx, dy, 0], center + [dx, dy, 0]
            return (xyz[:, 0] >= lo[0]) & (xyz[:, 0] <= hi[0]) & (xyz[:, 1] >= lo[1]) & (xyz[:, 1] <= hi[1])

        low, high = 0, len(scale) - 1
        while low < high:
            mid = int(math.ceil((low + high) / 2))
            if inside(scale[mid]).sum() <= self.max_npoint:
                low = mid
            else:
                high = mid - 1
        keep = inside(scale[high])
        out -= xyz[keep].min(0)
        return out, keep

    # -- :311-330 ------------------------------------------------------------------------------------------------
    @staticmethod
    def get_cropped_inst_label(instance_label, valid_idxs):
        """Re-compacts the ids after a crop exactly as the reference does: walking j upward, an empty id j takes over
        the points of the current largest id."""
        instance_label = instance_label[valid_idxs]
        if instance_label.size == 0:
            return instance_label
        j = 0
        while j < instance_label.max():
            if not (instance_label == j).any():
                instance_label[instance_label == instance_label.max()] = j
            j += 1
        return instance_label

    # -- :275-309 ------------------------------------------------------------------------------------------------
    @staticmethod
    def get_instance_info(xyz, instance_label):
        info = np.ones((xyz.shape[0], 9), dtype=np.float32) * -100.0
        pointnum = []
        n_inst = int(instance_label.max()) + 1 if instance_label.size else 0
        for i in range(n_inst):
            idx = np.where(instance_label == i)[0]
            pointnum.append(idx.size)
            if idx.size == 0:      # the reference would take min() of an empty array here; ids are compact, see above
                continue
            p = xyz[idx]
            info[idx, 0:3] = p.mean(0)
            info[idx, 3:6] = p.min(0)
            info[idx, 6:9] = p.max(0)
        return n_inst, {"instance_info": info, "instance_pointnum": pointnum}

    # -- :96-190 -------------------------------------------------------------------------------------------------
    def __call__(self, scene_tuple, graph):
        """(coords, colors, sem, inst, superpoint, scene), PlainGraph -> the 12-tuple ``__getitem__`` returns."""
        xyz_origin, rgb, semantic_label, instance_label, superpoint, scene = scene_tuple
        if self.subsample_train:                         # s3dis_dataset.py:135-144, before any other draw
            pick = self.rng.choice(len(xyz_origin), size=len(xyz_origin) // 4, replace=False)
            xyz_origin, rgb = np.asarray(xyz_origin)[pick], np.asarray(rgb)[pick]
            semantic_label, instance_label = np.asarray(semantic_label)[pick], np.asarray(instance_label)[pick]
            superpoint = np.asarray(superpoint)[pick]
        graph = graph.copy()
        flag = bool(self.aug_flag)
        xyz_middle = self.data_aug_with_graph(np.asarray(xyz_origin), graph, flag, flag, flag)
        xyz = xyz_middle * self.scale
        xyz_offset = xyz.min(0)
        xyz = xyz - xyz_offset
        valid = np.ones(len(xyz_middle), dtype=bool)
        if not self.test_mode:
            xyz, valid = self.crop(xyz) if self.crop_version == 1 else self.crop_v2(xyz)
        xyz_middle = xyz_middle[valid]
        xyz = xyz[valid]
        rgb = np.asarray(rgb)[valid]
        semantic_label = np.asarray(semantic_label)[valid]
        instance_label = self.get_cropped_inst_label(np.asarray(instance_label).copy(), valid)
        superpoint = np.asarray(superpoint)[valid]
        subset, new_superpoint = np.unique(superpoint, return_inverse=True)
        sub = graph.subgraph(subset)
        inst_num, infos = self.get_instance_info(xyz_middle, instance_label.astype(np.int32))
        feat = torch.from_numpy(np.ascontiguousarray(rgb))
        if self.aug_flag:
            feat = feat + torch.randn(3, generator=self.gen) * 0.1
        return (scene, torch.from_numpy(xyz).long(), torch.from_numpy(xyz_offset).long(),
                torch.from_numpy(xyz_middle), feat, torch.from_numpy(semantic_label),
                torch.from_numpy(instance_label), torch.from_numpy(new_superpoint.reshape(-1)), sub, inst_num,
                torch.from_numpy(infos["instance_info"]), infos["instance_pointnum"])


def _level_counts(voxel_locs, spatial_shape, n_levels=5):
    """host-side voxel counts of the UNet's strided levels (config ``blocks`` = 5); the device rulebook build sizes
    its tables from them instead of reading the counts back (spconv.ops.level_voxel_counts)"""
    import spconv
    return spconv.ops.level_voxel_counts(voxel_locs.numpy(), spatial_shape, n_levels)


def collate_fn(batch, full_scale_min=128, mode=4):
    """``collate_fn`` (:343-474): list of ``ScenePrep`` 12-tuples -> batch dict."""
    import pointgroup_ops
    from graphnet import GraphConvInfo
    locs, loc_offsets, locs_float, feats, sems, inss, sps = [], [], [], [], [], [], []
    infos, pointnum, scene_list = [], [], []
    sp_sem, sp_ins, sp_off, sp_vox, sp_size = [], [], [], [], []
    edge_sorted, feat_sorted, edges_ext, is1ins = [], [], [], []
    batch_offsets, sp_batch_offsets = [0], [0]
    sp_bias, total_inst = 0, 0
    for i, data in enumerate(batch):
        scene, loc, loc_offset, loc_float, feat, sem, ins, superpoint, graph, inst_num, inst_info, inst_pointnum = data
        scene_list.append(scene)
        superpoint = superpoint + sp_bias
        this_bias = sp_bias
        sp_bias = int(superpoint.max()) + 1
        sp_batch_offsets.append(sp_bias)
        ins = ins.clone()
        ins[ins != -100] += total_inst
        total_inst += inst_num
        batch_offsets.append(batch_offsets[-1] + loc.shape[0])
        locs.append(torch.cat([torch.full((loc.shape[0], 1), i, dtype=torch.int64), loc], 1))
        loc_offsets.append(loc_offset)
        locs_float.append(loc_float)
        feats.append(feat)
        sems.append(sem)
        inss.append(ins)
        sps.append(superpoint)
        sp_sem.append(torch.as_tensor(graph.vs["semantic_label"]))
        sp_ins.append(torch.as_tensor(graph.vs["instance_label"]))
        sp_off.append(torch.as_tensor(graph.vs["superpoint_offset_vector"]))
        sp_vox.append(torch.as_tensor(graph.vs["instance_voxel_num"]))
        sp_size.append(torch.as_tensor(graph.vs["instance_size"]))
        infos.append(inst_info)
        pointnum.extend(inst_pointnum)
        E = graph.edges
        order = np.argsort(E[:, 1], kind="stable")                 # ecc/GraphConvInfo.py:54-70 (sorted by target)
        edge_sorted.append(torch.from_numpy(E[order] + this_bias))
        feat_sorted.append(torch.from_numpy(graph.f[order]))
        edges_ext.append(torch.from_numpy(E + this_bias))          # original order (:455-457)
        is1ins.append(torch.from_numpy(graph.is1ins))
    locs = torch.cat(locs, 0)
    superpoint = torch.cat(sps, 0).long()
    if len(np.unique(superpoint.numpy())) != int(superpoint.max()) + 1:
        raise ValueError("superpoint ids are not dense after batching (:422)")
    spatial_shape = np.clip((locs.max(0)[0][1:] + 1).numpy(), full_scale_min, None)
    voxel_locs, p2v_map, v2p_map = pointgroup_ops.voxelization_idx(locs, len(batch), mode)
    GIs = [GraphConvInfo(torch.cat(edge_sorted, 0).t().contiguous(), torch.cat(feat_sorted, 0).float(), sp_bias)]
    edges = torch.cat(edges_ext, 0)
    return {
        "locs": locs, "locs_offset": torch.stack(loc_offsets), "voxel_locs": voxel_locs, "p2v_map": p2v_map,
        "v2p_map": v2p_map, "locs_float": torch.cat(locs_float, 0).to(torch.float32),
        "feats": torch.cat(feats, 0).to(torch.float32), "semantic_labels": torch.cat(sems, 0).long(),
        "instance_labels": torch.cat(inss, 0).long(), "instance_info": torch.cat(infos, 0).to(torch.float32),
        "instance_pointnum": torch.tensor(pointnum, dtype=torch.int),
        "offsets": torch.tensor(batch_offsets, dtype=torch.int), "spatial_shape": spatial_shape,
        "superpoint": superpoint, "GIs": GIs, "sp_batch_offsets": torch.tensor(sp_batch_offsets, dtype=torch.int),
        "edge_u_list": edges[:, 0].contiguous().long(), "edge_v_list": edges[:, 1].contiguous().long(),
        "edge_src_rows": (int(edges[:, 0].max()) + 1) if edges.shape[0] else 0,   # rows of scatter(.., edge_u)
        "level_counts": _level_counts(voxel_locs, spatial_shape),      # active voxels of the strided UNet levels
        "is1ins_labels": torch.cat(is1ins, 0),
        "superpoint_semantic_labels": torch.cat(sp_sem, 0).long(),
        "superpoint_instance_labels": torch.cat(sp_ins, 0).long(),
        "superpoint_offset_vector": torch.cat(sp_off, 0).to(torch.float32),
        "superpoint_instance_voxel_num": torch.log(torch.cat(sp_vox, 0).to(torch.float32)),
        "superpoint_instance_size": torch.cat(sp_size, 0).to(torch.float32),
        "scene_list": scene_list,
    }


def synthetic_scene_to_reference_format(sc):
    """A ``harness.make_scene`` scene as the reference's on-disk pair: the 6-tuple and the superpoint graph."""
    tup = (sc["xyz"].astype(np.float32), sc["rgb"].astype(np.float32), sc["sem_label"].astype(np.float64),
           sc["ins_label"].astype(np.float64), sc["superpoint"].astype(np.int64), "synthetic")
    g = PlainGraph({"v": np.arange(sc["S"]), "semantic_label": sc["sp_sem"], "instance_label": sc["sp_ins"],
                    "superpoint_offset_vector": sc["sp_offset"].astype(np.float64),
                    "instance_voxel_num": sc["sp_voxnum"], "instance_size": sc["sp_size"]},
                   sc["edges"], sc["edge_feats"])
    return tup, g


def _segment_mode(seg, values, n_seg):
    """per-segment mode of ``values`` (ties -> the smallest value, as ``scipy.stats.mode``)"""
    vals, inv = np.unique(values, return_inverse=True)
    pair = seg.astype(np.int64) * len(vals) + inv.reshape(-1)
    up, cnt = np.unique(pair, return_counts=True)
    s, v = up // len(vals), up % len(vals)
    order = np.lexsort((v, -cnt, s))                 # segment, then count descending, then value ascending
    first = np.ones(len(order), dtype=bool)
    first[1:] = s[order][1:] != s[order][:-1]
    out = np.full(n_seg, -100, dtype=vals.dtype)
    out[s[order][first]] = vals[v[order][first]]
    return out


def acquire_weak_label(xyz, semantic_labels, instance_labels, superpoint, graph, annotation_num=1, rng=None):
    """``ScanNetV2Inst_spg.acquire_weak_label`` (scannetv2_dataset.py:970-1036): per instance, ``annotation_num``
    superpoints are drawn with probability proportional to their point count (all of them if the instance has no
    more); the drawn ones keep their labels and get the offset to the centre of the drawn points of their instance,
    every other superpoint of ``graph`` (a PlainGraph, modified in place) is reset to label -100 / offset 0.
    ``rng``: ``numpy.random.RandomState`` (the reference draws from numpy's global state: ``RandomState(seed)``
    reproduces ``np.random.seed(seed)``).  Returns the list of annotated superpoint ids in drawing order."""
    rng = rng if rng is not None else np.random.RandomState()
    superpoint = np.asarray(superpoint).astype("int")
    sp_ids, sp_size = np.unique(superpoint, return_counts=True)
    n_sp = int(sp_ids.max()) + 1 if len(sp_ids) else 0
    sp_instance = _segment_mode(superpoint, np.asarray(instance_labels), n_sp)
    members = {}                                        # instance label -> [(superpoint, size)] in ascending id order
    for sp, size in zip(sp_ids, sp_size):
        members.setdefault(sp_instance[sp], []).append((sp, size))
    chosen_all = []
    off = np.array(graph.vs["superpoint_offset_vector"], dtype=np.float64, copy=True)
    for ins in np.unique(instance_labels):
        if ins not in members:
            continue
        ids = np.array([m[0] for m in members[ins]])
        num = np.array([m[1] for m in members[ins]])
        prob = num / num.sum()
        chosen = rng.choice(ids, size=annotation_num, p=prob, replace=False) if annotation_num < ids.shape[0] else ids
        chosen_all.extend(list(chosen))
        centre = np.mean(xyz[np.isin(superpoint, chosen)], axis=0)
        for sp in chosen:
            if graph.vs["v"][sp] != sp:
                raise ValueError("graph vertex ids must equal the superpoint ids (:1018)")
            off[sp] = centre - np.mean(xyz[superpoint == sp], axis=0)
    keep = np.zeros(graph.vcount, dtype=bool)
    keep[np.asarray(chosen_all, dtype=np.int64)] = True
    graph.vs["semantic_label"] = np.where(keep, graph.vs["semantic_label"], -100)
    graph.vs["instance_label"] = np.where(keep, graph.vs["instance_label"], -100)
    off[~keep] = 0.0
    graph.vs["superpoint_offset_vector"] = off
    return [int(c) for c in chosen_all]
