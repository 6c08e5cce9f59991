"""Synthetic ScanNet-shaped scenes, batch assembly and the per-iteration hot path of 3D-WSIS.

* ``make_scene``   -- seeded generator specified in SURVEY.md 8d (room = floor + 4 walls + boxes sampled at one
                      point per occupied 2 cm voxel plus ~30 % duplicates, 0.25 m superpoints, <0.3 m
                      superpoint graph, weak labels: one labelled superpoint per instance).
* ``collate``      -- the batch dict of modules/datasets/scannetv2_dataset.py:343-474, built
                      with the host ``pointgroup_ops.voxelization_idx``.
* ``to_device`` / ``train_step`` -- the iteration of train_scannetv2.py:143-252: H2D, superpoint centres,
                      ``voxelization`` -> ``SparseConvTensor`` -> ``Network`` -> ``MultiTaskLoss`` -> backward ->
                      ECC grad clamp -> AdamW step.
"""
import types

import os

import numpy as np
import torch

SCALE = 50            # config/ScanNet_v2_3D_WSIS.yaml:31 (2 cm voxels)
FULL_SCALE_MIN = 128  # :30 full_scale[0]


def default_cfg():
    model = types.SimpleNamespace(input_channel=3, use_coords=True, blocks=5, block_reps=2, media=32, classes=20,
                                  fix_module="[]")          # config/ScanNet_v2_3D_WSIS.yaml:37-45
    loss = types.SimpleNamespace(ignore_label=-100, supervise_instance_size=True, joint_training_epoch=0,
                                 semantic_dice=True, supervise_sp_offset=True)   # stage-3 switches (8d)
    return types.SimpleNamespace(model=model, loss=loss, mode=4, batch_size=1)


# -------------------------------------------------------------------------------------------------------------
def _face_voxels(origin, u, v, lu, lv, voxel):
    """voxel-centre samples of the rectangle origin + a*u + b*v, 0<=a<lu, 0<=b<lv (metres)."""
    na, nb = max(int(round(lu / voxel)), 1), max(int(round(lv / voxel)), 1)
    a, b = np.meshgrid((np.arange(na) + 0.5) * voxel, (np.arange(nb) + 0.5) * voxel, indexing="ij")
    return origin[None, :] + a.reshape(-1, 1) * u[None, :] + b.reshape(-1, 1) * v[None, :]


def _touching_superpoints(pts, superpoint, voxel):
    """undirected superpoint pairs whose voxels touch (26-neighbourhood) -- the synthetic counterpart of the
    mesh-face adjacency of data/ScanNetV2/prepare_data_inst_ScanNetV2.py:191-212"""
    v = np.floor(pts / voxel + 1e-6).astype(np.int64)
    v = v - v.min(0) + 1
    dims = v.max(0) + 2
    lin = (v[:, 0] * dims[1] + v[:, 1]) * dims[2] + v[:, 2]
    vkey, first = np.unique(lin, return_index=True)
    vsp = superpoint[first]
    pairs = []
    for dx in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dz in (-1, 0, 1):
                if (dx, dy, dz) <= (0, 0, 0):
                    continue                       # half space: every unordered voxel pair once
                nkey = vkey + (dx * dims[1] + dy) * dims[2] + dz
                pos = np.minimum(np.searchsorted(vkey, nkey), len(vkey) - 1)
                ok = vkey[pos] == nkey
                a, b = vsp[ok], vsp[pos[ok]]
                ne = a != b
                pairs.append(np.stack([np.minimum(a[ne], b[ne]), np.maximum(a[ne], b[ne])], 1))
    if not pairs:
        return np.zeros((0, 2), dtype=np.int64)
    return np.unique(np.concatenate(pairs), axis=0)


def make_scene(seed, room=(4.6, 3.6, 2.2), n_box=8, dup=0.3, voxel=0.02, sp_cell=0.25, classes=20,
               max_points=None, graph="knn8"):
    """returns a dict of numpy arrays describing one scene (all host side).
    ``graph``: "knn8" = both directions between centres closer than 0.3 m, <= 8 nearest (round-1 generator, the
    golden fixtures use it); "mesh" = superpoints whose voxels touch PLUS <= 5 KD-tree neighbours within 0.3 m, the
    rule of prepare_data_inst_ScanNetV2.py:191-225 (with ``sp_cell=0.19`` a C2 room has S ~ 2 k, E ~ 25 k)."""
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(seed)
    L, W, H = room
    ex, ey, ez = np.eye(3)
    faces = [(np.zeros(3), ex, ey, L, W),                                   # floor
             (np.zeros(3), ex, ez, L, H), (np.array([0, W, 0.]), ex, ez, L, H),   # walls y=0 / y=W
             (np.zeros(3), ey, ez, W, H), (np.array([L, 0, 0.]), ey, ez, W, H)]   # walls x=0 / x=L
    inst_of_face = [0, 1, 2, 3, 4]
    n_inst = 5
    for _ in range(n_box):
        sx, sy, sz = rng.uniform(0.3, 0.7), rng.uniform(0.3, 0.7), rng.uniform(0.3, 0.7)
        ox, oy = rng.uniform(0.1, L - sx - 0.1), rng.uniform(0.1, W - sy - 0.1)
        o = np.array([ox, oy, 0.0])
        box = [(o + np.array([0, 0, sz]), ex, ey, sx, sy),                 # top
               (o, ex, ez, sx, sz), (o + np.array([0, sy, 0]), ex, ez, sx, sz),
               (o, ey, ez, sy, sz), (o + np.array([sx, 0, 0]), ey, ez, sy, sz)]
        faces += box
        inst_of_face += [n_inst] * 5
        n_inst += 1
    pts, face_id = [], []
    for f, (o, u, v, lu, lv) in enumerate(faces):
        p = _face_voxels(np.asarray(o, float), u, v, lu, lv, voxel)
        pts.append(p)
        face_id.append(np.full(len(p), f))
    pts = np.concatenate(pts)
    face_id = np.concatenate(face_id)
    # one point per occupied voxel
    vox = np.floor(pts / voxel + 1e-6).astype(np.int64)
    _, first = np.unique(vox, axis=0, return_index=True)
    first.sort()
    pts, face_id, vox = pts[first], face_id[first], vox[first]
    jitter = (rng.random(pts.shape) - 0.5) * 0.6 * voxel
    base = (vox + 0.5) * voxel
    pts = base + jitter
    n_dup = int(dup * len(pts))
    d = rng.choice(len(pts), n_dup, replace=False)
    pts = np.concatenate([pts, base[d] + (rng.random((n_dup, 3)) - 0.5) * 0.6 * voxel])
    face_id = np.concatenate([face_id, face_id[d]])
    perm = rng.permutation(len(pts))
    if max_points is not None:
        perm = perm[:max_points]
    pts, face_id = pts[perm].astype(np.float32), face_id[perm]
    N = len(pts)
    rgb = rng.uniform(-1, 1, (N, 3)).astype(np.float32)

    # superpoints = 0.25 m cell x surface patch, dense renumbered
    cell = np.floor(pts / sp_cell).astype(np.int64)
    key = np.concatenate([face_id[:, None], cell], 1)
    _, superpoint = np.unique(key, axis=0, return_inverse=True)
    superpoint = superpoint.reshape(-1).astype(np.int64)
    S = int(superpoint.max()) + 1
    cnt = np.bincount(superpoint, minlength=S).astype(np.float64)
    centre = np.stack([np.bincount(superpoint, pts[:, j], S) / cnt for j in range(3)], 1)
    sp_face = np.zeros(S, dtype=np.int64)
    sp_face[superpoint] = face_id
    sp_inst = np.asarray(inst_of_face)[sp_face]

    # superpoint graph: both directions between centres closer than 0.3 m (<= 8 nearest), sorted tuples
    tree = cKDTree(centre)
    dist, nbr = tree.query(centre, k=min(9 if graph == "knn8" else 6, S), distance_upper_bound=0.3)
    und = set()
    if graph == "mesh":
        und.update((int(a), int(b)) for a, b in _touching_superpoints(pts, superpoint, voxel))
    else:
        assert graph == "knn8", graph
    for s in range(S):
        for dd, t in zip(dist[s][1:], nbr[s][1:]):
            if np.isfinite(dd) and t < S and t != s:
                und.add((min(s, int(t)), max(s, int(t))))
    if not und:
        und.add((0, min(1, S - 1)))
    edges = np.array(sorted(list(und) + [(b, a) for a, b in und]), dtype=np.int64)
    edge_feats = rng.standard_normal((len(edges), 13)).astype(np.float32)

    # labels: semantic per instance, weak supervision = one labelled superpoint per instance
    inst_sem = rng.integers(0, classes, n_inst)
    sp_sem_full = inst_sem[sp_inst]
    sp_sem = np.full(S, -100, dtype=np.int64)
    sp_ins = np.full(S, -100, dtype=np.int64)
    for i in range(n_inst):
        members = np.nonzero(sp_inst == i)[0]
        if len(members):
            c = members[rng.integers(len(members))]
            sp_sem[c], sp_ins[c] = sp_sem_full[c], i
    inst_centre = np.stack([np.array([pts[(sp_inst[superpoint] == i)].mean(0) if (sp_inst == i).any() else np.zeros(3)
                                      for i in range(n_inst)])])[0]
    sp_offset = (inst_centre[sp_inst] - centre).astype(np.float32)
    inst_vox = np.array([max(int(((sp_inst[superpoint]) == i).sum()), 1) for i in range(n_inst)])
    inst_size = np.array([np.linalg.norm(pts[sp_inst[superpoint] == i].max(0) - pts[sp_inst[superpoint] == i].min(0))
                          if (sp_inst == i).any() else 0.0 for i in range(n_inst)])
    return dict(xyz=pts, rgb=rgb, superpoint=superpoint, edges=edges, edge_feats=edge_feats,
                sem_label=sp_sem[superpoint], ins_label=sp_ins[superpoint], sp_sem=sp_sem, sp_ins=sp_ins,
                sp_offset=sp_offset, sp_voxnum=inst_vox[sp_inst].astype(np.float32),
                sp_size=inst_size[sp_inst].astype(np.float32), n_inst=n_inst, S=S)


def synthetic_predictions(scene, seed=0, noise=0.02):
    """plausible superpoint-level network outputs for a ``make_scene`` scene (inputs of the test-time grouping,
    test_scannetv2.py:244-260): a class per instance, ground-truth offsets / log-occupancy / size plus noise.
    -> (sp_semantic_pred int64 [S], offsets f32 [S,3], occupancy f32 [S,1], ins_size f32 [S,1])"""
    rng = np.random.default_rng(seed)
    S = scene["S"]
    sem = (np.round(scene["sp_size"] * 1000).astype(np.int64) % 20)          # constant per instance, 20 classes
    offsets = (scene["sp_offset"] + rng.normal(0, noise, (S, 3))).astype(np.float32)
    occupancy = (np.log(scene["sp_voxnum"]) + rng.normal(0, 0.1, S)).astype(np.float32).reshape(S, 1)
    size = (scene["sp_size"] * (1 + rng.normal(0, 0.05, S))).astype(np.float32).reshape(S, 1)
    return sem, offsets, occupancy, size


# -------------------------------------------------------------------------------------------------------------
def _assemble_host(scenes):
    """everything of ``collate_fn`` (scannetv2_dataset.py:343-474) that is plain concatenation: per-point and
    per-superpoint arrays with batch / superpoint / instance offsets, the two edge orders, the ECC graph -- no hashing"""
    from graphnet import GraphConvInfo
    locs, locs_float, feats, sem, ins, sps = [], [], [], [], [], []
    sp_sem, sp_ins, sp_off, sp_vox, sp_size = [], [], [], [], []
    edge_sorted, edge_feats_sorted, edges_ext = [], [], []
    batch_offsets, sp_batch_offsets = [0], [0]
    sp_bias, inst_bias = 0, 0
    for b, sc in enumerate(scenes):
        xyz = sc["xyz"]
        v = np.floor(xyz.astype(np.float64) * SCALE).astype(np.int64)
        v = v - v.min(0)
        locs.append(torch.cat([torch.full((len(v), 1), b, dtype=torch.int64), torch.from_numpy(v)], 1))
        locs_float.append(torch.from_numpy(xyz))
        feats.append(torch.from_numpy(sc["rgb"]))
        sem.append(torch.from_numpy(sc["sem_label"]))
        il = sc["ins_label"].copy()
        il[il != -100] += inst_bias
        ins.append(torch.from_numpy(il))
        sps.append(torch.from_numpy(sc["superpoint"] + sp_bias))
        sp_sem.append(torch.from_numpy(sc["sp_sem"]))
        si = sc["sp_ins"].copy()
        si[si != -100] += inst_bias
        sp_ins.append(torch.from_numpy(si))
        sp_off.append(torch.from_numpy(sc["sp_offset"]))
        sp_vox.append(torch.from_numpy(sc["sp_voxnum"]))
        sp_size.append(torch.from_numpy(sc["sp_size"]))
        E = sc["edges"]
        order = np.argsort(E[:, 1], kind="stable")          # ecc/GraphConvInfo.py:54 (sorted by target)
        edge_sorted.append(torch.from_numpy(E[order] + sp_bias))
        edge_feats_sorted.append(torch.from_numpy(sc["edge_feats"][order]))
        edges_ext.append(torch.from_numpy(E + sp_bias))     # original (sorted-tuple) order, :455-457
        sp_bias += sc["S"]
        inst_bias += sc["n_inst"]
        batch_offsets.append(batch_offsets[-1] + len(v))
        sp_batch_offsets.append(sp_bias)
    locs = torch.cat(locs, 0)
    spatial_shape = np.clip((locs.max(0)[0][1:] + 1).numpy(), FULL_SCALE_MIN, None)
    edge_indexes = torch.cat(edge_sorted, 0).t().contiguous()
    GIs = [GraphConvInfo(edge_indexes, torch.cat(edge_feats_sorted, 0), sp_bias)]
    edges = torch.cat(edges_ext, 0)
    return {
        "locs": locs,
        "locs_float": torch.cat(locs_float, 0).float(), "feats": torch.cat(feats, 0).float(),
        "semantic_labels": torch.cat(sem, 0).long(), "instance_labels": torch.cat(ins, 0).long(),
        "offsets": torch.tensor(batch_offsets, dtype=torch.int32), "spatial_shape": spatial_shape,
        "superpoint": torch.cat(sps, 0).long(), "GIs": GIs,
        "sp_batch_offsets": torch.tensor(sp_batch_offsets, dtype=torch.int32),
        "edge_u_list": edges[:, 0].contiguous().long(), "edge_v_list": edges[:, 1].contiguous().long(),
        # rows of scatter(..., edge_u): known here on the host, so the device step never has to read it back
        "edge_src_rows": (int(edges[:, 0].max()) + 1) if edges.shape[0] else 0,
        "superpoint_semantic_labels": torch.cat(sp_sem, 0).long(),
        "superpoint_instance_labels": torch.cat(sp_ins, 0).long(),
        "superpoint_offset_vector": torch.cat(sp_off, 0).float(),
        # (torch's CPU log, as the reference's collate_fn, scannetv2_dataset.py:438: numpy's differs in the last bit)
        "superpoint_instance_voxel_num": torch.log(torch.cat(sp_vox, 0).float()),
        "superpoint_instance_size": torch.cat(sp_size, 0).float(),
        "scene_list": [f"synthetic_{i}" for i in range(len(scenes))],
    }


def collate(scenes, mode=4, n_levels=5):
    """Batch dict with the schema of scannetv2_dataset.py:460-474; host tensors: the voxel hash runs on
    ONE host thread (libwsis_host.so), as in the reference's DataLoader workers (:445-449).
    ``n_levels``: UNet depth the host-side ``level_counts`` are computed for (config ``blocks``)."""
    import pointgroup_ops
    import spconv
    out = _assemble_host(scenes)
    voxel_locs, p2v_map, v2p_map = pointgroup_ops.voxelization_idx(out["locs"], len(scenes), mode)
    out.update(voxel_locs=voxel_locs, p2v_map=p2v_map, v2p_map=v2p_map)
    # active voxels of the UNet's strided levels, from the host-side coordinates (spconv.ops.level_voxel_counts)
    out["level_counts"] = spconv.ops.level_voxel_counts(voxel_locs.numpy(), out["spatial_shape"], n_levels)
    return out


def collate_device(scenes, device, mode=4, n_levels=5):
    """The same batch with the hashing on the DEVICE (``voxelization_idx`` off the loader's host threads):
    the host only concatenates the raw per-scene arrays and ships them; the GPU voxelizer (bit-exact first-occurrence
    contract) builds voxel_locs / p2v / v2p and the pyramid's level counts come from one sort per level -- two small
    read-backs in this loader stage (voxel count + list width, then the level counts), none in the training step.
    Returns what ``to_device(collate(scenes))`` returns."""
    import pointgroup_ops
    import spconv
    host = _assemble_host(scenes)
    locs_d = host["locs"].to(device, non_blocking=True)
    voxel_locs, p2v_map, v2p_map = pointgroup_ops.voxelization_idx(locs_d, len(scenes), mode)
    counts = spconv.ops.level_voxel_counts_device(voxel_locs, host["spatial_shape"], n_levels)
    host.update(voxel_locs=voxel_locs, p2v_map=p2v_map, v2p_map=v2p_map)
    host["level_counts"] = [int(v) for v in counts.tolist()]
    return to_device(host, device)


# ---- packed scenes: the loader's per-sample output as ONE pinned buffer, the batch assembled on the device -------------
_PACK_FIELDS = (("xyz", np.float32, 3), ("rgb", np.float32, 3), ("sem_label", np.int64, 0), ("ins_label", np.int64, 0),
                ("superpoint", np.int64, 0), ("sp_sem", np.int64, 0), ("sp_ins", np.int64, 0), ("sp_offset", np.float32, 3),
                ("sp_voxnum", np.float32, 0), ("sp_size", np.float32, 0), ("edges", np.int64, 2), ("edge_feats", np.float32, 13),
                ("vmin", np.int64, 0))


def pack_scene(sc, pin=True, buf=None):
    """What a loader worker hands over per sample (the output of ``__getitem__``, scannetv2_dataset.py:96-190): every
    per-scene array in its final dtype, back to back in ONE (pinned) buffer, plus the few numbers the host needs to lay
    out the batch without looking at the data again -- point / superpoint / edge / instance counts, the voxel extent
    (floor is monotonic: the extent of floor(xyz * 50) follows from xyz.min / xyz.max alone), the largest edge source and
    the instance-slot bound.  ``collate_packed`` ships the buffer with one H2D copy and does the concatenations, label
    offsets, edge sort, voxel hash and level counts on the device."""
    xyz = np.ascontiguousarray(sc["xyz"], dtype=np.float32)
    # (column by column: numpy's reduction along axis 0 of an [N, 3] array is ten times slower)
    lo = np.floor(np.array([xyz[:, j].min() for j in range(3)], dtype=np.float64) * SCALE).astype(np.int64)
    hi = np.floor(np.array([xyz[:, j].max() for j in range(3)], dtype=np.float64) * SCALE).astype(np.int64)
    # (the log of the voxel counts with torch's CPU log, exactly as ``_assemble_host`` and the reference's collate_fn,
    # scannetv2_dataset.py:438, take it: numpy's and the device's log differ in the last bit.  A few thousand values:
    # below the size at which a torch CPU op wakes its thread pool)
    voxnum = torch.from_numpy(np.ascontiguousarray(sc["sp_voxnum"], dtype=np.float32))
    arrays = dict(sc, xyz=xyz, vmin=lo, sp_voxnum=torch.log(voxnum).numpy())
    layout, off = {}, 0
    parts = []
    for name, dt, width in _PACK_FIELDS:
        a = np.ascontiguousarray(arrays[name], dtype=dt)
        layout[name] = (off, a.shape, dt)
        parts.append((off, a))
        off = (off + a.nbytes + 15) // 16 * 16
    if buf is None or buf.numel() < off:
        buf = torch.empty(off, dtype=torch.uint8, pin_memory=bool(pin and torch.cuda.is_available()))
    view = buf.numpy()
    for o, a in parts:
        view[o:o + a.nbytes] = a.view(np.uint8).reshape(-1)
    sp_ins, E = sc["sp_ins"], sc["edges"]
    return {"buf": buf, "layout": layout, "N": int(xyz.shape[0]), "S": int(sc["S"]), "E": int(E.shape[0]),
            "n_inst": int(sc["n_inst"]), "extent": (hi - lo + 1), "edge_src_max": int(E[:, 0].max()) if len(E) else -1,
            "sp_ins_max": int(sp_ins.max()) if len(sp_ins) else -100}


_TORCH_DT = {np.float32: torch.float32, np.int64: torch.int64}


def collate_packed(packs, device, mode=4, n_levels=5):
    """Batch of packed scenes (``pack_scene``) assembled ON THE DEVICE: per scene one H2D copy of its pinned buffer; the
    concatenations, batch / superpoint / instance offsets (scannetv2_dataset.py:383-396), the target-sorted edge order of
    the ECC graph (ecc/GraphConvInfo.py:54), voxel hash and pyramid counts all run there.  The host thread only lays out
    the batch from the packs' counts (a few dozen Python statements per scene).  Returns what
    ``to_device(collate(scenes))`` returns."""
    import pointgroup_ops
    import spconv
    from graphnet import GraphConvInfo
    dev = torch.device(device)
    cols = {k: [] for k in ("locs", "locs_float", "feats", "sem", "ins", "sps", "sp_sem", "sp_ins", "sp_off", "sp_vox",
                            "sp_size", "edge_sorted", "edge_feats_sorted", "edges_ext")}
    batch_offsets, sp_batch_offsets = [0], [0]
    sp_bias, inst_bias = 0, 0
    extent = np.zeros(3, dtype=np.int64)
    edge_src_rows, slots = 0, []
    for b, pk in enumerate(packs):
        d = pk["buf"].to(dev, non_blocking=True)

        def f(name):
            off, shape, dt = pk["layout"][name]
            n = int(np.prod(shape)) * np.dtype(dt).itemsize
            return d[off:off + n].view(_TORCH_DT[dt]).view(shape)
        xyz = f("xyz")
        v = torch.floor(xyz.double() * SCALE).long() - f("vmin")
        cols["locs"].append(torch.cat([torch.full((pk["N"], 1), b, dtype=torch.int64, device=dev), v], 1))
        cols["locs_float"].append(xyz)
        cols["feats"].append(f("rgb"))
        cols["sem"].append(f("sem_label"))
        il, si = f("ins_label"), f("sp_ins")
        cols["ins"].append(torch.where(il != -100, il + inst_bias, il) if inst_bias else il)
        cols["sp_ins"].append(torch.where(si != -100, si + inst_bias, si) if inst_bias else si)
        cols["sps"].append(f("superpoint") + sp_bias if sp_bias else f("superpoint"))
        cols["sp_sem"].append(f("sp_sem"))
        cols["sp_off"].append(f("sp_offset"))
        cols["sp_vox"].append(f("sp_voxnum"))
        cols["sp_size"].append(f("sp_size"))
        E = f("edges")
        order = torch.argsort(E[:, 1], stable=True)
        Eb = E + sp_bias if sp_bias else E
        cols["edge_sorted"].append(Eb[order])
        cols["edge_feats_sorted"].append(f("edge_feats")[order])
        cols["edges_ext"].append(Eb)
        if pk["edge_src_max"] >= 0:
            edge_src_rows = max(edge_src_rows, pk["edge_src_max"] + sp_bias + 1)
        # bound of the (batch-offset) instance ids of the scene's superpoints: to_device reads it from the label tensor
        slots.append(pk["sp_ins_max"] + inst_bias + 1 if pk["sp_ins_max"] >= 0 else 1)
        extent = np.maximum(extent, pk["extent"])
        sp_bias += pk["S"]
        inst_bias += pk["n_inst"]
        batch_offsets.append(batch_offsets[-1] + pk["N"])
        sp_batch_offsets.append(sp_bias)

    def cat(k):
        return cols[k][0] if len(cols[k]) == 1 else torch.cat(cols[k], 0)
    locs = cat("locs").contiguous()
    edges = cat("edges_ext")
    out = {
        "locs": locs, "locs_float": cat("locs_float").contiguous(), "feats": cat("feats").contiguous(),
        "semantic_labels": cat("sem").contiguous(), "instance_labels": cat("ins").contiguous(),
        "offsets": torch.tensor(batch_offsets, dtype=torch.int32),
        "spatial_shape": np.clip(extent, FULL_SCALE_MIN, None),
        "superpoint": cat("sps").contiguous(),
        "GIs": [GraphConvInfo(cat("edge_sorted").t().contiguous(), cat("edge_feats_sorted").contiguous(), sp_bias)],
        "sp_batch_offsets": torch.tensor(sp_batch_offsets, dtype=torch.int32),
        "edge_u_list": edges[:, 0].contiguous(), "edge_v_list": edges[:, 1].contiguous(),
        "edge_src_rows": edge_src_rows,
        "superpoint_semantic_labels": cat("sp_sem").contiguous(), "superpoint_instance_labels": cat("sp_ins").contiguous(),
        "superpoint_offset_vector": cat("sp_off").contiguous(),
        "superpoint_instance_voxel_num": cat("sp_vox").contiguous(),
        "superpoint_instance_size": cat("sp_size").contiguous(),
        "scene_list": [f"synthetic_{i}" for i in range(len(packs))],
        "sp_instance_slots": slots,
    }
    voxel_locs, p2v_map, v2p_map = pointgroup_ops.voxelization_idx(locs, len(packs), mode)
    counts = spconv.ops.level_voxel_counts_device(voxel_locs, out["spatial_shape"], n_levels)
    out.update(voxel_locs=voxel_locs, p2v_map=p2v_map, v2p_map=v2p_map)
    out["level_counts"] = [int(c) for c in counts.tolist()]
    out["voxel_coords_int"] = voxel_locs.int().contiguous()
    ev = torch.cuda.Event()
    ev.record()
    out["coords_ready_event"] = ev
    build_batch_graphs(out)
    return out


_DEVICE_KEYS = ("voxel_locs", "p2v_map", "v2p_map", "locs_float", "feats", "semantic_labels", "instance_labels",
                "superpoint", "edge_u_list", "edge_v_list", "superpoint_semantic_labels",
                "superpoint_instance_labels", "superpoint_offset_vector", "superpoint_instance_voxel_num",
                "superpoint_instance_size")


def to_device(batch, device):
    """H2D of train_scannetv2.py:149-172 plus the per-batch graph structures (CSR of the superpoint ids and of
    the edge lists) that are constant for the batch."""
    out = dict(batch)
    for k in _DEVICE_KEYS:
        out[k] = batch[k].to(device)          # (tensors a device-side collate already left on the GPU stay put)
    out["voxel_coords_int"] = out["voxel_locs"].int().contiguous()
    out["GIs"][0].cuda()
    ev = torch.cuda.Event()                # every index tensor of the batch is on the device behind this point
    ev.record()
    out["coords_ready_event"] = ev
    build_batch_graphs(out)
    # bound of the superpoint instance ids per scene, read while the labels are still host tensors: lets the loss
    # place the instances in fixed slots instead of torch.unique (sync) or an [S, S] same-instance matrix
    lab, offs = batch["superpoint_instance_labels"], [int(o) for o in batch["sp_batch_offsets"]]
    out["sp_instance_slots"] = [max(int(lab[b:e].max()) + 1, 1) if e > b else 1 for b, e in zip(offs[:-1], offs[1:])]
    return out


def build_batch_graphs(batch, side_stream=None):
    """per-batch device structures of the segmented reductions: CSR of the superpoint ids and of the point->voxel
    map, both edge directions of the superpoint graph and of the ECC graph (``GIs``).  Part of this build's ``scatter``
    cost for every NEW batch (upstream torch_scatter pays it as atomics inside every call), so ``bench.py`` rebuilds
    them inside the timed step.

    They only depend on the batch's index tensors, so -- like the rulebooks -- they are built on the side stream
    (``WSIS_GRAPH_STREAM=0`` / ``side_stream=False``: on the current stream): the ~50 small sort launches run next to
    whatever the main stream still has queued (the previous iteration's backward pass) instead of in front of the
    forward pass; the main stream joins before its first use."""
    from torch_scatter import SegmentCSR
    import wsis_ops
    import spconv
    S = int(batch["sp_batch_offsets"][-1])
    dev = batch["superpoint"].device
    if side_stream is None:
        side_stream = os.environ.get("WSIS_GRAPH_STREAM", "1") != "0"
    side = spconv.ops._side_stream(dev) if (side_stream and dev.type == "cuda") else None
    main = torch.cuda.current_stream(dev) if dev.type == "cuda" else None
    if side is not None:
        ev = batch.get("coords_ready_event")       # recorded after the H2D copies of the index tensors
        if ev is not None:
            side.wait_event(ev)
        else:
            side.wait_stream(main)
    with (torch.cuda.stream(side) if side is not None else spconv.ops._NullCtx()):
        eu, ev = batch["edge_u_list"], batch["edge_v_list"]
        n_src = batch.get("edge_src_rows")
        gis = [gi for gi in batch.get("GIs", [])
               if getattr(gi, "_edge_indexes", None) is not None and gi._edge_indexes.is_cuda]
        if (dev.type == "cuda" and n_src is not None and len(gis) <= 1 and eu.numel() > 0
                and os.environ.get("WSIS_CSR_BATCH", "1") != "0"):
            # all per-batch CSRs from ONE sort (torch_scatter.segment_csr_batch): ~12 launches instead of ~60
            from torch_scatter import segment_csr_batch
            pairs = [(batch["superpoint"], S), (batch["p2v_map"], int(batch["voxel_locs"].shape[0])),
                     (eu, int(n_src)), (ev, S)]
            for gi in gis:
                pairs += [(gi._edge_indexes[0], gi.num_nodes), (gi._edge_indexes[1], gi.num_nodes)]
            got = segment_csr_batch(pairs)
            batch["superpoint_csr"], batch["p2v_csr"] = got[0], got[1]
            batch["edge_graph"] = wsis_ops.EdgeGraph(eu, ev, S, num_src=n_src, csr_u=got[2], csr_v=got[3])
            for gi in gis:
                gi._csr, gi._csr_dst = got[4], got[5]
            csrs = got
        else:
            batch["superpoint_csr"] = SegmentCSR(batch["superpoint"], S)
            batch["p2v_csr"] = SegmentCSR(batch["p2v_map"], int(batch["voxel_locs"].shape[0]))
            batch["edge_graph"] = wsis_ops.EdgeGraph(eu, ev, S, num_src=n_src)
            csrs = [batch["superpoint_csr"], batch["p2v_csr"], batch["edge_graph"].csr_u, batch["edge_graph"].csr_v]
            for gi in gis:                             # a new batch has a new GraphConvInfo: its two CSRs are per-batch too
                gi._csr = gi._csr_dst = None
                csrs += [gi.csr(), gi.csr_dst()]
    if side is not None:
        main.wait_stream(side)
        for c in csrs:                             # allocated on the side stream, consumed on the main stream
            for t in (c.perm, c.offsets, c.index):
                t.record_stream(main)
    return batch


def bench_scene(seed, **kw):
    """the bench workload's scene: the C2 room of SURVEY 8d with the superpoint graph sized like a prepared ScanNet
    scene (S ~ 2.3 k superpoints, E ~ 20 k directed edges: touching superpoints + <= 5 KD-tree neighbours within
    0.3 m, prepare_data_inst_ScanNetV2.py:191-225)"""
    kw.setdefault("sp_cell", 0.19)
    kw.setdefault("graph", "mesh")
    return make_scene(seed, **kw)


def forward_loss(model, criterion, batch, cfg, epoch=5):
    """train_scannetv2.py:174-232 on a device batch -> (loss, ret)"""
    import pointgroup_ops
    import spconv
    from torch_scatter import scatter
    coords_float = batch["locs_float"]
    superpoint = batch["superpoint"]
    centre = scatter(coords_float, superpoint, dim=0, reduce="mean", csr=batch.get("superpoint_csr"))
    extra = {"superpoint": superpoint, "GIs": batch["GIs"], "edge_u_list": batch["edge_u_list"],
             "edge_v_list": batch["edge_v_list"], "superpoint_cenetr_xyz": centre,
             "superpoint_csr": batch.get("superpoint_csr"), "edge_graph": batch.get("edge_graph"),
             "p2v_csr": batch.get("p2v_csr"), "edge_src_rows": batch.get("edge_src_rows")}
    feats = batch["feats"]
    if cfg.model.use_coords:
        feats = torch.cat((feats, coords_float), 1)
    voxel_feats = pointgroup_ops.voxelization(feats, batch["v2p_map"], cfg.mode)
    # batch_size as the reference passes it (the configured one) -- but never below the scenes this batch really holds:
    # the rulebook build sizes its output-cell bitmap by it (upstream sizes its dense grid the same way)
    n_scenes = int(batch["sp_batch_offsets"].shape[0]) - 1 if "sp_batch_offsets" in batch else 0
    input_ = spconv.SparseConvTensor(voxel_feats, batch["voxel_coords_int"], batch["spatial_shape"],
                                     max(int(cfg.batch_size), n_scenes))
    input_._ready_event = batch.get("coords_ready_event")
    counts = batch.get("level_counts")
    if counts is not None and len(counts) == model.blocks - 1:
        input_._level_counts = counts
    rulebooks = batch.get("rulebooks")        # built ahead by a spconv.ops.RulebookPrefetcher (loader stage)
    if rulebooks is not None:
        rulebooks.attach(input_)
    ret = model(input_, batch["p2v_map"], extra)
    loss_inp = {
        "point_labels": (batch["semantic_labels"], batch["instance_labels"]),
        "semantic_scores": ret["semantic_scores"],
        "superpoint_labels": (batch["superpoint_semantic_labels"], batch["superpoint_instance_labels"]),
        "sp_semantic": ret["sp_semantic_scores"],
        "sp_offset_vector": (ret["pred_sp_offset_vectors"], batch["superpoint_offset_vector"]),
        "sp_occupancy": (ret["pred_sp_occupancy"], batch["superpoint_instance_voxel_num"]),
        "sp_instance_size": (ret["pred_sp_ins_size"], batch["superpoint_instance_size"]),
        "sp_discriminative_features": (ret["sp_discriminative_feats"], batch["sp_batch_offsets"]),
    }
    if "sp_instance_slots" in batch and os.environ.get("WSIS_LOSS_SLOTS", "1") != "0":
        loss_inp["sp_instance_slots"] = batch["sp_instance_slots"]
    loss, loss_out = criterion(loss_inp, epoch)
    return loss, ret


def train_step(model, criterion, optimizer, batch, cfg, epoch=5, grad_sync=None, pipeline=None):
    """one iteration of train_scannetv2.py:143-252 (forward, loss, backward, ECC grad clamp, AdamW step).
    ``pipeline``: a started spconv.ops.RulebookPipeline of the NEXT batch, advanced between the phases."""
    loss, ret = forward_loss(model, criterion, batch, cfg, epoch)
    if pipeline is not None:
        pipeline.pump()
    optimizer.zero_grad(set_to_none=True)
    loss.backward()
    if pipeline is not None:
        pipeline.pump()
    if grad_sync is not None:
        grad_sync(model)
    # train_scannetv2.py:247-249: the ECC gradients clamped to [-1, 1] -- inside the one-launch AdamW (the foreach forms
    # still ran one kernel per tensor: 20 launches in front of the optimizer), or as multi-tensor ops for another optimizer
    if hasattr(optimizer, "set_grad_clamp"):
        if getattr(optimizer, "_ecc_clamp_of", None) is not model:
            optimizer.set_grad_clamp(list(model.ecc.parameters()), 1.0)
            optimizer._ecc_clamp_of = model
    else:
        grads = [p.grad for p in model.ecc.parameters() if p.grad is not None]
        if grads:
            torch._foreach_clamp_min_(grads, -1.0)
            torch._foreach_clamp_max_(grads, 1.0)
    optimizer.step()
    if pipeline is not None:
        pipeline.pump()
    return loss.detach(), ret


def make_pipeline(model):
    """rulebook pipeline for ``model``'s UNet pyramid (next batch built while the current step runs)"""
    import spconv
    return spconv.ops.RulebookPipeline(model.blocks)


def start_rulebooks(pipeline, batch):
    pipeline.start(batch["voxel_coords_int"], batch["spatial_shape"], batch.get("coords_ready_event"))


def make_prefetcher(model):
    """rulebook prefetcher for ``model``'s UNet pyramid (one batch in flight)"""
    import spconv
    return spconv.ops.RulebookPrefetcher(model.blocks)


def prefetch_rulebooks(prefetcher, batch):
    """start building the rulebooks of device batch ``batch`` in the background (loader stage)"""
    prefetcher.submit(batch["voxel_coords_int"], batch["spatial_shape"], batch.get("coords_ready_event"))


def build_model(cfg, device, seed=123):
    import backbone_3D_WSIS
    import losses_3D_WSIS
    torch.manual_seed(seed)   # config/ScanNet_v2_3D_WSIS.yaml:3
    model = backbone_3D_WSIS.Network(cfg.model).to(device)
    criterion = losses_3D_WSIS.MultiTaskLoss(None, cfg.loss, cfg.model)
    # yaml:58-61.  On the GPU: optim.FlatAdamW (the same update rule in ONE launch over all 361 tensors;
    # WSIS_FLAT_ADAMW=0 selects torch's multi-tensor fused AdamW: 6 launches at a third of the CUs)
    if torch.device(device).type == "cuda" and os.environ.get("WSIS_FLAT_ADAMW", "1") != "0":
        import wsis_optim as optim
        optimizer = optim.FlatAdamW(model.parameters(), lr=1e-3, weight_decay=1e-4)
    else:
        optimizer = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=1e-4,
                                      fused=torch.device(device).type == "cuda")
    return model, criterion, optimizer


def train_step_smoke(device="cuda:0"):
    """tiny fwd+bwd+step of the whole path (used by __graft_entry__.smoke)."""
    cfg = default_cfg()
    scene = make_scene(0, room=(1.2, 1.0, 0.8), n_box=1)
    batch = to_device(collate([scene]), device)
    model, crit, opt = build_model(cfg, device)
    loss, ret = train_step(model, crit, opt, batch, cfg)
    assert torch.isfinite(loss).item(), "non-finite loss in smoke step"
    assert ret["edge_affinity"].shape[0] == batch["edge_u_list"].shape[0]
    return float(loss)


def save_checkpoint(model, filename, optimizer=None, meta=None):
    """checkpoint in the reference's container (utils/checkpoint.py:205-262): ``{"meta", "model", "optimizer"}``,
    weights on the CPU, a DataParallel-style wrapper unwrapped."""
    import time as _time
    model = getattr(model, "module", model)
    meta = dict(meta or {})
    meta.update(time=_time.asctime())
    ck = {"meta": meta, "model": {k: v.detach().cpu() for k, v in model.state_dict().items()}}
    if optimizer is not None:
        ck["optimizer"] = optimizer.state_dict()
    os.makedirs(os.path.dirname(os.path.abspath(filename)), exist_ok=True)
    torch.save(ck, filename)
    return ck


def load_checkpoint(model, filename, map_location="cpu", strict=True, optimizer=None):
    """``utils/checkpoint.py:105-135``: the state dict sits under "model" (or "state_dict", or is the file itself),
    a leading "module." is stripped; the published checkpoints of the reference load this way (the layout is pinned
    by tests/test_state_dict_layout.py).  Read with ``wsis_datasets.safe_torch_load`` (``weights_only=True`` + the numpy
    reconstructors): a checkpoint is somebody else's file and nothing of it is executed -- one that carries anything
    but tensors, arrays, containers, strings and numbers is refused.  Returns the whole checkpoint dict."""
    from wsis_datasets import safe_torch_load
    ck = safe_torch_load(filename, map_location=map_location)
    if not isinstance(ck, dict):
        raise RuntimeError(f"No state_dict found in checkpoint file {filename}")
    sd = ck.get("model", ck.get("state_dict", ck))
    if sd and next(iter(sd)).startswith("module."):
        sd = {k[7:]: v for k, v in sd.items()}
    getattr(model, "module", model).load_state_dict(sd, strict=strict)
    if optimizer is not None and "optimizer" in ck:
        optimizer.load_state_dict(ck["optimizer"])
    return ck
