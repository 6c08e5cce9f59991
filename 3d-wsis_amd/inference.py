"""Test-time instance grouping on the superpoint graph: drop-in for ``clustering_in_graph`` of the reference's
``test_scannetv2.py:281-455`` (called at :257-260 for every scene; ``test_s3dis.py`` uses the same routine).

Same signature, same return values ``(conf [n] float, label_id [n] int, ins_mask [n, N] int)``.  The reference
builds a boolean point mask per superpoint (``superpoint == spID``: O(S*N)), ORs masks while it grows a group and
calls the CPU ``voxelization_idx`` once per group.  Here the per-point work is segmented and runs on the MI355X:

  superpoint centres        one segmented mean over the points        (torch_scatter drop-in -> wsis_segment_reduce)
  graph BFS                 wsis_host_graph_bfs over the S superpoints (host, O(S + E), seed order = reference)
  group voxel counts        ONE wsis_voxelize_idx over (group id, floor-to-zero(xyz*50)) of all grouped points
  instance masks            one gather + compare on the device

The per-group scalars (occupancy, radii, centres) and the fragment absorption are the reference's expressions on
<= a few hundred groups (host numpy).  The group sets equal the reference's (the acceptance test depends on the seed
only, so a group is a connected component of accepted edges among unvisited superpoints -- independent of the
visiting order); float results agree to rounding (member order of a Python ``set`` is not reproducible).
"""
from math import sqrt

import numpy as np
import torch

import pointgroup_ops
import wsis_native as _n
from torch_scatter import scatter

# test_scannetv2.py:288-289
SEMANTIC_IND2LABEL = np.array([1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 14, 16, 24, 28, 33, 34, 36, 39])
INSTANCE_VALID_LABELS = np.array([3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 14, 16, 24, 28, 33, 34, 36, 39])


def adjacency_csr(graph, S):
    """``graph``: an igraph-like object (``neighbors(vertex=, mode='all')``) or a pair of edge index arrays
    (u, v).  -> (offsets int32 [S+1], neighbours int32 [nnz]) with mode='all' semantics (both directions)."""
    if hasattr(graph, "neighbors"):
        lists = [np.asarray(graph.neighbors(vertex=s, mode="all"), dtype=np.int64) for s in range(S)]
        off = np.zeros(S + 1, dtype=np.int32)
        off[1:] = np.cumsum([len(l) for l in lists])
        adj = np.concatenate(lists).astype(np.int32) if S else np.zeros(0, np.int32)
        return off, adj
    u, v = (np.asarray(a, dtype=np.int64).reshape(-1) for a in graph)
    src = np.concatenate([u, v])
    dst = np.concatenate([v, u])
    order = np.lexsort((dst, src))
    src, dst = src[order], dst[order]
    off = np.zeros(S + 1, dtype=np.int32)
    off[1:] = np.cumsum(np.bincount(src, minlength=S))
    return off, dst.astype(np.int32)


# test_s3dis.py:297-541: 13 classes, ceiling / floor / wall (0, 1, 2) are stuff, growth radius 0.8 * size
S3DIS_LABEL_IDX = np.array([1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13])
S3DIS_VALID_LABELS = S3DIS_LABEL_IDX[3:]


def graph_bfs(label, class_valid, centre, ins_size, adj_off, adj):
    """wsis_host_graph_bfs -> (group int32 [S], n_groups)"""
    import ctypes
    S = int(label.shape[0])
    label = np.ascontiguousarray(label, dtype=np.int32)
    class_valid = np.ascontiguousarray(class_valid, dtype=np.uint8)
    centre = np.ascontiguousarray(centre, dtype=np.float32)
    ins_size = np.ascontiguousarray(ins_size, dtype=np.float32).reshape(-1)
    adj_off = np.ascontiguousarray(adj_off, dtype=np.int32)
    adj = np.ascontiguousarray(adj, dtype=np.int32)
    group = np.empty(S, dtype=np.int32)
    ng = ctypes.c_int64(0)
    _n.check_host(_n.host().wsis_host_graph_bfs(label.ctypes.data, class_valid.ctypes.data, len(class_valid),
                                                centre.ctypes.data, ins_size.ctypes.data, adj_off.ctypes.data,
                                                adj.ctypes.data, S, group.ctypes.data, ctypes.addressof(ng)),
                  "graph_bfs")
    return group, int(ng.value)


def clustering_in_graph(scene_name, xyz_origin, superpoint, graph, sp_semnatic_pred, pred_sp_offset_vectors,
                        pred_sp_occupancy, pred_sp_ins_size, device="cuda", semantic_ind2label=SEMANTIC_IND2LABEL,
                        valid_labels=INSTANCE_VALID_LABELS, radius_factor=0.25, stuff_classes=()):
    """``radius_factor``: 0.25 (ScanNet, test_scannetv2.py:331) / 0.8 (S3DIS, test_s3dis.py:349).
    ``stuff_classes``: predicted classes reported as ONE instance each (confidence 1) when they cover more than 100
    points -- ceiling and floor of test_s3dis.py:524-531, appended after the grouped instances.  The reference's
    S3DIS walls additionally go through open3d's RANSAC ``segment_plane`` (utils/planeSegment.py), which is not part
    of this path."""
    assert len(xyz_origin) == len(superpoint)
    N, S = len(xyz_origin), len(sp_semnatic_pred)
    dev = torch.device(device)
    if dev.type != "cuda":
        raise _n.WsisError("clustering_in_graph runs its per-point stages on the MI355X (there is no CPU fallback)")
    xyz_h = np.ascontiguousarray(xyz_origin, dtype=np.float32)
    sp_h = np.ascontiguousarray(superpoint).astype(np.int64)
    label = np.asarray(sp_semnatic_pred).astype(np.int64)
    offs = np.asarray(pred_sp_offset_vectors, dtype=np.float32)
    occ = np.asarray(pred_sp_occupancy, dtype=np.float32).reshape(S, -1)
    size = np.asarray(pred_sp_ins_size, dtype=np.float32).reshape(S, -1)

    xyz = torch.from_numpy(xyz_h).to(dev)
    sp = torch.from_numpy(sp_h).to(dev)
    # superpoint centre + predicted offset = predicted instance centre (test_scannetv2.py:299-305)
    centre = scatter(xyz, sp, dim=0, reduce="mean")[:S]
    sp_count = torch.bincount(sp, minlength=S)[:S]
    inst_centre = (centre + torch.from_numpy(offs).to(dev)).cpu().numpy()
    sp_count_h = sp_count.cpu().numpy().astype(np.int64)

    class_valid = np.isin(semantic_ind2label, valid_labels)
    adj_off, adj = adjacency_csr(graph, S)
    # the operator compares against 0.25 * size[seed]: other radii go in through the size argument
    seed_size = size[:, 0] if radius_factor == 0.25 else (size[:, 0] * np.float32(radius_factor / 0.25))
    group, n_groups = graph_bfs(label, class_valid, inst_centre, seed_size, adj_off, adj)
    if n_groups == 0 and not stuff_classes:
        return np.array([]), np.array([]), np.array([])

    # points -> group id; voxels per group with ONE voxelization_idx over (group, trunc(xyz * 50))
    group_d = torch.from_numpy(group).to(dev)
    pg = group_d[sp]                                               # [N] group of every point, -1 = ungrouped
    if n_groups > 0:
        sel = torch.nonzero(pg >= 0).flatten()
        vox = (xyz[sel] * 50).long()                               # float32 product, truncation: as :381-383
        coords = torch.cat([pg[sel].long().unsqueeze(1), vox], 1).contiguous()
        voxel_locs, _, _ = pointgroup_ops.voxelization_idx(coords, n_groups, 4)
        group_voxels = torch.bincount(voxel_locs[:, 0], minlength=n_groups).cpu().numpy()
        group_n = torch.bincount(pg[sel].long(), minlength=n_groups).cpu().numpy()

    members = [np.nonzero(group == g)[0] for g in range(n_groups)]     # ascending superpoint ids
    seed_label = np.array([label[m[0]] for m in members])

    def occupancy_of(m):          # get_group_pred_occupancy, :345-349
        return np.exp(occ[m]).mean()

    def centre_of(m):             # get_group_instance_center, :352-360 (float64 accumulation)
        w = sp_count_h[m].astype(np.float64)
        return (inst_centre[m].astype(np.float64) * w[:, None]).sum(0) / w.sum()

    def size_of(m):               # get_group_instance_size, :362-364
        return np.mean(size[m])

    primaries, fragments = [], []
    for g in range(n_groups):
        m = members[g]
        group_occ = occupancy_of(m)
        if group_voxels[g] < 0.3 * group_occ:                          # :388-393
            fragments.append({"groups": [g], "members": m, "classLabel": seed_label[g], "centre": centre_of(m),
                              "group_n": int(group_n[g])})
        else:
            r_set = max(0.01 * sqrt(group_n[g]), 0.02 * sqrt(group_occ), size_of(m))   # :395-399
            primaries.append({"groups": [g], "members": m, "classLabel": seed_label[g], "centre": centre_of(m),
                              "r_set": r_set, "group_n": int(group_n[g])})

    for frag in fragments:                                             # :410-438
        index, dis_min = -1, float("inf")
        for i, prim in enumerate(primaries):
            dis = np.linalg.norm(frag["centre"] - prim["centre"], ord=2)
            if frag["classLabel"] == prim["classLabel"] and dis < dis_min:
                index, dis_min = i, dis
        if not primaries:
            break
        closest = primaries[index]
        if dis_min < closest["r_set"]:
            m = np.concatenate([frag["members"], closest["members"]])
            n_pts = frag["group_n"] + closest["group_n"]               # the masks are disjoint: |a or b| = |a| + |b|
            closest["r_set"] = max(0.02 * sqrt(occupancy_of(m)), 0.01 * sqrt(n_pts), closest["r_set"], size_of(m))
            closest["centre"] = centre_of(m)
            closest["group_n"] = n_pts
            closest["members"] = np.concatenate([closest["members"], frag["members"]])
            closest["groups"] += frag["groups"]

    # ---- results (:441-455) -------------------------------------------------------------------------------
    conf, label_id = [], []
    inst_of_group = np.full(max(n_groups, 1), -1, dtype=np.int64)
    for i, prim in enumerate(primaries):
        conf.append(min(prim["group_n"] / occupancy_of(prim["members"]), 1))
        label_id.append(semantic_ind2label[prim["classLabel"]])
        inst_of_group[prim["groups"]] = i
    mask_rows = []
    if primaries:
        inst_d = torch.from_numpy(inst_of_group).to(dev)
        pi = torch.where(pg >= 0, inst_d[pg.clamp(min=0).long()], torch.full_like(pg, -1, dtype=torch.int64))
        mask_rows.append((pi.unsqueeze(0) == torch.arange(len(primaries), device=dev).unsqueeze(1)).to(torch.int64))
    if stuff_classes:                                                  # test_s3dis.py:524-531
        point_label = torch.from_numpy(label).to(dev)[sp]
        for c in stuff_classes:
            m = point_label == int(c)
            if int(m.sum()) > 100:
                conf.append(1)
                label_id.append(semantic_ind2label[int(c)])
                mask_rows.append(m.to(torch.int64).unsqueeze(0))
    if not mask_rows:
        return np.array([]), np.array([]), np.array([])
    return np.array(conf), np.array(label_id), torch.cat(mask_rows, 0).cpu().numpy()


def superpoint_majority_label(point_pred, superpoint, n_class, device="cuda"):
    """Middle-level semantic prediction of test_scannetv2.py:216-224: every point gets the most frequent point-level
    class of its superpoint (``scipy.stats.mode`` semantics: the SMALLEST class among ties).  The reference loops over
    the superpoints with one ``np.where`` each (O(S*N)); here it is one [S, n_class] histogram (index_add) + argmax
    + gather on the device.  Returns (per-point labels int64 [N], per-superpoint labels int64 [S])."""
    dev = torch.device(device)
    if dev.type != "cuda":
        raise _n.WsisError("superpoint_majority_label runs on the MI355X (there is no CPU fallback)")
    pred = torch.as_tensor(point_pred).to(dev).long()
    sp = torch.as_tensor(superpoint).to(dev).long()
    S = int(sp.max().item()) + 1 if sp.numel() else 0
    hist = torch.zeros(S * n_class, dtype=torch.int32, device=dev)
    hist.index_add_(0, sp * n_class + pred, torch.ones_like(pred, dtype=torch.int32))
    hist = hist.view(S, n_class)
    # argmax with ties -> smallest class: torch.argmax may return any maximal index, so compare against the row max
    first = (hist == hist.max(1, keepdim=True)[0]).to(torch.int32).argmax(1) if S else hist.new_zeros(0).long()
    return first[sp], first


def broadcast_superpoint_label(sp_label, superpoint, device="cuda"):
    """test_scannetv2.py:236-240: point_level_pred[superpoint == spID] = sp_label[spID] for every spID, as one gather"""
    dev = torch.device(device)
    return torch.as_tensor(sp_label).to(dev)[torch.as_tensor(superpoint).to(dev).long()]
