"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's test-time grouping ``clustering_in_graph``
(/root/reference/test_scannetv2.py:281-455), in the reference's own shape: a boolean point mask per superpoint,
deque BFS, one unique-voxel count per group, sequential fragment absorption.  Pinned against the reference function
itself: tests/golden/cluster_golden.npz holds inputs and the outputs of the real function run in this container
(tests/golden/make_cluster_golden.py), tests/test_oracle_cluster.py compares.

``neighbours``: list of neighbour id arrays per superpoint (igraph ``neighbors(mode='all')``).
"""
import collections
from math import sqrt

import numpy as np

SEMANTIC_IND2LABEL = np.array([1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 14, 16, 24, 28, 33, 34, 36, 39])   # :288
INSTANCE_VALID = np.array([3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 14, 16, 24, 28, 33, 34, 36, 39])             # :289


def neighbour_lists(edges, S):
    """mode='all' neighbour lists (ascending, as igraph returns them) from a directed edge array [E,2]"""
    out = [[] for _ in range(S)]
    for u, v in np.asarray(edges):
        out[int(u)].append(int(v))
        out[int(v)].append(int(u))
    return [np.array(sorted(l), dtype=np.int64) for l in out]


def clustering_in_graph(xyz_origin, superpoint, neighbours, sp_sem_pred, pred_offsets, pred_occupancy, pred_size):
    N, S = len(xyz_origin), len(sp_sem_pred)
    masks = [superpoint == s for s in range(S)]                                            # :297
    centre = np.stack([xyz_origin[m].mean(0) for m in masks]) + pred_offsets               # :301-302
    visited = np.zeros(S, dtype=bool)

    def bfs(seed):                                                                          # :312-340
        visited[seed] = True
        queue = collections.deque([seed])
        group = [seed]
        while queue:
            cur = queue.popleft()
            for nb in neighbours[cur]:
                if sp_sem_pred[nb] == sp_sem_pred[seed] and not visited[nb]:
                    d = np.linalg.norm(centre[cur] - centre[nb], ord=2)
                    if d < 0.25 * pred_size[seed]:
                        group.append(int(nb))
                        visited[nb] = True
                        queue.append(nb)
        return sorted(group)

    def occupancy_of(g):                                                                    # :345-349
        return np.exp(pred_occupancy[np.array(g)]).mean()

    def centre_of(g):                                                                       # :352-360
        c, n = np.zeros(3), 0
        for s in g:
            c += centre[s] * masks[s].sum()
            n += masks[s].sum()
        return c / n

    def size_of(g):                                                                         # :362-364
        return np.mean(pred_size[np.array(g)])

    primaries, fragments = [], []
    for seed in range(S):                                                                   # :368-408
        if SEMANTIC_IND2LABEL[sp_sem_pred[seed]] not in INSTANCE_VALID or visited[seed]:
            continue
        g = bfs(seed)
        mask = np.zeros(N, dtype=bool)
        for s in g:
            mask |= masks[s]
        occ = occupancy_of(g)
        vox = (xyz_origin[mask] * 50).astype(np.float32)
        n_vox = len(np.unique(np.trunc(vox).astype(np.int64), axis=0))                      # :381-385
        n = mask.sum()
        if n_vox < 0.3 * occ:
            fragments.append(dict(mask=mask, cls=sp_sem_pred[seed], centre=centre_of(g), group=g, n=n))
        else:
            r = max(0.01 * sqrt(n), 0.02 * sqrt(occ), size_of(g))
            primaries.append(dict(mask=mask, cls=sp_sem_pred[seed], centre=centre_of(g), r=r, group=g, n=n))

    for f in fragments:                                                                     # :410-438
        index, dmin = -1, float("inf")
        for i, p in enumerate(primaries):
            d = np.linalg.norm(f["centre"] - p["centre"], ord=2)
            if f["cls"] == p["cls"] and d < dmin:
                index, dmin = i, d
        if not primaries:
            break
        p = primaries[index]
        if dmin < p["r"]:
            both = f["group"] + p["group"]
            m = f["mask"] | p["mask"]
            p["r"] = max(0.02 * sqrt(occupancy_of(both)), 0.01 * sqrt(m.sum()), p["r"], size_of(both))
            p["centre"] = centre_of(both)
            p["mask"], p["n"] = m, m.sum()
            p["group"] = p["group"] + f["group"]

    conf = [min(p["n"] / occupancy_of(p["group"]), 1) for p in primaries]                   # :441-451
    label_id = [SEMANTIC_IND2LABEL[p["cls"]] for p in primaries]
    ins = [p["mask"].astype(int) for p in primaries]
    return np.array(conf), np.array(label_id), np.array(ins)
