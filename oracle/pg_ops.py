"""Oracle for pointgroup_ops [UPSTREAM dvlab-research/PointGroup lib/pointgroup_ops, master, unpinned;
semantics restated in SURVEY.md App. A.2].  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
parity unpinned: the upstream source is not in /root/reference."""
import numpy as np


def voxelization_idx(coords, batchsize=None, mode=4):
    """coords int64 [N,4] -> (voxel_locs int64 [M,4], p2v int32 [N], v2p int32 [M,1+maxActive]).
    First-occurrence voxel ids, ascending point lists (call sites: scannetv2_dataset.py:449)."""
    coords = np.asarray(coords, dtype=np.int64)
    N = coords.shape[0]
    table = {}
    p2v = np.empty(N, dtype=np.int32)
    lists = []
    for p in range(N):
        key = (int(coords[p, 0]), int(coords[p, 1]), int(coords[p, 2]), int(coords[p, 3]))
        v = table.get(key)
        if v is None:
            v = len(lists)
            table[key] = v
            lists.append([])
        lists[v].append(p)
        p2v[p] = v
    M = len(lists)
    max_active = max((len(l) for l in lists), default=0)
    v2p = np.zeros((M, 1 + max_active), dtype=np.int32)
    locs = np.zeros((M, 4), dtype=np.int64)
    for v, l in enumerate(lists):
        v2p[v, 0] = len(l)
        v2p[v, 1:1 + len(l)] = l
        locs[v] = coords[l[0]]
    return locs, p2v, v2p


def voxelization(feats, v2p, mode=4):
    """out[m] = sum_i w*feats[v2p[m,1+i]], w=1/n (mode 4), sequential fp32 accumulation in list order
    (train_scannetv2.py:189)."""
    feats = np.asarray(feats, dtype=np.float32)
    M = v2p.shape[0]
    out = np.zeros((M, feats.shape[1]), dtype=np.float32)
    max_active = v2p.shape[1] - 1
    n = v2p[:, 0]
    w = np.where((mode == 4) & (n > 0), np.float32(1.0) / np.maximum(n, 1).astype(np.float32), np.float32(1.0))
    w = w.astype(np.float32)
    for i in range(max_active):
        live = n > i
        out[live] += (w[live, None] * feats[v2p[live, 1 + i]]).astype(np.float32)
    return out


def voxelization_backward(dout, v2p, N, mode=4):
    dout = np.asarray(dout, dtype=np.float32)
    d = np.zeros((N, dout.shape[1]), dtype=np.float32)
    n = v2p[:, 0]
    w = np.where((mode == 4) & (n > 0), np.float32(1.0) / np.maximum(n, 1).astype(np.float32), np.float32(1.0))
    w = w.astype(np.float32)
    for i in range(v2p.shape[1] - 1):
        live = n > i
        d[v2p[live, 1 + i]] += (w[live, None] * dout[live]).astype(np.float32)
    return d


def ballquery_batch_p(coords, batch_idxs, batch_offsets, radius, mean_active=None):
    """per point: ascending same-batch k with |x_p-x_k|^2 < r^2 (fp32, strict), capped at 1000;
    start = exclusive prefix sum of counts.  -> (idx int32 [nActive], start_len int32 [N,2])"""
    coords = np.asarray(coords, dtype=np.float32)
    N = coords.shape[0]
    r2 = np.float32(radius) * np.float32(radius)
    idx_parts = []
    start_len = np.zeros((N, 2), dtype=np.int32)
    cursor = 0
    for p in range(N):
        b = int(batch_idxs[p])
        lo, hi = int(batch_offsets[b]), int(batch_offsets[b + 1])
        d = coords[p][None, :] - coords[lo:hi]
        d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]).astype(np.float32) + (d[:, 2] * d[:, 2]).astype(np.float32)
        hit = np.nonzero(d2 < r2)[0][:1000] + lo
        idx_parts.append(hit.astype(np.int32))
        start_len[p] = (cursor, hit.size)
        cursor += hit.size
    idx = np.concatenate(idx_parts) if idx_parts else np.zeros(0, np.int32)
    return idx.astype(np.int32), start_len


def bfs_cluster(semantic_label, ball_query_idxs, start_len, threshold):
    """FIFO BFS connected components -> (cluster_idxs int32 [sumN,2], offsets int32 [nC+1])"""
    N = len(semantic_label)
    visited = np.zeros(N, dtype=bool)
    idxs = []
    offsets = [0]
    nc = 0
    for i in range(N):
        if visited[i]:
            continue
        queue = [i]
        visited[i] = True
        head = 0
        lab = semantic_label[i]
        while head < len(queue):
            cur = queue[head]
            head += 1
            s, l = int(start_len[cur, 0]), int(start_len[cur, 1])
            for nb in ball_query_idxs[s:s + l]:
                nb = int(nb)
                if visited[nb] or semantic_label[nb] != lab:
                    continue
                visited[nb] = True
                queue.append(nb)
        if len(queue) >= threshold:
            idxs.extend((nc, q) for q in queue)
            offsets.append(offsets[-1] + len(queue))
            nc += 1
    return (np.asarray(idxs, dtype=np.int32).reshape(-1, 2), np.asarray(offsets, dtype=np.int32))
