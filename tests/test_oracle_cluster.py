"""CPU: the oracle restatement of clustering_in_graph (oracle/cluster_ref.py) against the outputs of the
reference's own function (tests/golden/cluster_golden.npz, made by tests/golden/make_cluster_golden.py), and the
host BFS operator of the drop-in against the oracle's groups."""
import os

import numpy as np
import pytest

from oracle import cluster_ref

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cluster_golden.npz"))


def _case(tag):
    S = len(G[f"{tag}_sem"])
    return dict(xyz=G[f"{tag}_xyz"], sp=G[f"{tag}_superpoint"].astype(np.int64), S=S, edges=G[f"{tag}_edges"],
                lists=cluster_ref.neighbour_lists(G[f"{tag}_edges"], S), sem=G[f"{tag}_sem"].astype(np.int64),
                off=G[f"{tag}_off"], occ=G[f"{tag}_occ"], size=G[f"{tag}_size"], conf=G[f"{tag}_conf"],
                label_id=G[f"{tag}_label_id"], inst=G[f"{tag}_inst"])


def _inst(masks, N):
    inst = np.full(N, -1, dtype=np.int64)
    for i, m in enumerate(masks):
        inst[np.asarray(m).astype(bool)] = i
    return inst


@pytest.mark.parametrize("tag", ["a", "b"])
def test_oracle_matches_the_reference_function(tag):
    c = _case(tag)
    conf, label_id, masks = cluster_ref.clustering_in_graph(c["xyz"], c["sp"], c["lists"], c["sem"], c["off"],
                                                            c["occ"], c["size"])
    assert np.array_equal(label_id, c["label_id"])
    assert np.array_equal(_inst(masks, len(c["xyz"])), c["inst"]), "instance masks must be identical"
    assert np.allclose(conf, c["conf"], rtol=1e-6, atol=0)     # float32 means over Python-set member order
    assert len(conf) >= 4


@pytest.mark.parametrize("tag", ["a", "b"])
def test_host_graph_bfs_groups_match_the_oracle_walk(tag):
    import inference
    c = _case(tag)
    centre = np.stack([c["xyz"][c["sp"] == s].mean(0) for s in range(c["S"])]) + c["off"]
    valid = np.isin(cluster_ref.SEMANTIC_IND2LABEL, cluster_ref.INSTANCE_VALID)
    off, adj = inference.adjacency_csr((c["edges"][:, 0], c["edges"][:, 1]), c["S"])
    # the edge list holds both directions: mode='all' lists every neighbour twice, like igraph would
    assert all(sorted(set(adj[off[s]:off[s + 1]])) == list(c["lists"][s][::2]) for s in range(c["S"]))
    group, n = inference.graph_bfs(c["sem"], valid, centre.astype(np.float32), c["size"][:, 0], off, adj)
    # oracle walk (groups in seed order)
    visited = np.zeros(c["S"], bool)
    expect = np.full(c["S"], -1)
    g = 0
    for seed in range(c["S"]):
        if not valid[c["sem"][seed]] or visited[seed]:
            continue
        visited[seed] = True
        queue = [seed]
        expect[seed] = g
        while queue:
            cur = queue.pop(0)
            for nb in c["lists"][cur]:
                if c["sem"][nb] == c["sem"][seed] and not visited[nb]:
                    d = np.linalg.norm(centre[cur].astype(np.float32) - centre[nb].astype(np.float32))
                    if d < 0.25 * c["size"][seed]:
                        visited[nb] = True
                        expect[nb] = g
                        queue.append(nb)
        g += 1
    assert n == g and np.array_equal(group, expect)


def test_host_graph_bfs_edge_cases():
    import inference
    valid = np.ones(20, bool)
    # empty graph
    g, n = inference.graph_bfs(np.zeros(0, np.int64), valid, np.zeros((0, 3), np.float32), np.zeros(0, np.float32),
                               np.zeros(1, np.int32), np.zeros(0, np.int32))
    assert n == 0 and len(g) == 0
    # isolated superpoints: every valid one is its own group, invalid classes stay -1
    valid2 = valid.copy()
    valid2[:2] = False
    lab = np.array([0, 5, 5, 1, 7])
    g, n = inference.graph_bfs(lab, valid2, np.zeros((5, 3), np.float32), np.ones(5, np.float32),
                               np.zeros(6, np.int32), np.zeros(0, np.int32))
    assert n == 3 and list(g) == [-1, 0, 1, -1, 2]
    # a chain 0-1-2 of one class: distance threshold of the SEED decides (0.25 * size[seed])
    centre = np.array([[0, 0, 0], [0.2, 0, 0], [0.4, 0, 0]], np.float32)
    off, adj = inference.adjacency_csr((np.array([0, 1]), np.array([1, 2])), 3)
    g, n = inference.graph_bfs(np.array([3, 3, 3]), valid, centre, np.array([1.0, 0.1, 0.1], np.float32), off, adj)
    assert n == 1 and list(g) == [0, 0, 0]          # seed 0: thr 0.25 > 0.2 -> whole chain
    g, n = inference.graph_bfs(np.array([3, 3, 3]), valid, centre, np.array([0.1, 1.0, 0.1], np.float32), off, adj)
    assert n == 2 and list(g) == [0, 1, 1]          # seed 0: thr 0.025 -> alone; seed 1: thr 0.25 -> takes 2
