"""``_spg.dat`` ingestion without igraph (the reference reads them with ``igraph.Graph.Read_Pickle``,
modules/datasets/scannetv2_dataset.py:79).  igraph is not in this image, so the file of this test is written by a
throw-away class with python-igraph's ``Graph.__reduce__`` layout -- ``(cls, (vcount, edgelist, directed, graph_attrs,
vertex_attrs, edge_attrs), __dict__)`` -- registered as ``igraph.Graph`` for the duration of the dump: the stream then
names the global ``igraph Graph`` exactly like a file written by ``graph.write_pickle`` (prepare_data_inst_ScanNetV2.py:88)."""
import gzip
import pickle
import sys
import types

import numpy as np
import pytest

import wsis_datasets as ds


def _fake_igraph_pickle(path, n, edges, vattrs, eattrs, compress=False, module="igraph"):
    mod = types.ModuleType(module)

    class Graph(object):
        def __init__(self, n, edges, directed, gattrs, vattrs, eattrs):
            self._args = (n, edges, directed, gattrs, vattrs, eattrs)

        def __reduce__(self):
            return (self.__class__, self._args, {})

    Graph.__module__, Graph.__qualname__ = module, "Graph"
    mod.Graph = Graph
    saved = {k: sys.modules.get(k) for k in (module, module.split(".")[0])}
    sys.modules[module] = mod
    if "." in module:
        top = types.ModuleType(module.split(".")[0])
        setattr(top, module.split(".")[1], mod)
        sys.modules[module.split(".")[0]] = top
    try:
        data = pickle.dumps(Graph(n, edges, True, {}, vattrs, eattrs), protocol=pickle.HIGHEST_PROTOCOL)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    with (gzip.open if compress else open)(path, "wb") as fh:
        fh.write(data)
    assert b"igraph" in data and "igraph" not in sys.modules


@pytest.mark.parametrize("compress,module", [(False, "igraph"), (True, "igraph"), (False, "igraph.graph")])
def test_spg_pickle_reads_without_igraph(tmp_path, compress, module):
    rng = np.random.default_rng(5)
    n = 23
    und = sorted({(min(a, b), max(a, b)) for a, b in rng.integers(0, n, (60, 2)) if a != b})
    edges = sorted(und + [(b, a) for a, b in und])                       # both directions, sorted tuples (:231)
    f = rng.standard_normal((len(edges), 13))
    is1 = rng.integers(-1, 2, len(edges)).tolist()
    vattrs = {"v": list(range(n)), "semantic_label": rng.integers(0, 20, n).tolist(),
              "instance_label": rng.integers(-100, 5, n).tolist(),
              "superpoint_feature": [row for row in rng.standard_normal((n, 6))],
              "superpoint_offset_vector": rng.standard_normal((n, 3)).astype(np.float32)}
    path = str(tmp_path / "scene0000_00_spg.dat")
    _fake_igraph_pickle(path, n, edges, vattrs, {"f": [row for row in f], "is1ins": is1}, compress, module)
    g = ds.read_spg_pickle(path)
    assert isinstance(g, ds.PlainGraph) and g.vcount == n
    assert np.array_equal(g.edges, np.asarray(edges))
    assert np.allclose(g.f, f.astype(np.float32)) and np.array_equal(g.is1ins, np.asarray(is1))
    for k, v in vattrs.items():
        assert np.array_equal(g.vs[k], np.asarray(v)), k
    # and it is the graph type the per-scene transform works on
    sub = g.subgraph(np.arange(0, n, 2))
    assert sub.vcount == (n + 1) // 2 and sub.edges.max() < sub.vcount


def test_spg_pickle_refuses_other_globals(tmp_path):
    path = str(tmp_path / "bad_spg.dat")
    with open(path, "wb") as fh:
        pickle.dump(types.SimpleNamespace(a=1), fh)
    with pytest.raises(pickle.UnpicklingError):
        ds.read_spg_pickle(path)


class _Evil(object):
    """a pickle that would run code through REDUCE if a whole top-level module were allow-listed"""

    def __init__(self, fn, args):
        self.fn, self.args = fn, args

    def __reduce__(self):
        return (self.fn, self.args)


@pytest.mark.parametrize("payload", [
    _Evil(eval, ("__import__('os').getpid()",)),
    _Evil(getattr, (int, "__add__")),
    _Evil(__import__, ("os",)),
    _Evil(np.testing.assert_equal, (1, 1)),
])
def test_spg_pickle_refuses_callables_of_allowed_modules(tmp_path, payload):
    """builtins.eval / getattr / __import__ and arbitrary numpy callables must not resolve"""
    path = str(tmp_path / "evil_spg.dat")
    with open(path, "wb") as fh:
        pickle.dump(payload, fh)
    with pytest.raises(pickle.UnpicklingError):
        ds.read_spg_pickle(path)
