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


# ---- the reference's torch files: per-scene .pth and checkpoints (round 6: no loader of this package unpickles) ----------
class _Payload(object):
    def __init__(self, marker):
        self.marker = str(marker)

    def __reduce__(self):
        import os
        return (os.system, ("echo pwned > " + self.marker,))


def _rename_numpy_module(src, dst):
    """rewrite a torch.save archive so that its pickle names the array reconstructor the way numpy 1.x did
    (``numpy.core.multiarray``): what the reference's own files, written in 2021, contain."""
    import zipfile
    with zipfile.ZipFile(src) as zi, zipfile.ZipFile(dst, "w") as zo:
        for it in zi.infolist():
            data = zi.read(it.filename)
            if it.filename.endswith("data.pkl"):
                assert b"numpy._core.multiarray" in data
                data = data.replace(b"cnumpy._core.multiarray\n", b"cnumpy.core.multiarray\n")
            zo.writestr(it, data)


def test_scene_file_loader_executes_nothing(tmp_path):
    """``load_scene_file`` (scannetv2_dataset.py:62-73) reads numpy tuples through ``weights_only=True`` + an allow-list;
    a file whose pickle carries a REDUCE of ``os.system`` is refused and the command never runs."""
    import pickle
    import torch
    rng = np.random.default_rng(0)
    tup = (rng.random((7, 3), dtype=np.float32), rng.random((7, 3)).astype(np.float32), np.arange(7.0),
           np.arange(7, dtype=np.int32), np.arange(7, dtype=np.int64), "scene0000_00")
    torch.save(tup, tmp_path / "ok.pth")
    back = ds.load_scene_file(tmp_path / "ok.pth")
    assert back[5] == "scene0000_00" and all(np.array_equal(a, b) for a, b in zip(back[:5], tup[:5]))
    if np.lib.NumpyVersion(np.__version__) >= "2.0.0":
        _rename_numpy_module(tmp_path / "ok.pth", tmp_path / "ok_numpy1.pth")
        back = ds.load_scene_file(tmp_path / "ok_numpy1.pth")
        assert all(np.array_equal(a, b) for a, b in zip(back[:5], tup[:5]))
    marker = tmp_path / "pwned_scene"
    torch.save(tup[:4] + (_Payload(marker), "x"), tmp_path / "evil.pth")
    with pytest.raises(pickle.UnpicklingError):
        ds.load_scene_file(tmp_path / "evil.pth")
    assert not marker.exists()
    torch.save(tup[:4] + (np.array([_Payload(marker)], dtype=object), "x"), tmp_path / "evil_obj.pth")
    with pytest.raises((pickle.UnpicklingError, ValueError)):
        ds.load_scene_file(tmp_path / "evil_obj.pth")
    assert not marker.exists()


def test_checkpoint_loader_executes_nothing(tmp_path):
    """``harness.load_checkpoint`` (utils/checkpoint.py:105-135): same loader; a checkpoint with a code-carrying entry
    next to a valid state dict is refused before anything is applied to the model."""
    import pickle
    import torch
    import harness
    model = torch.nn.Linear(3, 2)
    before = {k: v.clone() for k, v in model.state_dict().items()}
    other = torch.nn.Linear(3, 2)
    marker = tmp_path / "pwned_ck"
    torch.save({"meta": {"epoch": 3, "hook": _Payload(marker)}, "model": other.state_dict()}, tmp_path / "evil_ck.pth")
    with pytest.raises(pickle.UnpicklingError):
        harness.load_checkpoint(model, tmp_path / "evil_ck.pth")
    assert not marker.exists()
    assert all(torch.equal(before[k], v) for k, v in model.state_dict().items())
    torch.save({"meta": {"epoch": 3, "time": "now", "lr": np.float64(1e-3)}, "model": other.state_dict()},
               tmp_path / "ok_ck.pth")
    ck = harness.load_checkpoint(model, tmp_path / "ok_ck.pth")
    assert ck["meta"]["epoch"] == 3 and float(ck["meta"]["lr"]) == 1e-3
    assert all(torch.equal(a, b) for a, b in zip(other.state_dict().values(), model.state_dict().values()))
