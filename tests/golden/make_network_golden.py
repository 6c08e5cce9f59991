"""network_golden.npz: outputs of the REFERENCE's own model file on a small seeded batch.

Runs, in this container only, ``/root/reference/modules/model/backbone_3D_WSIS.py`` (``Network.__init__`` and
``Network.forward``, with it ``sparse_unet3d.py`` ResidualBlock / UBlock, ``graphnet.py`` GraphNetwork and
``spg_modules.py`` NNConv / RNNGraphConvModule / GRUCellEx) on the CPU.  The third-party packages it imports are absent
(SURVEY 8c), so they are registered as stand-ins BEFORE the import:

  spconv            SparseConvTensor / SparseSequential / SparseModule / SubMConv3d / SparseConv3d / SparseInverseConv3d
                    with the constructor signatures the reference uses; the arithmetic is oracle/spconv_ref.py
                    (hash rulebook, per offset index_select -> mm -> index_add_ : [UPSTREAM spconv v1.0, SURVEY A.1])
  torch_scatter     oracle/scatter_ref.py                                     [UPSTREAM torch_scatter 2.0.x, A.3]
  torch_geometric   nn.conv.MessagePassing as PyG 1.6 defines it: ``__init__(aggr='add', flow='source_to_target',
                    node_dim=0)``; ``propagate`` resolves ``message``'s arguments by name (``x_j`` = x[edge_index[j]],
                    ``edge_index_i``, ``size_i``), reduces the messages over edge_index[i] with ``aggr`` and calls
                    ``update``; (i, j) = (1, 0) for source_to_target.  nn.inits.uniform.                    [A.3]
  pointgroup_ops, igraph, cupy, pynvrtc, treelib, htree, cluster, utils       import-only stubs (nothing is called)

What this pins is the GLUE that IS in the reference tree: the module wiring, the residual / skip / concat order, which
BatchNorm sits where, the argument order of NNConv.message, the direction of the message passing (NNConv never hands
its ``flow`` argument on, so the PyG default runs), GRUCellEx, the head layout and the affinity block.  The expected
outputs then check oracle/network_ref.py (CPU) and the HIP path (-m gpu).  Weights: tests/util.seeded_state_dict (drawn
per NAME, so every implementation loads the same values).  Only arrays are written -- no reference source text.

    python tests/golden/make_network_golden.py          (needs /root/reference; never runs on the GPU box)
"""
import importlib
import inspect
import os
import sys
import types
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


# ---------------------------------------------------------------- phase A: a small host batch from this build's harness
def make_batch():
    importlib.import_module("3d-wsis_amd")
    import harness
    from oracle import pg_ops, scatter_ref
    scenes = [harness.make_scene(41, room=(0.62, 0.5, 0.44), n_box=0, graph="mesh", sp_cell=0.11),
              harness.make_scene(42, room=(0.5, 0.56, 0.4), n_box=0, graph="mesh", sp_cell=0.11)]
    b = harness.collate(scenes)
    feats = torch.cat((b["feats"], b["locs_float"]), 1)
    voxel_feats = pg_ops.voxelization(feats.numpy(), b["v2p_map"].numpy(), 4)
    centre = scatter_ref.scatter(b["locs_float"], b["superpoint"], 0, None, "mean")
    gi = b["GIs"][0]
    return dict(voxel_feats=voxel_feats.astype(np.float32), voxel_locs=b["voxel_locs"].numpy().astype(np.int32),
                spatial_shape=np.asarray(b["spatial_shape"], dtype=np.int64), batch_size=np.int64(len(scenes)),
                p2v_map=b["p2v_map"].numpy(), superpoint=b["superpoint"].numpy(), centre=centre.numpy(),
                edge_indexes=gi._edge_indexes.numpy(), edgefeats=gi._edgefeats.numpy(),
                edge_u=b["edge_u_list"].numpy(), edge_v=b["edge_v_list"].numpy())


# ---------------------------------------------------------------- phase B: stand-ins for the absent packages
def register_standins():
    from oracle import scatter_ref
    from oracle import spconv_ref as sref
    for name in list(sys.modules):      # this build's own drop-in packages must not shadow the stand-ins
        if name.split(".")[0] in ("spconv", "pointgroup_ops", "torch_scatter", "graphnet", "backbone_3D_WSIS",
                                  "sparse_unet3d", "losses_3D_WSIS"):
            del sys.modules[name]
    sys.path[:] = [p for p in sys.path if "3d-wsis_amd" not in p]

    # ---- spconv [UPSTREAM A.1]
    class SparseConvTensor(object):
        def __init__(self, features, indices, spatial_shape, batch_size, grid=None):
            self.features, self.indices = features, indices
            self.spatial_shape, self.batch_size = spatial_shape, batch_size
            self.indice_dict, self.grid = {}, grid

        def find_indice_pair(self, key):
            return self.indice_dict.get(key) if key is not None else None

    class SparseModule(nn.Module):
        pass

    def is_spconv_module(m):
        return isinstance(m, SparseModule)

    class SparseSequential(SparseModule):
        def __init__(self, *args, **kwargs):
            super().__init__()
            if len(args) == 1 and isinstance(args[0], OrderedDict):
                for key, module in args[0].items():
                    self.add_module(key, module)
            else:
                for idx, module in enumerate(args):
                    self.add_module(str(idx), module)
            for name, module in kwargs.items():
                self.add_module(name, module)

        def forward(self, input):
            for k, module in self._modules.items():
                if is_spconv_module(module):            # the tensor goes in, a tensor comes out
                    input = module(input)
                else:                                   # ordinary layers map .features, re-bound on the same object
                    if isinstance(input, SparseConvTensor):
                        if input.indices.shape[0] != 0:
                            input.features = module(input.features)
                    else:
                        input = module(input)
            return input

    class _Conv(SparseModule):
        def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=0, dilation=1, groups=1, bias=True,
                     indice_key=None, subm=False, inverse=False):
            super().__init__()
            self.in_channels, self.out_channels = in_channels, out_channels
            self.kernel_size, self.stride, self.padding = sref._triple(kernel_size), sref._triple(stride), sref._triple(padding)
            self.subm, self.inverse, self.indice_key = subm, inverse, indice_key
            self.weight = nn.Parameter(torch.zeros(*self.kernel_size, in_channels, out_channels))
            if bias:
                self.bias = nn.Parameter(torch.zeros(out_channels))
            else:
                self.register_parameter("bias", None)

        def forward(self, input):
            feats, idx = input.features, np.asarray(input.indices)
            shape = [int(v) for v in input.spatial_shape]
            out_t = SparseConvTensor(None, input.indices, input.spatial_shape, input.batch_size)
            out_t.indice_dict = input.indice_dict            # the rulebook cache travels with the tensor
            if int(np.prod(self.kernel_size)) == 1 and self.subm:
                out = feats @ self.weight.view(self.in_channels, self.out_channels)
            elif self.subm:
                cached = input.find_indice_pair(self.indice_key)
                if cached is None:
                    cached = sref.subm_pairs_fast(idx, shape, self.kernel_size, self.padding)
                    if self.indice_key is not None:
                        input.indice_dict[self.indice_key] = cached
                out = sref.pairs_conv(feats, self.weight, cached, feats.shape[0])
            elif not self.inverse:
                out_idx, out_shape, pairs = sref.down_pairs_fast(idx, shape, self.kernel_size, self.stride, self.padding)
                input.indice_dict[self.indice_key] = (idx, shape, pairs)
                out = sref.pairs_conv(feats, self.weight, pairs, out_idx.shape[0])
                out_t.indices, out_t.spatial_shape = torch.from_numpy(out_idx.astype(np.int32)), out_shape
            else:
                in_idx, in_shape, pairs = input.find_indice_pair(self.indice_key)
                out = sref.pairs_conv(feats, self.weight, sref.inverse_pairs(pairs), in_idx.shape[0])
                out_t.indices, out_t.spatial_shape = torch.from_numpy(np.asarray(in_idx).astype(np.int32)), in_shape
            if self.bias is not None:
                out = out + self.bias
            out_t.features = out
            return out_t

    class SubMConv3d(_Conv):
        def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                     indice_key=None):
            super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias, indice_key,
                             subm=True)

    class SparseConv3d(_Conv):
        def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                     indice_key=None):
            super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias, indice_key)

    class SparseInverseConv3d(_Conv):
        def __init__(self, in_channels, out_channels, kernel_size, indice_key, bias=True):
            super().__init__(in_channels, out_channels, kernel_size, bias=bias, indice_key=indice_key, inverse=True)

    sp = _stub("spconv", SparseConvTensor=SparseConvTensor, SparseSequential=SparseSequential, SparseModule=SparseModule,
               SubMConv3d=SubMConv3d, SparseConv3d=SparseConv3d, SparseInverseConv3d=SparseInverseConv3d)
    sp.modules = _stub("spconv.modules", SparseModule=SparseModule, SparseSequential=SparseSequential)

    # ---- torch_scatter [UPSTREAM A.3]
    def scatter(src, index, dim=-1, out=None, dim_size=None, reduce="sum"):
        return scatter_ref.scatter(src, index, 0 if dim in (0, -src.dim()) else dim, dim_size, reduce)

    _stub("torch_scatter", scatter=scatter,
          scatter_mean=lambda s, i, dim=-1, out=None, dim_size=None: scatter(s, i, dim, None, dim_size, "mean"),
          scatter_add=lambda s, i, dim=-1, out=None, dim_size=None: scatter(s, i, dim, None, dim_size, "sum"),
          scatter_min=lambda s, i, dim=-1, out=None, dim_size=None: (scatter(s, i, dim, None, dim_size, "min"), None),
          scatter_max=lambda s, i, dim=-1, out=None, dim_size=None: (scatter(s, i, dim, None, dim_size, "max"), None))

    # ---- torch_geometric.nn.conv.MessagePassing [UPSTREAM PyG 1.6, A.3]
    class MessagePassing(nn.Module):
        def __init__(self, aggr="add", flow="source_to_target", node_dim=0):
            super().__init__()
            assert aggr in ("add", "mean", "max") and flow in ("source_to_target", "target_to_source")
            self.aggr, self.flow, self.node_dim = aggr, flow, node_dim

        def propagate(self, edge_index, size=None, **kwargs):
            i, j = (1, 0) if self.flow == "source_to_target" else (0, 1)
            n = None
            for v in kwargs.values():
                if torch.is_tensor(v) and n is None and v.dim() >= 1:
                    n = v.size(self.node_dim)
            sizes = [n, n] if size is None else list(size)
            args = {}
            for name in list(inspect.signature(self.message).parameters):
                if name.endswith("_i") or name.endswith("_j"):
                    base, which = name[:-2], (i if name.endswith("_i") else j)
                    if base == "edge_index":
                        args[name] = edge_index[which]
                    elif base == "size":
                        args[name] = sizes[which]
                    else:
                        args[name] = kwargs[base].index_select(self.node_dim, edge_index[which])
                else:
                    args[name] = kwargs[name]
            msg = self.message(**args)
            out = scatter_ref.scatter(msg, edge_index[i], 0, sizes[i], {"add": "sum"}.get(self.aggr, self.aggr))
            upd = {k: kwargs[k] for k in list(inspect.signature(self.update).parameters)[1:]}
            return self.update(out, **upd)

        def message(self, x_j):
            return x_j

        def update(self, aggr_out):
            return aggr_out

    def uniform(size, tensor):
        if tensor is not None:
            bound = 1.0 / np.sqrt(size)
            tensor.data.uniform_(-bound, bound)

    tg = _stub("torch_geometric")
    tg.nn = _stub("torch_geometric.nn")
    tg.nn.conv = _stub("torch_geometric.nn.conv", MessagePassing=MessagePassing)
    tg.nn.inits = _stub("torch_geometric.nn.inits", uniform=uniform)

    # ---- import-only stubs
    _stub("pointgroup_ops")
    _stub("igraph")
    cp = _stub("cupy")
    cp.cuda = _stub("cupy.cuda")
    pn = _stub("pynvrtc")
    pn.compiler = _stub("pynvrtc.compiler", Program=object)
    _stub("treelib", Tree=object)
    _stub("htree")
    cl = _stub("cluster")
    cl.hierarchy = _stub("cluster.hierarchy", linkage=None)
    _stub("utils")


class _GI(object):
    """what Network.forward needs from ecc.GraphConvInfo (ecc/GraphConvInfo.py:75-87)"""

    def __init__(self, edge_indexes, edgefeats):
        self._edge_indexes, self._edgefeats = edge_indexes, edgefeats

    def cuda(self):
        pass

    def get_buffers(self):
        return None, None, None, None, self._edgefeats

    def get_pyg_buffers(self):
        return self._edge_indexes


def main():
    batch = make_batch()
    register_standins()
    sys.path.insert(0, os.path.join(REF, "modules", "model"))
    import backbone_3D_WSIS as ref_model
    import spconv
    assert ref_model.__file__.startswith(REF)
    from tests.util import seeded_state_dict
    param = types.SimpleNamespace(input_channel=3, use_coords=True, blocks=5, block_reps=2, media=32, classes=20,
                                  fix_module="[]")
    torch.manual_seed(0)
    net = ref_model.Network(param)
    sd = net.state_dict()
    net.load_state_dict(seeded_state_dict({k: v.shape for k, v in sd.items()}), strict=True)
    out = {"in_" + k: v for k, v in batch.items()}
    for mode in ("train", "eval"):
        net.train(mode == "train")
        inp = spconv.SparseConvTensor(torch.from_numpy(batch["voxel_feats"]), torch.from_numpy(batch["voxel_locs"]),
                                      batch["spatial_shape"], int(batch["batch_size"]))
        extra = {"superpoint": torch.from_numpy(batch["superpoint"]),
                 "GIs": [_GI(torch.from_numpy(batch["edge_indexes"]), torch.from_numpy(batch["edgefeats"]))],
                 "superpoint_cenetr_xyz": torch.from_numpy(batch["centre"]),
                 "edge_u_list": torch.from_numpy(batch["edge_u"]), "edge_v_list": torch.from_numpy(batch["edge_v"])}
        with torch.no_grad():
            ret = net(inp, torch.from_numpy(batch["p2v_map"]), extra)
        for k, v in ret.items():
            v = v.numpy()
            if k == "semantic_scores":      # [N, 20]: every 4th point row and the column sums of all rows
                out[f"{mode}_{k}_colsum"] = v.astype(np.float64).sum(0)
                v = v[::4]
            out[f"{mode}_{k}"] = v
    # the training pass moved the running statistics: part of the contract (momentum 0.1, unbiased variance)
    for k, v in net.state_dict().items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            out["stat_" + k] = v.numpy()
    out["state_names"] = np.array(sorted(sd.keys()))
    np.savez_compressed(os.path.join(HERE, "network_golden.npz"), **out)
    print("network_golden.npz:", {k: tuple(v.shape) for k, v in out.items() if k.startswith("train_")},
          "voxels", batch["voxel_feats"].shape, "superpoints", int(batch["superpoint"].max()) + 1,
          "edges", batch["edge_u"].shape[0])


if __name__ == "__main__":
    main()
