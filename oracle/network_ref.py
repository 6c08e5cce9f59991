"""Oracle restatement of the whole per-scene forward of 3D-WSIS on torch-CPU.  TEST INFRASTRUCTURE ONLY
(tests/, bench.py cpu_baseline leg, smoke) -- see oracle/__init__.py.

Follows modules/model/backbone_3D_WSIS.py:164-255 (Network.forward), modules/model/sparse_unet3d.py:163-172,
321-350 (ResidualBlock / UBlock), modules/model/graphnet.py:98-114 + modules/model/spg_modules.py:61-121,
152-185,226-253 (ECC GNN) with the third-party operators replaced by the oracle restatements
(oracle/spconv_ref.py: per-offset index_select -> mm -> index_add_, oracle/scatter_ref.py, oracle/pg_ops.py).
Module names equal the reference's so a product/reference state_dict loads with strict=True."""
import functools
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import spconv_ref as sref
from .affinity_ref import edge_affinity
from .scatter_ref import scatter


class SparseT(object):
    """features + indices + shape + shared rulebook cache (what spconv.SparseConvTensor carries)"""

    def __init__(self, features, indices, shape, cache=None):
        self.features, self.indices, self.shape = features, indices, [int(s) for s in shape]
        self.cache = {} if cache is None else cache


class RefConv(nn.Module):
    def __init__(self, cin, cout, k, kind, key=None, stride=1, padding=0):
        super().__init__()
        self.k, self.kind, self.key, self.stride, self.padding = sref._triple(k), kind, key, stride, padding
        self.weight = nn.Parameter(torch.zeros(*self.k, cin, cout))

    def forward(self, t):
        M = t.features.shape[0]
        if int(np.prod(self.k)) == 1:
            cin, cout = self.weight.shape[-2:]
            return SparseT(t.features @ self.weight.view(cin, cout), t.indices, t.shape, t.cache)
        if self.kind == "subm":
            ck = ("subm", self.key, M)
            if ck not in t.cache:
                t.cache[ck] = sref.subm_pairs_fast(t.indices, t.shape, self.k, self.padding)
            return SparseT(sref.pairs_conv(t.features, self.weight, t.cache[ck], M), t.indices, t.shape, t.cache)
        if self.kind == "down":
            out_idx, out_shape, pairs = sref.down_pairs_fast(t.indices, t.shape, self.k, self.stride, self.padding)
            t.cache[("down", self.key)] = (t.indices, t.shape, pairs)
            return SparseT(sref.pairs_conv(t.features, self.weight, pairs, out_idx.shape[0]), out_idx, out_shape,
                           t.cache)
        in_idx, in_shape, pairs = t.cache[("down", self.key)]
        return SparseT(sref.pairs_conv(t.features, self.weight, sref.inverse_pairs(pairs), in_idx.shape[0]),
                       in_idx, in_shape, t.cache)


class RefSeq(nn.Module):
    """SparseSequential: conv modules take the tensor, the others map .features (re-bound on the same object)"""

    def __init__(self, *mods):
        super().__init__()
        if len(mods) == 1 and isinstance(mods[0], OrderedDict):
            for k, m in mods[0].items():
                self.add_module(k, m)
        else:
            for i, m in enumerate(mods):
                self.add_module(str(i), m)

    def forward(self, t):
        for m in self._modules.values():
            if isinstance(m, (RefConv, RefSeq, RefResidual)):
                t = m(t)
            elif t.features.shape[0] != 0:
                t.features = m(t.features)
        return t


class RefResidual(nn.Module):
    def __init__(self, cin, cout, norm_fn, key):
        super().__init__()
        self.i_branch = RefSeq(nn.Identity()) if cin == cout else RefSeq(RefConv(cin, cout, 1, "subm"))
        self.conv_branch = RefSeq(norm_fn(cin), nn.ReLU(), RefConv(cin, cout, 3, "subm", key, padding=1),
                                  norm_fn(cout), nn.ReLU(), RefConv(cout, cout, 3, "subm", key, padding=1))

    def forward(self, t):
        identity = SparseT(t.features, t.indices, t.shape)
        out = self.conv_branch(t)
        out.features = out.features + self.i_branch(identity).features
        return out


class RefUBlock(nn.Module):
    def __init__(self, planes, norm_fn, reps, kid):
        super().__init__()
        self.planes = planes
        p0 = planes[0]
        self.blocks = RefSeq(OrderedDict((f"block{i}", RefResidual(p0, p0, norm_fn, f"subm{kid}")) for i in range(reps)))
        if len(planes) > 1:
            self.conv = RefSeq(norm_fn(p0), nn.ReLU(), RefConv(p0, planes[1], 2, "down", f"spconv{kid}", stride=2))
            self.u = RefUBlock(planes[1:], norm_fn, reps, kid + 1)
            self.deconv = RefSeq(norm_fn(planes[1]), nn.ReLU(), RefConv(planes[1], p0, 2, "inverse", f"spconv{kid}"))
            self.blocks_tail = RefSeq(OrderedDict(
                (f"block{i}", RefResidual(p0 * (2 - i), p0, norm_fn, f"subm{kid}")) for i in range(reps)))

    def forward(self, t):
        out = self.blocks(t)
        identity = SparseT(out.features, out.indices, out.shape)
        if len(self.planes) > 1:
            dec = self.deconv(self.u(self.conv(out)))
            out.features = torch.cat((identity.features, dec.features), dim=1)
            out = self.blocks_tail(out)
        return out


class RefGRUCellEx(nn.GRUCell):
    def __init__(self, n):
        super().__init__(n, n, True)
        self.add_module("ig", nn.Linear(n, n, bias=True))
        self.ini = nn.InstanceNorm1d(1, eps=1e-5, affine=False, track_running_stats=False)
        self.inh = nn.InstanceNorm1d(1, eps=1e-5, affine=False, track_running_stats=False)

    def forward(self, input, hidden):           # spg_modules.py:226-253
        input = torch.sigmoid(self.ig(hidden)) * input
        gi = self.ini(F.linear(input, self.weight_ih).unsqueeze(1)).squeeze(1)
        gh = self.inh(F.linear(hidden, self.weight_hh).unsqueeze(1)).squeeze(1)
        i_r, i_i, i_n = gi.chunk(3, 1)
        h_r, h_i, h_n = gh.chunk(3, 1)
        bih_r, bih_i, bih_n = self.bias_ih.chunk(3)
        bhh_r, bhh_i, bhh_n = self.bias_hh.chunk(3)
        resetgate = torch.sigmoid(i_r + bih_r + h_r + bhh_r)
        inputgate = torch.sigmoid(i_i + bih_i + h_i + bhh_i)
        newgate = torch.tanh(i_n + bih_n + resetgate * (h_n + bhh_n))
        return newgate + inputgate * (hidden - newgate)


class RefRNNGraphConv(nn.Module):
    def __init__(self, n, reps):
        super().__init__()
        self._cell = RefGRUCellEx(n)
        self._fnet = nn.Sequential(nn.Linear(13, 32), nn.ReLU(True), nn.Linear(32, 128), nn.ReLU(True),
                                   nn.Linear(128, 64), nn.BatchNorm1d(64), nn.ReLU(True), nn.Linear(64, n * n))
        self.reps = reps

    def forward(self, hx, edge_indexes, edgefeats):
        nc = hx.size(1)
        weights = self._fnet(edgefeats).view(-1, nc, nc)
        # NNConv never forwards its ``flow`` argument to MessagePassing (spg_modules.py:61-72), so PyG's default
        # source_to_target runs: x_j = x[edge_index[0]], aggregated at edge_index[1] -- the sum of the non-PyG branch too
        # (ecc/GraphConvModule.py:49-78); pinned by tests/golden/network_golden.npz
        src, dst = edge_indexes[0], edge_indexes[1]
        hxs = [hx]
        for _ in range(self.reps):
            msg = torch.matmul(hx[src].unsqueeze(1), weights).squeeze(1)      # NNConv.message, vv=False
            inp = scatter(msg, dst, 0, hx.size(0), "mean")                     # aggr='mean' over the in-edges
            hx = self._cell(inp, hx)
            hxs.append(hx)
        return torch.cat(hxs, 1)


class RefECC(nn.Module):
    def __init__(self, n):
        super().__init__()
        self.add_module("0", RefRNNGraphConv(n, 7))
        self.add_module("1", nn.Linear(n * 8, 64))
        self.add_module("2", nn.BatchNorm1d(64, eps=1e-5))

    def forward(self, x, edge_indexes, edgefeats):
        x = self._modules["0"](x, edge_indexes, edgefeats)
        return F.relu(self._modules["2"](self._modules["1"](x)))


class RefNetwork(nn.Module):
    def __init__(self, classes=20, media=32, blocks=5, reps=2, in_ch=6):
        super().__init__()
        norm_fn = functools.partial(nn.BatchNorm1d, eps=1e-4, momentum=0.1)
        self.input_conv = RefSeq(RefConv(in_ch, media, 3, "subm", "subm1", padding=1))
        self.unet = RefUBlock([media * (i + 1) for i in range(blocks)], norm_fn, reps, 1)
        self.output_layer = RefSeq(norm_fn(media), nn.ReLU())

        def head(cin, cout):
            return nn.Sequential(nn.Linear(cin, cin), norm_fn(cin), nn.ReLU(), nn.Linear(cin, cout))

        self.linear = head(media, classes)
        self.ecc = RefECC(media)
        self.sp_sem_seg, self.sp_offset_vector_head = head(64, classes), head(64, 3)
        self.sp_occupancy_head, self.sp_ins_size_head = head(64, 1), head(64, 1)
        self.fc_position = nn.Sequential(nn.Linear(3, 16), nn.ReLU(), nn.Linear(16, 1))
        self.w_qs, self.w_ks, self.w_vs = (nn.Linear(64, 64, bias=False) for _ in range(3))
        self.feature_term = head(64, 7)

    def forward(self, voxel_feats, voxel_indices, spatial_shape, p2v, superpoint, centre, edge_indexes, edgefeats,
                edge_u, edge_v):
        ret = {}
        t = SparseT(voxel_feats, np.asarray(voxel_indices), spatial_shape)
        out = self.output_layer(self.unet(self.input_conv(t)))
        output_feats = out.features[p2v.long()]
        ret["semantic_scores"] = self.linear(output_feats)
        emb = scatter(output_feats, superpoint.long(), 0, None, "mean")
        e = self.ecc(emb, edge_indexes, edgefeats)
        ret["sp_semantic_scores"] = self.sp_sem_seg(e)
        ret["pred_sp_offset_vectors"] = self.sp_offset_vector_head(e)
        ret["pred_sp_occupancy"] = self.sp_occupancy_head(e).squeeze(-1)
        ret["pred_sp_ins_size"] = self.sp_ins_size_head(e).squeeze(-1)
        q, k, v = self.w_qs(e), self.w_ks(e), self.w_vs(e)
        pos_enc = self.fc_position(centre[edge_u] - centre[edge_v]).reshape(-1)
        aff, res = edge_affinity(q, k, v, pos_enc, edge_u, edge_v)
        ret["edge_affinity"] = aff
        sp_feat = torch.zeros_like(e) + e
        sp_feat = torch.cat((sp_feat[:res.shape[0]] + res, sp_feat[res.shape[0]:]), 0)
        ret["sp_discriminative_feats"] = self.feature_term(sp_feat)
        return ret


def forward_loss_cpu(ref_model, criterion, batch, mode=4, epoch=5, dtype=None):
    """the iteration of train_scannetv2.py:174-232 entirely on the oracle (host batch dict from harness.collate).
    ``dtype=torch.float64`` (with ``ref_model.double()``) evaluates the same graph in double precision: the
    reference against which fp32 rounding drift of the HIP path is measured (tests/test_gpu_network.py)."""
    from . import pg_ops
    f = (lambda t: t.to(dtype)) if dtype is not None else (lambda t: t)
    coords_float, superpoint = f(batch["locs_float"]), batch["superpoint"]
    centre = scatter(coords_float, superpoint, 0, None, "mean")
    feats = torch.cat((f(batch["feats"]), coords_float), 1)
    if dtype is not None and dtype != torch.float32:      # mean pooling of the voxel's points in the working precision
        v2p = batch["v2p_map"].long()
        n = v2p[:, 0].clamp(min=1)
        voxel_feats = torch.zeros(v2p.shape[0], feats.shape[1], dtype=dtype)
        for i in range(v2p.shape[1] - 1):
            live = v2p[:, 0] > i
            voxel_feats[live] += feats[v2p[live, 1 + i]] / n[live, None].to(dtype)
    else:
        voxel_feats = torch.from_numpy(pg_ops.voxelization(feats.numpy(), batch["v2p_map"].numpy(), mode))
    gi = batch["GIs"][0]
    ret = ref_model(voxel_feats, batch["voxel_locs"].numpy(), batch["spatial_shape"], batch["p2v_map"], superpoint,
                    centre, gi._edge_indexes.cpu(), f(gi._edgefeats.cpu()), batch["edge_u_list"], batch["edge_v_list"])
    loss_inp = {
        "point_labels": (batch["semantic_labels"], batch["instance_labels"]),
        "semantic_scores": ret["semantic_scores"],
        "superpoint_labels": (batch["superpoint_semantic_labels"], batch["superpoint_instance_labels"]),
        "sp_semantic": ret["sp_semantic_scores"],
        "sp_offset_vector": (ret["pred_sp_offset_vectors"], f(batch["superpoint_offset_vector"])),
        "sp_occupancy": (ret["pred_sp_occupancy"], f(batch["superpoint_instance_voxel_num"])),
        "sp_instance_size": (ret["pred_sp_ins_size"], f(batch["superpoint_instance_size"])),
        "sp_discriminative_features": (ret["sp_discriminative_feats"], batch["sp_batch_offsets"]),
    }
    loss, _ = criterion(loss_inp, epoch)
    return loss, ret
