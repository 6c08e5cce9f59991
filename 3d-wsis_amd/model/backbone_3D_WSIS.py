"""``Network`` of 3D-WSIS on the MI355X operator surface.

Same constructor argument (``param`` with input_channel/use_coords/blocks/block_reps/media/classes/
fix_module), same ``forward(input, input_map, extra_data)`` signature, same ``ret`` keys and the same module
tree / state-dict names as the reference's modules/model/backbone_3D_WSIS.py:24-255, so the reference's
train/test scripts and checkpoints can use it as a drop-in.

Differences, all inside the hot path: sparse convs, voxel->point gather, superpoint scatter-mean and the
edge-affinity attention run as HIP kernels from libwsis_hip.so; the attention block (:218-244) is ONE fused
kernel (fwd + bwd) instead of ~10 elementwise/scatter launches.
"""
import ast
import functools
import math

import torch
import torch.nn as nn

import spconv
import graphnet
import wsis_ops
from sparse_unet3d import ResidualBlock, UBlock
from torch_scatter import SegmentCSR, scatter

__all__ = ["Network"]


class _Head(nn.Sequential):
    """Linear -> BatchNorm1d -> ReLU -> Linear with the reference's child names (0,1,2,3); BN+ReLU run fused."""

    def forward(self, x):
        import os
        if not x.is_cuda:
            return super().forward(x)
        x = wsis_ops.tall_linear(x, self[0])
        if os.environ.get("WSIS_FUSE_BN", "1") == "0" and wsis_ops.sync_group(self[1]) is None:
            x = self[2](self[1](x))
        else:
            x = wsis_ops.batch_norm_relu(x, self[1], relu=True)
        return wsis_ops.tall_linear(x, self[3])


class Network(nn.Module):
    def __init__(self, param):
        super().__init__()
        self.input_channel = param.input_channel
        self.use_coords = param.use_coords
        self.blocks = param.blocks
        self.block_reps = param.block_reps
        self.media = param.media
        self.classes = param.classes
        fix = getattr(param, "fix_module", "[]")
        self.fix_module = ast.literal_eval(fix) if isinstance(fix, str) else list(fix)

        if self.use_coords:
            self.input_channel += 3

        self.input_conv = spconv.SparseSequential(
            spconv.SubMConv3d(self.input_channel, self.media, kernel_size=3, padding=1, bias=False,
                              indice_key="subm1"))
        norm_fn = functools.partial(nn.BatchNorm1d, eps=1e-4, momentum=0.1)
        block_list = [self.media * (i + 1) for i in range(self.blocks)]
        self.unet = UBlock(block_list, norm_fn, self.block_reps, ResidualBlock, indice_key_id=1)
        self.output_layer = spconv.SparseSequential(norm_fn(self.media), nn.ReLU(inplace=True))

        def head(cin, cout):
            return _Head(nn.Linear(cin, cin, bias=True), norm_fn(cin), nn.ReLU(inplace=True),
                         nn.Linear(cin, cout))

        self.linear = head(self.media, self.classes)                       # point semantic
        self.ecc = graphnet.GraphNetwork("gru_7_0,f_64,b,r", nfeat=self.media, fnet_widths=[13, 32, 128, 64],
                                         fnet_orthoinit=True, fnet_llbias=True, fnet_bnidx=2, use_pyg=True,
                                         cuda=True)
        sp_feat_dim = 64
        self.sp_sem_seg = head(sp_feat_dim, self.classes)
        self.sp_offset_vector_head = head(sp_feat_dim, 3)
        self.sp_occupancy_head = head(sp_feat_dim, 1)
        self.sp_ins_size_head = head(sp_feat_dim, 1)

        d_model = 64
        self.fc_position = nn.Sequential(nn.Linear(3, 16), nn.ReLU(), nn.Linear(16, 1))
        self.w_qs = nn.Linear(d_model, d_model, bias=False)
        self.w_ks = nn.Linear(d_model, d_model, bias=False)
        self.w_vs = nn.Linear(d_model, d_model, bias=False)
        self.feature_term = head(d_model, 7)

        for name in self.fix_module:
            module = getattr(self, name)
            module.eval()
            for p in module.parameters():
                p.requires_grad = False

    @staticmethod
    def freeze_bn(module):
        for child in module.modules():
            if isinstance(child, nn.BatchNorm1d):
                child.weight.requires_grad_(False)
                child.bias.requires_grad_(False)

    @staticmethod
    def set_bn_init(m):
        if m.__class__.__name__.find("BatchNorm") != -1:
            try:
                m.weight.data.fill_(1.0)
                m.bias.data.fill_(0.0)
            except Exception:
                pass

    def forward(self, input, input_map, extra_data):
        ret = {}
        for name in self.fix_module:
            getattr(self, name).eval()

        import os
        import wsis_parallel
        sync_bn = self.training and wsis_parallel.sync_batchnorm_active(self)
        # work that does not depend on the UNet goes to branch streams (WSIS_BRANCH=0: one stream): the filter net of the
        # superpoint graph here, the point-level head below.  Not with shared BatchNorm statistics: the ranks' collectives
        # must be issued in one order.
        self.ecc.set_info(extra_data["GIs"], cuda=True)
        branch = None if sync_bn else (wsis_ops.branch_stream(input.features.device, 0)
                                       if input.features.is_cuda else None)
        params_ready = None
        if branch is not None:
            params_ready = torch.cuda.Event()
            params_ready.record()            # behind the previous optimizer step, in front of this forward pass
        # SyncBatchNorm keeps the native executor: the op list is issued in parts around the statistics exchange of every
        # layer (unet_native._BnSync; WSIS_SYNC_BN_NATIVE=0: the per-module walk with _SyncBatchNormReLU)
        sync_native = sync_bn and os.environ.get("WSIS_SYNC_BN_NATIVE", "1") != "0"
        if input.features.is_cuda and os.environ.get("WSIS_NATIVE_UNET", "1") != "0" and (not sync_bn or sync_native):
            # input_conv -> unet -> output_layer recorded as an op list and issued by one native call per pass
            # (model/unet_native.py); WSIS_NATIVE_UNET=0 walks the modules instead (same kernels, same results)
            import unet_native
            group = None
            if sync_native:
                group = next((wsis_ops.sync_group(m) for m in self.unet.modules()
                              if isinstance(m, nn.BatchNorm1d) and wsis_ops.sync_group(m) is not None), None)
            voxel_feats = unet_native.run_unet(self, input, sync_group=group)
            if branch is not None:
                # issued behind the rulebook build of this pass on the same stream: runs while the UNet does
                for gconv in self.ecc.gconvs:
                    gconv.prefetch_filter_state(branch, params_ready)
        else:
            if input.features.is_cuda and os.environ.get("WSIS_PREBUILD", "1") != "0":
                # all 5 + 4 rulebooks up front: their host syncs happen before the first conv is queued
                spconv.ops.prebuild_unet_rulebooks(input, self.blocks)
            elif input.features.is_cuda:
                # rulebooks built lazily by the modules: the weight-gradient launch-plan hint (process-global, sticky)
                # still has to be THIS batch's, not whatever batch ran last
                spconv.ops.hint_batch_rows(int(input.indices.shape[0]))
            output = self.input_conv(input)
            output = self.unet(output)
            output = self.output_layer(output)
            voxel_feats = output.features
            if voxel_feats.is_cuda:
                spconv.ops.verify_pending_counts()      # hint-sized rulebooks are checked inside the same pass
        if voxel_feats.is_cuda:                                       # [N, m] voxel -> point
            output_feats = wsis_ops.gather_rows(voxel_feats, input_map, extra_data.get("p2v_csr"))
        else:
            output_feats = voxel_feats[input_map.long()]

        head_branch = None if sync_bn else (wsis_ops.branch_stream(output_feats.device, 1)
                                            if output_feats.is_cuda else None)
        if head_branch is not None:
            # the point-level head (4 launches forward, ~14 backward over N ~ 2*10^5 rows) beside the superpoint chain
            # (~60 launches over S ~ 2*10^3 rows each way); joined at the end of this forward
            main = torch.cuda.current_stream()
            head_branch.wait_stream(main)
            with torch.cuda.stream(head_branch):
                ret["semantic_scores"] = self.linear(output_feats)  # [N, nClass]
            output_feats.record_stream(head_branch)
        else:
            ret["semantic_scores"] = self.linear(output_feats)      # [N, nClass]

        superpoint = extra_data["superpoint"].long()
        sp_csr = extra_data.get("superpoint_csr")                   # optional reuse (extension)
        embeddings = scatter(output_feats, superpoint, dim=0, reduce="mean", csr=sp_csr)

        ecc_outputs = self.ecc(embeddings)

        # the four heads and the q / k / v layers read the same rows: ONE operator (3 launches forward, 4 backward,
        # csrc/heads.hip) where it applies, else module by module
        fused = wsis_ops.sp_heads(ecc_outputs, [self.sp_sem_seg, self.sp_offset_vector_head, self.sp_occupancy_head,
                                                self.sp_ins_size_head], [self.w_qs, self.w_ks, self.w_vs])
        if fused is not None:
            (sem, off, occ, size), (q, k, v) = fused
        else:
            sem, off = self.sp_sem_seg(ecc_outputs), self.sp_offset_vector_head(ecc_outputs)
            occ, size = self.sp_occupancy_head(ecc_outputs), self.sp_ins_size_head(ecc_outputs)
            q, k, v = (wsis_ops.tall_linear(ecc_outputs, m) if ecc_outputs.is_cuda else m(ecc_outputs)
                       for m in (self.w_qs, self.w_ks, self.w_vs))
        ret["sp_semantic_scores"] = sem
        ret["pred_sp_offset_vectors"] = off
        ret["pred_sp_occupancy"] = occ.squeeze(-1)
        ret["pred_sp_ins_size"] = size.squeeze(-1)

        # ---- affinity between adjacent superpoints (backbone_3D_WSIS.py:207-253) ----
        centre = extra_data["superpoint_cenetr_xyz"]
        edge_u, edge_v = extra_data["edge_u_list"], extra_data["edge_v_list"]
        graph = extra_data.get("edge_graph")
        if graph is None:
            graph = wsis_ops.EdgeGraph(edge_u, edge_v, ecc_outputs.shape[0], num_src=extra_data.get("edge_src_rows"))
        pos_enc = wsis_ops.edge_position_encoding(self.fc_position, centre, edge_u, edge_v)      # one launch each way
        if pos_enc is None:
            pos_enc = wsis_ops.tall_sequential(self.fc_position, centre[edge_u] - centre[edge_v]).reshape(-1)
        affinity, res = wsis_ops.edge_affinity(q, k, v, pos_enc, graph, 1.0 / math.sqrt(k.size(-1)))
        ret["edge_affinity"] = affinity

        if res.shape[0] == ecc_outputs.shape[0]:
            sp_feat = ecc_outputs + res
        else:
            pad = torch.zeros((ecc_outputs.shape[0] - res.shape[0], res.shape[1]), dtype=res.dtype,
                              device=res.device)
            sp_feat = ecc_outputs + torch.cat((res, pad), 0)
        fused = wsis_ops.sp_heads(sp_feat, [self.feature_term])
        ret["sp_discriminative_feats"] = fused[0][0] if fused is not None else self.feature_term(sp_feat)
        if head_branch is not None:
            torch.cuda.current_stream().wait_stream(head_branch)
            ret["semantic_scores"].record_stream(torch.cuda.current_stream())
        return ret
