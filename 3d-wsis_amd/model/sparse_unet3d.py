"""Sparse UNet blocks of 3D-WSIS on the MI355X spconv drop-in.

Mirrors the module tree (and therefore the state-dict key names) of the reference's
modules/model/sparse_unet3d.py: ``ResidualBlock`` (:103-172) and the recursive ``UBlock`` (:213-350).
Only the pre-norm (normalize_before=True) residual configuration the active model uses is built
(backbone_3D_WSIS.py:46-50); VGG/Asym blocks are not selected by any reference config.
"""
import functools
from collections import OrderedDict

import torch
import torch.nn as nn

import spconv
from spconv.modules import SparseModule


def _snapshot(t):
    """A fresh SparseConvTensor sharing features/indices but NOT the rulebook dict -- the reference builds
    ``identity`` this way before SparseSequential re-binds ``.features`` (sparse_unet3d.py:164-167,325-328)."""
    return spconv.SparseConvTensor(t.features, t.indices, t.spatial_shape, t.batch_size)


class ResidualBlock(SparseModule):
    def __init__(self, in_channels, out_channels, norm_fn=None, indice_key=None, normalize_before=True):
        super().__init__()
        if norm_fn is None:
            norm_fn = functools.partial(nn.BatchNorm1d, eps=1e-4, momentum=0.1)
        assert normalize_before, "3D-WSIS uses the pre-norm residual block"
        if in_channels == out_channels:
            self.i_branch = spconv.SparseSequential(nn.Identity())
        else:
            self.i_branch = spconv.SparseSequential(
                spconv.SubMConv3d(in_channels, out_channels, kernel_size=1, bias=False))
        self.conv_branch = spconv.SparseSequential(
            norm_fn(in_channels),
            nn.ReLU(),
            spconv.SubMConv3d(in_channels, out_channels, kernel_size=3, padding=1, bias=False,
                              indice_key=indice_key),
            norm_fn(out_channels),
            nn.ReLU(),
            spconv.SubMConv3d(out_channels, out_channels, kernel_size=3, padding=1, bias=False,
                              indice_key=indice_key))

    def forward(self, input):
        identity = _snapshot(input)
        output = self.conv_branch(input)
        output.features = output.features + self.i_branch(identity).features
        return output


class UBlock(nn.Module):
    def __init__(self, nPlanes, norm_fn=None, block_reps=2, block=ResidualBlock, indice_key_id=1,
                 normalize_before=True, return_blocks=False):
        super().__init__()
        if norm_fn is None:
            norm_fn = functools.partial(nn.BatchNorm1d, eps=1e-4, momentum=0.1)
        assert normalize_before and not return_blocks
        self.nPlanes = nPlanes
        p0 = nPlanes[0]
        self.blocks = spconv.SparseSequential(OrderedDict(
            (f"block{i}", block(p0, p0, norm_fn, indice_key=f"subm{indice_key_id}")) for i in range(block_reps)))
        if len(nPlanes) > 1:
            self.conv = spconv.SparseSequential(
                norm_fn(p0),
                nn.ReLU(),
                spconv.SparseConv3d(p0, nPlanes[1], kernel_size=2, stride=2, bias=False,
                                    indice_key=f"spconv{indice_key_id}"))
            self.u = UBlock(nPlanes[1:], norm_fn, block_reps, block, indice_key_id=indice_key_id + 1)
            self.deconv = spconv.SparseSequential(
                norm_fn(nPlanes[1]),
                nn.ReLU(),
                spconv.SparseInverseConv3d(nPlanes[1], p0, kernel_size=2, bias=False,
                                           indice_key=f"spconv{indice_key_id}"))
            self.blocks_tail = spconv.SparseSequential(OrderedDict(
                (f"block{i}", block(p0 * (2 - i), p0, norm_fn, indice_key=f"subm{indice_key_id}"))
                for i in range(block_reps)))

    def forward(self, input):
        output = self.blocks(input)
        identity = _snapshot(output)
        if len(self.nPlanes) > 1:
            output_decoder = self.conv(output)
            output_decoder = self.u(output_decoder)
            output_decoder = self.deconv(output_decoder)
            output.features = torch.cat((identity.features, output_decoder.features), dim=1)
            output = self.blocks_tail(output)
        return output
