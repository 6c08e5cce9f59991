"""spconv.modules [UPSTREAM spconv v1.0 modules.py]: SparseModule marker and SparseSequential.

SparseSequential semantics relied on by the reference:
  * SparseModule children receive the SparseConvTensor;
  * ordinary nn.Modules (BatchNorm1d, ReLU, Identity) are applied to ``.features`` and the result is
    re-bound ON THE SAME tensor object (which is why modules/model/sparse_unet3d.py:164-169 snapshots
    ``identity`` first), skipped when the tensor has no active voxel;
  * children are named positionally ("0","1",...) or by the keys of one OrderedDict
    (sparse_unet3d.py:127,254) -- this fixes the state-dict key names.
"""
from collections import OrderedDict

import torch
from torch import nn

from .tensor import SparseConvTensor


def _fuse_bn_relu():
    import os
    # default on: with the step GPU-bound the fused kernels win (A/B on one MI355X: 21.5 vs 24.2 ms/step);
    # WSIS_FUSE_BN=0 falls back to torch.nn.BatchNorm1d + ReLU as separate modules
    return os.environ.get("WSIS_FUSE_BN", "1") != "0"


class SparseModule(nn.Module):
    """place holder: every module subclassing this takes a SparseConvTensor"""
    pass


def is_spconv_module(module):
    return isinstance(module, SparseModule)


def is_sparse_conv(module):
    from .conv import SparseConvolution
    return isinstance(module, SparseConvolution)


class SparseSequential(SparseModule):
    def __init__(self, *args, **kwargs):
        super(SparseSequential, self).__init__()
        if len(args) == 1 and isinstance(args[0], OrderedDict):
            for key, module in args[0].items():
                self.add_module(key, module)
        else:
            for idx, module in enumerate(args):
                self.add_module(str(idx), module)
        for name, module in kwargs.items():
            if name in self._modules:
                raise ValueError("name exists.")
            self.add_module(name, module)

    def __getitem__(self, idx):
        if not (-len(self) <= idx < len(self)):
            raise IndexError("index {} is out of range".format(idx))
        if idx < 0:
            idx += len(self)
        it = iter(self._modules.values())
        for _ in range(idx):
            next(it)
        return next(it)

    def __len__(self):
        return len(self._modules)

    def add(self, module, name=None):
        if name is None:
            name = str(len(self._modules))
            if name in self._modules:
                raise KeyError("name exists")
        self.add_module(name, module)

    def forward(self, input):
        mods = list(self._modules.values())
        i = 0
        while i < len(mods):
            module = mods[i]
            if is_spconv_module(module):
                input = module(input)
            elif isinstance(input, SparseConvTensor):
                synced = False
                if isinstance(module, nn.BatchNorm1d) and input.features.is_cuda:
                    import wsis_ops
                    synced = wsis_ops.sync_group(module) is not None
                if synced:
                    # statistics shared across ranks (wsis_parallel.convert_sync_batchnorm): EVERY rank enters the
                    # collective, also with zero rows here and whatever WSIS_FUSE_BN says -- a rank that skipped the
                    # layer would leave its peers waiting in the all-reduce
                    nxt = mods[i + 1] if i + 1 < len(mods) else None
                    relu = type(nxt) is nn.ReLU
                    input.features = wsis_ops.batch_norm_relu(input.features, module, relu=relu)
                    i += 1 if relu else 0
                elif input.indices.shape[0] != 0:
                    # peephole: BatchNorm1d followed by ReLU runs as ONE fused HIP operator (same module
                    # objects, parameters and state-dict; only the execution is fused)
                    nxt = mods[i + 1] if i + 1 < len(mods) else None
                    if (isinstance(module, nn.BatchNorm1d) and type(nxt) is nn.ReLU and input.features.is_cuda
                            and _fuse_bn_relu()):
                        import wsis_ops
                        input.features = wsis_ops.batch_norm_relu(input.features, module, relu=True)
                        i += 1
                    else:
                        input.features = module(input.features)
            else:
                input = module(input)
            i += 1
        return input
