"""Drop-in for ``import spconv`` (llijiang fork of traveller59/spconv v1.0 [UPSTREAM]) restricted to
the surface 3D-WSIS uses (modules/model/sparse_unet3d.py:8-36,112-143,254-298,
modules/model/backbone_3D_WSIS.py:42-55, train_scannetv2.py:191-194):

    SparseConvTensor, SparseSequential, SparseModule (also spconv.modules.SparseModule),
    SubMConv3d, SparseConv3d, SparseInverseConv3d

Every convolution runs on libwsis_hip.so (hash rulebook -> output-stationary implicit GEMM on the
fp32 MFMA); there is no CPU path.
"""
from .tensor import SparseConvTensor
from .modules import SparseModule, SparseSequential
from .conv import SparseConvolution, SubMConv3d, SparseConv3d, SparseInverseConv3d
from . import ops

__all__ = ["SparseConvTensor", "SparseModule", "SparseSequential", "SparseConvolution", "SubMConv3d",
           "SparseConv3d", "SparseInverseConv3d", "ops"]
