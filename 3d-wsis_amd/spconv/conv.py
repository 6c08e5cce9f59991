"""SubMConv3d / SparseConv3d / SparseInverseConv3d [UPSTREAM spconv v1.0 conv.py], constructor kwargs
and parameter layout as the reference uses them (modules/model/sparse_unet3d.py:17-36,113-143,254-298):
``weight`` is ``[k0,k1,k2,Cin,Cout]`` (state-dict compatible), ``bias`` optional.
"""
import math

import numpy as np
import torch
from torch import nn
from torch.nn import init
from torch.nn.parameter import Parameter

from . import ops
from .modules import SparseModule
from .tensor import SparseConvTensor


class SparseConvolution(SparseModule):
    def __init__(self, ndim, in_channels, out_channels, kernel_size=3, stride=1, padding=0, dilation=1,
                 groups=1, bias=True, subm=False, output_padding=0, transposed=False, inverse=False,
                 indice_key=None):
        super(SparseConvolution, self).__init__()
        assert groups == 1, "groups != 1 is not used by 3D-WSIS"
        assert ndim == 3
        kernel_size = ops._triple(kernel_size)
        stride = ops._triple(stride)
        padding = ops._triple(padding)
        dilation = ops._triple(dilation)
        assert dilation == [1, 1, 1], "dilation != 1 is not used by 3D-WSIS"
        assert not transposed, "SparseConvTranspose3d is not used by 3D-WSIS"
        self.ndim = ndim
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.kernel_size = kernel_size
        self.conv1x1 = int(np.prod(kernel_size)) == 1
        self.stride = stride
        self.padding = padding
        self.dilation = dilation
        self.transposed = transposed
        self.inverse = inverse
        self.output_padding = ops._triple(output_padding)
        self.groups = groups
        self.subm = subm
        self.indice_key = indice_key
        self.weight = Parameter(torch.Tensor(*kernel_size, in_channels, out_channels))
        if bias:
            self.bias = Parameter(torch.Tensor(out_channels))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        # upstream applies kaiming_uniform_(a=sqrt(5)) to the 5-D [k0,k1,k2,Cin,Cout] tensor
        init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in, _ = init._calculate_fan_in_and_fan_out(self.weight)
            bound = 1 / math.sqrt(fan_in)
            init.uniform_(self.bias, -bound, bound)

    def extra_repr(self):
        return "{}, {}, kernel_size={}, stride={}, padding={}, subm={}, inverse={}, indice_key={}".format(
            self.in_channels, self.out_channels, self.kernel_size, self.stride, self.padding, self.subm,
            self.inverse, self.indice_key)

    def forward(self, input):
        assert isinstance(input, SparseConvTensor)
        features = input.features
        indices = input.indices
        spatial_shape = [int(s) for s in input.spatial_shape]
        batch_size = input.batch_size
        if not self.subm:
            if self.inverse:
                out_spatial_shape = None  # restored from the rulebook below
            else:
                out_spatial_shape = ops.get_conv_output_size(spatial_shape, self.kernel_size, self.stride,
                                                             self.padding, self.dilation)
        else:
            out_spatial_shape = spatial_shape

        if self.conv1x1:
            # upstream: features @ weight.view(Cin, Cout) (+ bias), no rulebook
            M = features.shape[0]
            out_features = ops.sparse_conv(features, self.weight, self.bias, None, None, None, None, 0, M)
            out_tensor = SparseConvTensor(out_features, indices, input.spatial_shape, batch_size)
            out_tensor.indice_dict = input.indice_dict
            out_tensor.grid = input.grid
            out_tensor._hash = input._hash
            return out_tensor

        rb = input.find_indice_pair(self.indice_key)
        if self.inverse:
            assert rb is not None and rb.kind == "down", "SparseInverseConv3d needs the rulebook of its SparseConv3d"
            assert rb.out_indices.shape[0] == indices.shape[0], "inverse conv input rows != coupled conv output rows"
            M_out = rb.in_indices.shape[0]
            out_features = ops.sparse_conv(features, self.weight, self.bias, rb.nbr_up_p, rb.order_up, rb.nbr_p,
                                           rb.order, 0, M_out)
            out_tensor = SparseConvTensor(out_features, rb.in_indices, np.array(rb.in_shape), batch_size)
            out_tensor.indice_dict = input.indice_dict
            out_tensor.grid = input.grid
            return out_tensor

        if self.subm:
            if rb is None or rb.kind != "subm" or rb.in_indices.shape[0] != indices.shape[0]:
                hash_tab = input._hash
                if hash_tab is None:
                    hash_tab = ops.build_hash(indices, spatial_shape)
                    input._hash = hash_tab
                rb = ops.build_subm_rulebook(indices, spatial_shape, self.kernel_size, self.padding, hash_tab)
                if self.indice_key is not None:
                    input.indice_dict[self.indice_key] = rb
            M = indices.shape[0]
            out_features = ops.sparse_conv(features, self.weight, self.bias, rb.nbr_p, rb.order, rb.nbr_p, rb.order,
                                           1, M)
            out_tensor = SparseConvTensor(out_features, indices, input.spatial_shape, batch_size)
            out_tensor.indice_dict = input.indice_dict
            out_tensor.grid = input.grid
            out_tensor._hash = input._hash
            return out_tensor

        # strided SparseConv3d
        if rb is None or rb.kind != "down" or rb.in_indices.shape[0] != indices.shape[0]:
            rb = ops.build_down_rulebook(indices, spatial_shape, self.kernel_size, self.stride, self.padding)
            if self.indice_key is not None:
                input.indice_dict[self.indice_key] = rb
        M_out = rb.out_indices.shape[0]
        out_features = ops.sparse_conv(features, self.weight, self.bias, rb.nbr_p, rb.order, rb.nbr_up_p,
                                       rb.order_up, 0, M_out)
        out_tensor = SparseConvTensor(out_features, rb.out_indices, np.array(out_spatial_shape), batch_size)
        out_tensor.indice_dict = input.indice_dict
        out_tensor.grid = input.grid
        out_tensor._hash = rb.out_hash
        return out_tensor


class SparseConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 bias=True, indice_key=None):
        super(SparseConv3d, self).__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation,
                                           groups, bias, indice_key=indice_key)


class SubMConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 bias=True, indice_key=None):
        super(SubMConv3d, self).__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation,
                                         groups, bias, True, indice_key=indice_key)


class SparseInverseConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, indice_key, bias=True):
        super(SparseInverseConv3d, self).__init__(3, in_channels, out_channels, kernel_size, bias=bias,
                                                  inverse=True, indice_key=indice_key)
