"""spconv.SparseConvTensor [UPSTREAM spconv v1.0 __init__.py], used at train_scannetv2.py:191-194 and
modules/model/sparse_unet3d.py:164-167,325-328.  A plain holder; ``features`` is re-assignable."""
import numpy as np
import torch


class SparseConvTensor(object):
    def __init__(self, features, indices, spatial_shape, batch_size, grid=None):
        """
        features: [num_points, num_features] float tensor
        indices:  [num_points, 1 + 3] int32 tensor, column 0 = batch index
        spatial_shape: 3 ints (numpy array / list), same column order as ``indices[:, 1:]``
        batch_size: int >= number of batch items actually present (train_scannetv2.py:341-344
            passes the configured batch size with a one-scene batch)
        """
        self.features = features
        self.indices = indices
        if self.indices.dtype != torch.int32:
            self.indices = self.indices.int()
        if not self.indices.is_contiguous():
            self.indices = self.indices.contiguous()   # the kernels read raw [M,4] rows
        self.spatial_shape = spatial_shape
        self.batch_size = batch_size
        self.indice_dict = {}   # rulebook cache, shared by reference with outputs of spconv modules
        self.grid = grid
        self._hash = None       # (keys, vals, cap) coordinate hash of ``indices`` (built lazily)

    @property
    def spatial_size(self):
        return int(np.prod(self.spatial_shape))

    def find_indice_pair(self, key):
        if key is None:
            return None
        if key in self.indice_dict:
            return self.indice_dict[key]
        return None

    def dense(self, channels_first=True):
        """[B, C, S0, S1, S2] dense tensor (debug / tests)."""
        B = int(self.batch_size)
        S = [int(s) for s in self.spatial_shape]
        C = self.features.shape[1]
        out = torch.zeros([B] + S + [C], dtype=self.features.dtype, device=self.features.device)
        idx = self.indices.long()
        out[idx[:, 0], idx[:, 1], idx[:, 2], idx[:, 3]] = self.features
        if not channels_first:
            return out
        return out.permute(0, 4, 1, 2, 3).contiguous()

    @property
    def sparity(self):
        return self.indices.shape[0] / np.prod(self.spatial_shape) / self.batch_size
