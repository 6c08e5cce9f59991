"""Shared helpers for the tests: seeded synthetic sparse inputs."""
import numpy as np
import torch


def random_sparse_coords(seed, batch=2, shape=(9, 8, 7), density=0.3, surface=False):
    """unique int32 [M,4] (batch, x, y, z) voxel coordinates in random row order."""
    rng = np.random.default_rng(seed)
    rows = []
    for b in range(batch):
        if surface:
            # a wavy sheet: spatially coherent like a scanned surface
            xs, ys = np.meshgrid(np.arange(shape[0]), np.arange(shape[1]), indexing="ij")
            zs = ((np.sin(xs / 3.0 + b) + np.cos(ys / 4.0)) * shape[2] / 6 + shape[2] / 2).astype(np.int64)
            zs = np.clip(zs, 0, shape[2] - 1)
            pts = np.stack([xs.ravel(), ys.ravel(), zs.ravel()], 1)
            keep = rng.random(len(pts)) < density
            pts = pts[keep]
        else:
            occ = rng.random(shape) < density
            pts = np.argwhere(occ)
        pts = pts[rng.permutation(len(pts))]
        rows.append(np.concatenate([np.full((len(pts), 1), b), pts], 1))
    idx = np.concatenate(rows, 0).astype(np.int32)
    return idx


def dense_from_sparse(indices, feats, batch, shape):
    """[B, C, S0, S1, S2] float64"""
    C = feats.shape[1]
    d = torch.zeros([batch, C] + list(shape), dtype=torch.float64)
    ii = torch.as_tensor(indices).long()
    d[ii[:, 0], :, ii[:, 1], ii[:, 2], ii[:, 3]] = feats.double()
    return d
