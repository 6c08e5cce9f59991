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


def seeded_state_dict(shapes, seed=20260):
    """deterministic weights for a model whose state-dict has the reference's names: every entry is drawn
    from a generator seeded by its NAME, so the reference model (tests/golden/make_network_golden.py), the oracle and
    the product all load the same values whatever their construction order.  ``shapes``: {name: shape}."""
    import zlib
    out = {}
    for name in sorted(shapes):
        shape = tuple(int(v) for v in shapes[name])
        g = torch.Generator().manual_seed(seed + zlib.crc32(name.encode()))
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            out[name] = torch.zeros(shape, dtype=torch.int64)
        elif leaf == "running_var":
            out[name] = 1.0 + 0.2 * torch.rand(shape, generator=g)
        elif leaf == "running_mean":
            out[name] = 0.1 * torch.randn(shape, generator=g)
        elif len(shape) == 1 and leaf == "weight":          # BatchNorm scale
            out[name] = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif len(shape) == 1:                                # biases
            out[name] = 0.1 * torch.randn(shape, generator=g)
        else:
            # conv [k0,k1,k2,Cin,Cout]: fan-in k^3 Cin; Linear / GRU [out, in]: fan-in = in
            fan_in = int(np.prod(shape[:-1])) if len(shape) == 5 else shape[-1]
            out[name] = torch.randn(shape, generator=g) * (1.5 / np.sqrt(fan_in))
    return out
