"""Oracle for torch_scatter.scatter [UPSTREAM torch_scatter 2.0.x, pip dependency, SURVEY App. A.3].
TEST INFRASTRUCTURE ONLY.  parity unpinned (source not in /root/reference)."""
import torch


def scatter(src, index, dim=0, dim_size=None, reduce="sum"):
    """1-D index over dim 0 (modules/model/backbone_3D_WSIS.py:188,225,232,244)."""
    assert dim == 0
    index = index.long()
    S = int(index.max()) + 1 if dim_size is None else int(dim_size)
    shape = (S,) + tuple(src.shape[1:])
    if reduce in ("sum", "add"):
        return torch.zeros(shape, dtype=src.dtype).index_add_(0, index, src)
    if reduce == "mean":
        s = torch.zeros(shape, dtype=src.dtype).index_add_(0, index, src)
        cnt = torch.zeros(S, dtype=src.dtype).index_add_(0, index, torch.ones_like(index, dtype=src.dtype))
        cnt = cnt.clamp(min=1)
        return s / cnt.view((S,) + (1,) * (src.dim() - 1))
    if reduce in ("max", "min"):
        idx = index.view((-1,) + (1,) * (src.dim() - 1)).expand_as(src)
        out = torch.zeros(shape, dtype=src.dtype)
        return out.scatter_reduce(0, idx, src, "amax" if reduce == "max" else "amin", include_self=False)
    raise ValueError(reduce)
