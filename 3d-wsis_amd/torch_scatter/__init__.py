"""Drop-in for ``from torch_scatter import scatter, scatter_mean, scatter_min, scatter_max``
(torch_scatter 2.0.x [UPSTREAM]) for the forms 3D-WSIS uses: 1-D ``index`` over ``dim=0`` with
reduce in {'sum','add','mean','max','min'} (modules/model/backbone_3D_WSIS.py:188,225,232,244,
train_scannetv2.py:177).

Implementation: the index vector becomes a CSR once (stable radix sort), then one wavefront reduces one
segment in ascending point order -- no atomics, run-to-run deterministic (libwsis_hip.so).
"""
import torch
from torch.autograd import Function

import wsis_native as _n

__all__ = ["scatter", "scatter_sum", "scatter_add", "scatter_mean", "scatter_max", "scatter_min",
           "SegmentCSR", "segment_csr", "segment_csr_batch"]

_RED = {"sum": 0, "add": 0, "mean": 1, "max": 2}


class SegmentCSR(object):
    """perm int32 [N] (stable argsort of index) and offsets int32 [S+1]; reusable across calls that share
    the same index tensor (e.g. superpoint ids for centres and for features)."""

    def __init__(self, index, dim_size=None):
        _n.require_cuda(index)
        assert index.dim() == 1
        index = index.contiguous()
        if index.dtype != torch.int64:
            index = index.long()
        self.index = index
        N = index.numel()
        if dim_size is None:
            dim_size = int(index.max().item()) + 1 if N > 0 else 0
        self.N, self.S = N, int(dim_size)
        lib = _n.hip()
        dev = index.device
        ws_bytes = lib.wsis_segment_csr_workspace_bytes(N, self.S)
        if ws_bytes < 0:
            raise _n.WsisError("segment_csr workspace query failed")
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        self.perm = torch.empty(max(N, 1), dtype=torch.int32, device=dev)
        self.offsets = torch.empty(self.S + 1, dtype=torch.int32, device=dev)
        _n.check(lib.wsis_segment_csr(_n.ptr(index), N, self.S, _n.ptr(self.perm), _n.ptr(self.offsets),
                                      _n.ptr(ws), ws_bytes, _n.stream_ptr()), "segment_csr")


def segment_csr(index, dim_size=None):
    return SegmentCSR(index, dim_size)


def segment_csr_batch(pairs):
    """``[(index, dim_size), ...]`` (at most 8, all on one device, sizes known on the host) -> the SegmentCSR of each
    from ONE sort (wsis_segment_csr_batch): the per-batch structures of a step -- superpoint ids, p2v map, the edge
    lists of the affinity and ECC graphs -- cost one chain of ~12 launches instead of six."""
    import ctypes
    assert 1 <= len(pairs) <= 8
    idx = []
    for index, S in pairs:
        _n.require_cuda(index)
        index = index.contiguous()
        idx.append(index if index.dtype == torch.int64 else index.long())
    dev = idx[0].device
    n = len(idx)
    Ns = [int(t.numel()) for t in idx]
    Ss = [int(S) for _, S in pairs]
    lib = _n.hip()
    ws_bytes = lib.wsis_segment_csr_batch_workspace_bytes(sum(Ns))
    if ws_bytes < 0:
        raise _n.WsisError("segment_csr_batch workspace query failed")
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    perm_all = torch.empty(max(sum(Ns), 1), dtype=torch.int32, device=dev)
    off_all = torch.empty(sum(Ss) + n, dtype=torch.int32, device=dev)
    _n.check(lib.wsis_segment_csr_batch(n, (ctypes.c_void_p * n)(*[t.data_ptr() for t in idx]), (ctypes.c_int64 * n)(*Ns),
                                        (ctypes.c_int64 * n)(*Ss), _n.ptr(perm_all), _n.ptr(off_all), _n.ptr(ws), ws_bytes,
                                        _n.stream_ptr()), "segment_csr_batch")
    out, p0, o0 = [], 0, 0
    for index, N, S in zip(idx, Ns, Ss):
        c = SegmentCSR.__new__(SegmentCSR)
        c.index, c.N, c.S = index, N, S
        c.perm = perm_all[p0:p0 + max(N, 1)] if N > 0 else torch.empty(1, dtype=torch.int32, device=dev)
        c.offsets = off_all[o0:o0 + S + 1]
        out.append(c)
        p0 += N
        o0 += S + 1
    return out


class _SegmentReduce(Function):
    @staticmethod
    def forward(ctx, src, csr, reduce):
        x = src.contiguous().float()
        N = x.shape[0]
        assert N == csr.N, "src rows != index length"
        x2 = x.view(N, -1)
        C = x2.shape[1]
        out = torch.empty((csr.S, C), dtype=torch.float32, device=x.device)
        argmax = torch.empty((csr.S, C), dtype=torch.int32, device=x.device) if reduce == 2 else None
        if C > 0:
            _n.check(_n.hip().wsis_segment_reduce_fwd(_n.ptr(x2), _n.ptr(csr.perm), _n.ptr(csr.offsets),
                                                      _n.ptr(out), _n.ptr(argmax), N, csr.S, C, reduce,
                                                      _n.stream_ptr()), "segment_reduce_fwd")
        ctx.csr, ctx.reduce, ctx.argmax, ctx.in_shape = csr, reduce, argmax, src.shape
        return out.view((csr.S,) + tuple(src.shape[1:]))

    @staticmethod
    def backward(ctx, grad):
        csr, reduce, argmax = ctx.csr, ctx.reduce, ctx.argmax
        g = grad.contiguous().float().view(csr.S, -1)
        C = g.shape[1]
        if reduce == 2:
            dsrc = torch.zeros((csr.N, C), dtype=torch.float32, device=g.device)
        else:
            dsrc = torch.empty((csr.N, C), dtype=torch.float32, device=g.device)
        if C > 0 and csr.N > 0:
            _n.check(_n.hip().wsis_segment_reduce_bwd(_n.ptr(g), _n.ptr(csr.index), _n.ptr(csr.offsets),
                                                      _n.ptr(argmax), _n.ptr(dsrc), csr.N, csr.S, C, reduce,
                                                      _n.stream_ptr()), "segment_reduce_bwd")
        return dsrc.view(ctx.in_shape), None, None


def scatter(src, index, dim=0, out=None, dim_size=None, reduce="sum", csr=None):
    """torch_scatter.scatter for 1-D ``index`` along ``dim=0``.  ``csr`` (a SegmentCSR of the same index) is
    an extension that skips the sort when the index is reused."""
    assert out is None, "out= is not used by 3D-WSIS"
    if dim < 0:
        dim += src.dim()
    assert dim == 0 and index.dim() == 1, "only 1-D index over dim 0 (the forms 3D-WSIS uses)"
    _n.require_cuda(src, index)
    if csr is None:
        csr = SegmentCSR(index, dim_size)
    if reduce == "min":
        return -_SegmentReduce.apply(-src, csr, 2)
    if reduce not in _RED:
        raise ValueError(f"unsupported reduce {reduce!r}")
    return _SegmentReduce.apply(src, csr, _RED[reduce])


def scatter_sum(src, index, dim=0, out=None, dim_size=None):
    return scatter(src, index, dim, out, dim_size, "sum")


scatter_add = scatter_sum


def scatter_mean(src, index, dim=0, out=None, dim_size=None):
    return scatter(src, index, dim, out, dim_size, "mean")


def _arg_from(src, index, dim_size, sign):
    csr = SegmentCSR(index, dim_size)
    x = (sign * src).contiguous().float()
    val = _SegmentReduce.apply(x, csr, 2)
    # recompute arg (int64, torch_scatter returns src.size(dim) for empty segments)
    N = x.shape[0]
    x2 = x.view(N, -1)
    out = torch.empty((csr.S, x2.shape[1]), dtype=torch.float32, device=x.device)
    arg = torch.empty((csr.S, x2.shape[1]), dtype=torch.int32, device=x.device)
    _n.check(_n.hip().wsis_segment_reduce_fwd(_n.ptr(x2.detach()), _n.ptr(csr.perm), _n.ptr(csr.offsets),
                                              _n.ptr(out), _n.ptr(arg), N, csr.S, x2.shape[1], 2,
                                              _n.stream_ptr()), "segment_reduce_fwd")
    arg = arg.long()
    arg[arg < 0] = N
    return sign * val, arg.view(val.shape)


def scatter_max(src, index, dim=0, out=None, dim_size=None):
    assert dim in (0, -src.dim()) and out is None
    return _arg_from(src, index, dim_size, 1.0)


def scatter_min(src, index, dim=0, out=None, dim_size=None):
    assert dim in (0, -src.dim()) and out is None
    return _arg_from(src, index, dim_size, -1.0)
