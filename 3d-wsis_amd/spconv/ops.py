"""Rulebooks and the sparse-convolution autograd Function on top of libwsis_hip.so.

Restates [UPSTREAM] spconv v1.0 ops.get_indice_pairs / functional.Sparse*ConvFunction
 for the call sites in modules/model/sparse_unet3d.py:130,261,292.

Native rulebook = gather tables ``nbr[K, rows]`` (row of the other side or -1) instead of upstream's
per-offset pair lists; ``Rulebook.to_pairs()`` exports the upstream ``(indice_pairs, indice_pair_num)``
form for parity tests.
"""
import os

import numpy as np
import torch
from torch.autograd import Function

import wsis_native as _n


def _use_mask_order():
    return os.environ.get("WSIS_MASK_ORDER", "1") != "0"


def _pow2_cap(m):
    cap = 16
    while cap < 2 * m:
        cap <<= 1
    return cap


def _triple(v):
    if isinstance(v, (list, tuple, np.ndarray)):
        v = [int(x) for x in v]
        assert len(v) == 3
        return v
    return [int(v)] * 3


def get_conv_output_size(input_size, kernel_size, stride, padding, dilation):
    out = []
    for i in range(len(input_size)):
        size = (input_size[i] + 2 * padding[i] - dilation[i] * (kernel_size[i] - 1) - 1) // stride[i] + 1
        out.append(int(size))
    return out


def _check_indices(indices):
    if indices.dtype != torch.int32 or indices.dim() != 2 or indices.shape[1] != 4 or not indices.is_contiguous():
        raise _n.WsisError("indices must be a contiguous int32 [M,4] tensor")


def build_hash(indices, spatial_shape):
    """coordinate hash (keys int64[cap], vals int32[cap], cap) of int32 [M,4] indices."""
    _n.require_cuda(indices)
    _check_indices(indices)
    M = indices.shape[0]
    cap = _pow2_cap(M)
    keys = torch.empty(cap, dtype=torch.int64, device=indices.device)
    vals = torch.empty(cap, dtype=torch.int32, device=indices.device)
    _n.check(_n.hip().wsis_hash_build(_n.ptr(indices), M, _n.i32x3(spatial_shape), _n.ptr(keys), _n.ptr(vals),
                                      cap, _n.stream_ptr()), "hash_build")
    return keys, vals, cap


def _tile_block_shift():
    return int(os.environ.get("WSIS_TILE_BLOCK_SHIFT", "4"))


def _tile_order(indices, mask):
    """stable sort of the rows by (batch, Morton block, offset mask) -> int32 [M] tile order"""
    bs = _tile_block_shift()
    if bs < 0:
        return _mask_order(mask)
    M = indices.shape[0]
    lib = _n.hip()
    ws_bytes = lib.wsis_tile_order_workspace_bytes(M)
    if ws_bytes < 0:
        raise _n.WsisError("tile_order workspace query failed")
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=indices.device)
    order = torch.empty(M, dtype=torch.int32, device=indices.device)
    _n.check(lib.wsis_tile_order(_n.ptr(indices), _n.ptr(mask), M, bs, _n.ptr(order), _n.ptr(ws), ws_bytes,
                                 _n.stream_ptr()), "tile_order")
    return order


def _tile_batch_enabled(batch_size):
    """one sort for all tables of a pyramid (WSIS_TILE_BATCH=0: one sort per table)"""
    return (os.environ.get("WSIS_TILE_BATCH", "1") != "0" and _use_mask_order() and _tile_block_shift() >= 0
            and 1 <= int(batch_size) <= 16)


def finish_tile_orders(deferred, batch_size):
    """``deferred``: [(rulebook, "order" | "order_up", indices, mask)] collected while the tables of a pyramid were
    built without their tile orders.  One ``wsis_tile_order_batch`` call (a single sort with the table number in the
    top key bits) produces all of them -- the same orders ``_tile_order`` gives table by table -- then every rulebook
    is packed."""
    import ctypes
    lib = _n.hip()
    todo = [d for d in deferred if d[2].shape[0] > 0]
    rbs = []
    for d in deferred:
        if not any(d[0] is r for r in rbs):
            rbs.append(d[0])
    for i in range(0, len(todo), 16):
        part = todo[i:i + 16]
        n = len(part)
        Ms = [int(d[2].shape[0]) for d in part]
        N = sum(Ms)
        dev = part[0][2].device
        ws_bytes = lib.wsis_tile_order_batch_workspace_bytes(N)
        if ws_bytes < 0:
            raise _n.WsisError("tile_order_batch workspace query failed")
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        order_all = torch.empty(N, dtype=torch.int32, device=dev)
        h_ind = (ctypes.c_void_p * n)(*[d[2].data_ptr() for d in part])
        h_mask = (ctypes.c_void_p * n)(*[d[3].data_ptr() for d in part])
        h_M = (ctypes.c_int64 * n)(*Ms)
        _n.check(lib.wsis_tile_order_batch(n, h_ind, h_mask, h_M, _tile_block_shift(), int(batch_size),
                                           _n.ptr(order_all), _n.ptr(ws), ws_bytes, _n.stream_ptr()),
                 "tile_order_batch")
        off = 0
        for d, M in zip(part, Ms):
            setattr(d[0], d[1], order_all[off:off + M])
            off += M
    # packed tables (columns in tile order) of every rulebook, 16 tables per launch
    jobs = []
    for rb in rbs:
        for nbr, order, name in ((rb.nbr, rb.order, "nbr_p"), (rb.nbr_up, rb.order_up, "nbr_up_p")):
            if nbr is None:
                continue
            if order is None or nbr.shape[1] == 0:
                setattr(rb, name, nbr)
                continue
            out = torch.empty_like(nbr)
            setattr(rb, name, out)
            jobs.append((nbr, order, out))
    for i in range(0, len(jobs), 16):
        part = jobs[i:i + 16]
        n = len(part)
        _n.check(lib.wsis_rulebook_pack_batch(
            n, (ctypes.c_void_p * n)(*[j[0].data_ptr() for j in part]),
            (ctypes.c_void_p * n)(*[j[1].data_ptr() for j in part]),
            (ctypes.c_void_p * n)(*[j[2].data_ptr() for j in part]),
            (ctypes.c_int64 * n)(*[int(j[0].shape[1]) for j in part]),
            (ctypes.c_int32 * n)(*[int(j[0].shape[0]) for j in part]), _n.stream_ptr()), "rulebook_pack_batch")


def _mask_order(mask):
    M = mask.shape[0]
    lib = _n.hip()
    ws_bytes = lib.wsis_mask_order_workspace_bytes(M)
    if ws_bytes < 0:
        raise _n.WsisError("mask_order workspace query failed")
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=mask.device)
    order = torch.empty(M, dtype=torch.int32, device=mask.device)
    _n.check(lib.wsis_mask_order(_n.ptr(mask), M, _n.ptr(order), _n.ptr(ws), ws_bytes, _n.stream_ptr()),
             "mask_order")
    return order


def _pack(nbr, order):
    if order is None or nbr.shape[1] == 0:
        return nbr
    K, M = nbr.shape
    packed = torch.empty_like(nbr)
    _n.check(_n.hip().wsis_rulebook_pack(_n.ptr(nbr), _n.ptr(order), _n.ptr(packed), M, K, _n.stream_ptr()),
             "rulebook_pack")
    return packed


class Rulebook(object):
    """What ``indice_dict[indice_key]`` holds (upstream: a 5-tuple)."""

    def __init__(self, kind, ksize, stride, padding, in_indices, out_indices, in_shape, out_shape):
        self.kind = kind                # "subm" | "down"
        self.ksize, self.stride, self.padding = ksize, stride, padding
        self.in_indices, self.out_indices = in_indices, out_indices
        self.in_shape, self.out_shape = in_shape, out_shape
        self.K = int(np.prod(ksize))
        # subm: nbr [K, M]; down: nbr (= nbr_down) [K, M_out] and nbr_up [K, M_in]
        self.nbr = None
        self.nbr_up = None
        self.order = None
        self.order_up = None
        self.nbr_p = None       # packed tables (columns in tile order) fed to the kernels
        self.nbr_up_p = None
        self.out_hash = None

    def pack(self):
        self.nbr_p = _pack(self.nbr, self.order)
        if self.nbr_up is not None:
            self.nbr_up_p = _pack(self.nbr_up, self.order_up)

    # upstream-format export (tests / INTEGRATION.md): indice_pairs int32 [K,2,maxP] (-1 padded),
    # indice_pair_num int32 [K]; pair order inside an offset = ascending output row.
    def to_pairs(self):
        nbr = self.nbr
        K, M = nbr.shape
        valid = nbr >= 0
        num = valid.sum(1).to(torch.int32)
        maxp = int(num.max().item()) if K > 0 and M > 0 else 0
        pairs = torch.full((K, 2, max(maxp, 1)), -1, dtype=torch.int32, device=nbr.device)
        for k in range(K):
            o = torch.nonzero(valid[k], as_tuple=False).flatten()
            pairs[k, 0, :o.numel()] = nbr[k, o]
            pairs[k, 1, :o.numel()] = o.to(torch.int32)
        return pairs, num


def build_subm_rulebook(indices, spatial_shape, ksize, padding, hash_tab=None, deferred=None):
    """SubMConv3d rulebook (a5): out rows == in rows, nbr[k][o] = i with coord_i = coord_o - pad + kappa.
    With ``deferred`` (a list) the tile order and the packed table are left to ``finish_tile_orders``."""
    _n.require_cuda(indices)
    _check_indices(indices)
    M = indices.shape[0]
    K = int(np.prod(ksize))
    dev = indices.device
    if hash_tab is None:
        hash_tab = build_hash(indices, spatial_shape)
    keys, vals, cap = hash_tab
    rb = Rulebook("subm", ksize, [1, 1, 1], padding, indices, indices, list(spatial_shape), list(spatial_shape))
    rb.nbr = torch.empty((K, M), dtype=torch.int32, device=dev)
    mask = torch.empty(M, dtype=torch.int32, device=dev) if K <= 32 else None
    _n.check(_n.hip().wsis_rulebook_subm(_n.ptr(indices), M, _n.i32x3(spatial_shape), _n.i32x3(ksize),
                                         _n.i32x3(padding), _n.ptr(keys), _n.ptr(vals), cap, _n.ptr(rb.nbr),
                                         _n.ptr(mask), _n.stream_ptr()), "rulebook_subm")
    rb.out_hash = hash_tab
    if deferred is not None and mask is not None and _use_mask_order():
        deferred.append((rb, "order", indices, mask))
        return rb
    if mask is not None and _use_mask_order() and M > 0:
        rb.order = _tile_order(indices, mask)
    rb.pack()
    return rb


def build_down_rulebook(indices, spatial_shape, ksize, stride, padding):
    """SparseConv3d rulebook (a6): output rows = ascending linear index of the reachable coarse voxels."""
    gen = _down_rulebook_gen(indices, spatial_shape, ksize, stride, padding)
    try:
        while True:
            next(gen)
    except StopIteration as done:
        return done.value


_PENDING_COUNTS = []     # (device count, host value it was assumed to have): verified at the next pyramid build


_CHECK_STREAMS = {}


def verify_pending_counts():
    """compare the output row counts the device found with the host values the rulebooks were sized by
    (``level_voxel_counts``); one small D2H copy for all levels of the build, on a stream of its own that waits for the
    event recorded behind the last count only (not for the convolutions queued since).  Called right after the UNet
    forward pass has been issued -- the same pass that uses the tables, so a wrong hint raises before the loss, the
    backward pass or an inference result is used -- and again at the next build (a no-op then).
    A mismatch means the hint did not belong to the coordinates: the tables built from it are invalid."""
    if not _PENDING_COUNTS:
        return
    pend = list(_PENDING_COUNTS)
    del _PENDING_COUNTS[:]
    dev = pend[0][0].device
    key = (dev.type, dev.index)
    chk = _CHECK_STREAMS.get(key)
    if chk is None:
        chk = _CHECK_STREAMS[key] = torch.cuda.Stream(device=dev)
    chk.wait_event(pend[-1][2])          # recorded on the building stream behind the last count (stream order)
    # the error words of the device's sync blocks ride in the same copy: a bounded wait inside a launch that ran out
    # (csrc/common.h SyncSlot::err, word 19 of a slot) would otherwise be wrong numerics nobody looks at
    errs = _n.sync_err_words(dev)
    with torch.cuda.stream(chk):
        # (one cat + one copy: the counts and the error words are int32 views already)
        parts = [c.reshape(-1) for c, _, _ in pend] + [e.reshape(-1) for e in errs]
        parts = [t if t.dtype == torch.int32 else t.to(torch.int32) for t in parts]
        got = torch.cat(parts).tolist()
    for c, _, _ in pend:
        c.record_stream(chk)
    if any(int(v) != 0 for v in got[len(pend):]):
        raise _n.WsisError("a bounded wait inside a launch ran out (sync slots %s): results of that launch are invalid"
                           % (_n.sync_errors(),))
    for g, (_, want, _) in zip(got, pend):
        if int(g) != int(want):
            if int(want) == 0:
                raise _n.WsisError("%d voxel(s) of the batch carry a batch index outside [0, SparseConvTensor.batch_size): "
                                   "the tables built from them are invalid" % int(g))
            raise _n.WsisError("strided rulebook: the device found %d output voxels, the batch's level_counts said %d "
                               "(a hint that does not belong to these coordinates, or a SparseConvTensor batch_size below "
                               "the number of scenes in the batch: batch indices >= batch_size are not part of the build)"
                               % (int(g), int(want)))


def level_voxel_counts(voxel_locs, spatial_shape, n_levels, ksize=2, stride=2):
    """Active voxels of levels 1 .. n_levels-1 of a UBlock pyramid (SparseConv3d k2 s2 p0 between levels) from the
    level-0 coordinates [M, 4] (batch, x, y, z), on the HOST: output voxel = floor(coordinate / 2) where that lies
    inside the output shape (sparse_unet3d.py:170-172 with spconv's output-size rule).  A loader has the coordinates
    on the host anyway (``voxelization_idx`` runs in ``collate_fn``); with these counts in the batch the device
    rulebook build sizes its tables without reading the counts back, i.e. without stopping the host four times per
    forward pass."""
    assert ksize == 2 and stride == 2
    idx = np.asarray(voxel_locs, dtype=np.int64).reshape(-1, 4)
    shape = [int(v) for v in spatial_shape]
    counts = []
    for _ in range(int(n_levels) - 1):
        out_shape = get_conv_output_size(shape, [2] * 3, [2] * 3, [0] * 3, [1] * 3)
        oa = np.asarray(out_shape, dtype=np.int64)
        o = idx[:, 1:] >> 1
        ok = np.all(o < oa, axis=1)
        b, o = idx[ok, 0], o[ok]
        lin = np.unique(((b * oa[0] + o[:, 0]) * oa[1] + o[:, 1]) * oa[2] + o[:, 2])
        counts.append(int(lin.shape[0]))
        nxt = np.empty((lin.shape[0], 4), dtype=np.int64)
        nxt[:, 3] = lin % oa[2]
        lin = lin // oa[2]
        nxt[:, 2] = lin % oa[1]
        lin = lin // oa[1]
        nxt[:, 1] = lin % oa[0]
        nxt[:, 0] = lin // oa[0]
        idx, shape = nxt, out_shape
    return counts


def level_voxel_counts_device(voxel_locs, spatial_shape, n_levels):
    """``level_voxel_counts`` on the device: ``voxel_locs`` int64 [M, 4] (batch, x, y, z) on the GPU -> the active voxels
    of levels 1 .. n_levels-1 as ONE device tensor (a sort + neighbour compare per level; nothing is read back here:
    the caller fetches the n_levels-1 numbers with a single copy).  A row survives to level l when floor(c / 2^j) lies
    inside the level-j shape for every j <= l (an odd last plane is dropped by the k2 s2 convolution and stays dropped
    below)."""
    _n.require_cuda(voxel_locs)
    idx = voxel_locs.long()
    shape = [int(v) for v in spatial_shape]
    b, c = idx[:, 0], idx[:, 1:]
    ok = torch.ones(idx.shape[0], dtype=torch.bool, device=idx.device)
    counts = []
    for lvl in range(1, int(n_levels)):
        shape = get_conv_output_size(shape, [2] * 3, [2] * 3, [0] * 3, [1] * 3)
        o = c >> lvl
        oa = torch.tensor(shape, dtype=torch.int64, device=idx.device)
        ok = ok & (o < oa).all(1)
        lin = ((b * shape[0] + o[:, 0]) * shape[1] + o[:, 1]) * shape[2] + o[:, 2]
        lin = torch.where(ok, lin, torch.full_like(lin, -1))       # dropped rows: one extra key (-1)
        k = torch.sort(lin).values
        n_keys = (k[1:] != k[:-1]).sum() + 1 if k.numel() else torch.zeros((), dtype=torch.int64, device=idx.device)
        counts.append(n_keys - (~ok).any().long())
    return torch.stack(counts) if counts else torch.zeros(0, dtype=torch.int64, device=idx.device)


def _down_rulebook_gen(indices, spatial_shape, ksize, stride, padding, deferred=None, m_out=None):
    """generator form: yields once, right before the host reads the output row count (the one sync of the level),
    so that a caller can do other work while the candidate / sort / unique kernels run (RulebookPipeline).
    ``m_out``: the count known on the host (level_voxel_counts): no read, no yield; the device count is kept for
    ``verify_pending_counts``."""
    _n.require_cuda(indices)
    _check_indices(indices)
    lib = _n.hip()
    dev = indices.device
    M_in = indices.shape[0]
    K = int(np.prod(ksize))
    out_shape = get_conv_output_size(list(spatial_shape), ksize, stride, padding, [1, 1, 1])
    k3, s3, p3 = _n.i32x3(ksize), _n.i32x3(stride), _n.i32x3(padding)
    in3, out3 = _n.i32x3(spatial_shape), _n.i32x3(out_shape)
    st = _n.stream_ptr()
    n_cand = lib.wsis_rulebook_down_ncand(M_in, k3, s3, p3)
    ws_bytes = lib.wsis_rulebook_down_workspace_bytes(n_cand)
    if ws_bytes < 0:
        raise _n.WsisError("rulebook_down workspace query failed")
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    cand = torch.empty(max(n_cand, 1), dtype=torch.int64, device=dev)
    out_keys = torch.empty(max(n_cand, 1), dtype=torch.int64, device=dev)
    count = torch.zeros(1, dtype=torch.int32, device=dev)
    _n.check(lib.wsis_rulebook_down_keys(_n.ptr(indices), M_in, in3, out3, k3, s3, p3, _n.ptr(cand),
                                         _n.ptr(out_keys), _n.ptr(count), _n.ptr(ws), ws_bytes, st),
             "rulebook_down_keys")
    if m_out is not None:
        M_out = int(m_out)
        ev = torch.cuda.Event()
        ev.record()                      # behind the kernel that wrote ``count`` on the building stream
        _PENDING_COUNTS.append((count, M_out, ev))
    else:
        yield
        M_out = int(count.item())   # the one host sync per level (upstream has the same one)
    out_indices = torch.empty((M_out, 4), dtype=torch.int32, device=dev)
    cap = _pow2_cap(M_out)
    keys = torch.empty(cap, dtype=torch.int64, device=dev)
    vals = torch.empty(cap, dtype=torch.int32, device=dev)
    rb = Rulebook("down", ksize, stride, padding, indices, out_indices, list(spatial_shape), out_shape)
    rb.nbr = torch.empty((K, M_out), dtype=torch.int32, device=dev)
    rb.nbr_up = torch.empty((K, M_in), dtype=torch.int32, device=dev)
    use_mask = K <= 32
    mask_down = torch.empty(M_out, dtype=torch.int32, device=dev) if use_mask else None
    mask_up = torch.empty(M_in, dtype=torch.int32, device=dev) if use_mask else None
    _n.check(lib.wsis_rulebook_down_fill(_n.ptr(indices), M_in, in3, out3, k3, s3, p3, _n.ptr(out_keys), M_out,
                                         _n.ptr(out_indices), _n.ptr(keys), _n.ptr(vals), cap, _n.ptr(rb.nbr),
                                         _n.ptr(rb.nbr_up), _n.ptr(mask_down), _n.ptr(mask_up), st),
             "rulebook_down_fill")
    rb.out_hash = (keys, vals, cap)
    if deferred is not None and use_mask and _use_mask_order():
        deferred.append((rb, "order", out_indices, mask_down))
        deferred.append((rb, "order_up", indices, mask_up))
        return rb
    if use_mask and _use_mask_order():
        if M_out > 0:
            rb.order = _tile_order(out_indices, mask_down)
        if M_in > 0:
            rb.order_up = _tile_order(indices, mask_up)
    rb.pack()
    return rb


class KernelProfiler(object):
    """Live per-launch timing of the conv kernels for bench.py's roofline.  The durations come from HIP events
    that libwsis_hip.so records on the launch stream directly around each forward / dIn product and each
    weight-gradient product -- the main kernel AND the fixed-order slab sum that finishes it where there is one
    (wsis_prof_enable / wsis_prof_records); the algorithmic bytes per launch follow SURVEY.md 8d:
    P*(Cin+Cout)*4 + P*8, flops 2*P*Cin*Cout, with P = number of rulebook pairs of the launch (entries >= 0 of
    its gather table; M rows for the dense 1x1 case).  ``summary()["..."]["per_launch"]`` lists every product in
    issue order: (rows, Cin, Cout, K-table-or-dense, bytes, flops, main-kernel ms, ms with the finishing launch)."""

    NAMES = ("spconv_fwd_kernel", "spconv_dw_kernel", "bn_op")     # (bn_op: the BatchNorm ops of the native executor)

    def __init__(self):
        self.acc = {n: [0, 0, 0] for n in self.NAMES}   # launches, bytes, flops
        self.log = {n: [] for n in self.NAMES}          # per product: (rows, Cin, Cout, pairs, bytes, flops)
        self._pairs = {}
        _n.check(_n.hip().wsis_prof_enable(1), "prof_enable")

    def pairs(self, nbr, M_out):
        if nbr is None:
            return int(M_out)
        key = (nbr.data_ptr(), tuple(nbr.shape))
        if key not in self._pairs:
            self._pairs[key] = int((nbr >= 0).sum().item())
        return self._pairs[key]

    def begin(self):
        return None

    def end(self, name, start, nbytes, flops, shape=None):
        a = self.acc[name]
        a[0] += 1
        a[1] += nbytes
        a[2] += flops
        self.log[name].append((shape or (0, 0, 0, 0)) + (nbytes, flops))

    def summary(self):
        import ctypes
        torch.cuda.synchronize()
        lib = _n.hip()
        _n.check(lib.wsis_prof_enable(0), "prof_enable")
        out = {}
        for which, name in enumerate(self.NAMES):
            launches, nbytes, flops = self.acc[name]
            cap = max(launches, 1) if name != "bn_op" else max(launches, 8192)
            main = (ctypes.c_double * cap)()
            total = (ctypes.c_double * cap)()
            n = ctypes.c_int64(0)
            _n.check(lib.wsis_prof_records(which, ctypes.addressof(main), ctypes.addressof(total), cap,
                                           ctypes.addressof(n)), "prof_records")
            if name == "bn_op" and n.value != launches:
                continue      # BatchNorm layers that ran outside the executor's ops (SyncBatchNorm parts, module walk)
            assert n.value == launches, (name, n.value, launches)
            per = [rec + (main[i], total[i]) for i, rec in enumerate(self.log[name])]
            out[name] = {"launches": launches, "ms": float(sum(total[i] for i in range(launches))),
                         "ms_main": float(sum(main[i] for i in range(launches))), "bytes": nbytes, "flops": flops,
                         "per_launch": per}
        return out


PROFILER = None


_WS_CACHE = {}


def _fwd_ws_bytes(lib, M_out, K, Cin, Cout):
    key = (M_out, K, Cin, Cout)
    v = _WS_CACHE.get(key)
    if v is None:
        v = lib.wsis_spconv_fwd_workspace_bytes(M_out, K, Cin, Cout)
        if len(_WS_CACHE) > 4096:
            _WS_CACHE.clear()
        _WS_CACHE[key] = v
    return v


def _conv(X, nbr, order, W, bias, residual, M_out):
    """out[r] = sum_k X[nbr[k][r]] @ W[k]  (W [K,Cin,Cout] contiguous)."""
    K, Cin, Cout = W.shape
    out = torch.empty((M_out, Cout), dtype=torch.float32, device=X.device)
    prof = PROFILER
    if prof is not None:
        P = prof.pairs(nbr, M_out)
        t0 = prof.begin()
    lib = _n.hip()
    ws_bytes = _fwd_ws_bytes(lib, M_out, K, Cin, Cout)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=X.device) if ws_bytes > 256 else None
    _n.check(lib.wsis_spconv_fwd(_n.ptr(X), _n.ptr(nbr), _n.ptr(order), _n.ptr(W), _n.ptr(bias),
                                 _n.ptr(residual), _n.ptr(out), X.shape[0], M_out, K, Cin, Cout, _n.ptr(ws),
                                 ws_bytes, _n.stream_ptr()), "spconv_fwd")
    if prof is not None:
        prof.end("spconv_fwd_kernel", t0, P * (Cin + Cout) * 4 + P * 8, 2 * P * Cin * Cout, (int(M_out), Cin, Cout, P))
    return out


def _use_fwd2(K, Cin, Cout, rows=0):
    """the wave-autonomous LDS-DMA kernel (csrc/spconv2.hip) where it applies (channel multiples of 32, K <= 32, gathered
    tensor of `rows` x Cin floats below 2 GiB: 32-bit gather offsets); WSIS_FWD2=0 keeps the round-1 kernel"""
    return (os.environ.get("WSIS_FWD2", "1") != "0" and bool(_n.hip().wsis_spconv_fwd_t_supported(K, Cin, Cout))
            and rows * Cin * 4 < (1 << 31))


_WS2_CACHE = {}


def _conv_t(X, nbr, order, WT, flip, bias, residual, M_out, stats=None):
    """out[r] = sum_k X[nbr[k][r]] @ W[k] with the weights as B^T: WT [K,Cout,Cin] (slice K-1-k when ``flip``).
    ``stats``: optional fp32 [ceil(M_out/32), 2, Cout] buffer that receives the BatchNorm partials of the output."""
    K, Cout, Cin = WT.shape
    out = torch.empty((M_out, Cout), dtype=torch.float32, device=X.device)
    prof = PROFILER
    if prof is not None:
        P = prof.pairs(nbr, M_out)
    lib = _n.hip()
    key = (M_out, K, Cin, Cout)
    ws_bytes = _WS2_CACHE.get(key)
    if ws_bytes is None:
        ws_bytes = lib.wsis_spconv_fwd_t_workspace_bytes(M_out, K, Cin, Cout)
        if len(_WS2_CACHE) > 4096:
            _WS2_CACHE.clear()
        _WS2_CACHE[key] = ws_bytes
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=X.device) if ws_bytes > 256 else None
    _n.check(lib.wsis_spconv_fwd_t(_n.ptr(X), _n.ptr(nbr), _n.ptr(order), _n.ptr(WT), int(flip), _n.ptr(bias),
                                   _n.ptr(residual), _n.ptr(out), _n.ptr(stats), X.shape[0], M_out, K, Cin, Cout, _n.ptr(ws),
                                   ws_bytes, _n.ptr(_n.sync_block(X.device)), _n.stream_ptr()), "spconv_fwd_t")
    if prof is not None:
        prof.end("spconv_fwd_kernel", None, P * (Cin + Cout) * 4 + P * 8, 2 * P * Cin * Cout,
                 (int(M_out), Cin, Cout, P))
    return out


def _weight_t(W, flip):
    K, Cin, Cout = W.shape
    WT = torch.empty((K, Cout, Cin), dtype=torch.float32, device=W.device)
    _n.check(_n.hip().wsis_weight_transpose(_n.ptr(W), _n.ptr(WT), K, Cin, Cout, int(flip), _n.stream_ptr()),
             "weight_transpose")
    return WT


def _dw(X, nbr, order, dY, K, Cin, Cout):
    lib = _n.hip()
    M_out = dY.shape[0]
    ws_bytes = lib.wsis_spconv_dw_workspace_bytes(M_out, K, Cin, Cout)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=X.device)
    dW = torch.empty((K, Cin, Cout), dtype=torch.float32, device=X.device)
    prof = PROFILER
    if prof is not None:
        P = prof.pairs(nbr, M_out)
        t0 = prof.begin()
    _n.check(lib.wsis_spconv_dw(_n.ptr(X), _n.ptr(nbr), _n.ptr(order), _n.ptr(dY), _n.ptr(dW), X.shape[0], M_out,
                                K, Cin, Cout, _n.ptr(ws), ws_bytes, _n.stream_ptr()), "spconv_dw")
    if prof is not None:
        prof.end("spconv_dw_kernel", t0, P * (Cin + Cout) * 4 + P * 8, 2 * P * Cin * Cout, (int(M_out), Cin, Cout, P))
    return dW


class SparseConvFunction(Function):
    """features [M_in,Cin], weight [k0,k1,k2,Cin,Cout] -> [M_out,Cout].

    nbr_f/order_f: PACKED gather table + tile order of the forward pass (rows = outputs);
    nbr_b/order_b: packed gather table + tile order of the dIn pass (rows = inputs);
    flip: subm dIn uses W[K-1-k]^T."""

    @staticmethod
    def forward(ctx, features, weight, bias, nbr_f, order_f, nbr_b, order_b, flip, M_out):
        _n.require_cuda(features, weight)
        X = features if (features.dtype == torch.float32 and features.is_contiguous()) else features.contiguous().float()
        Cin, Cout = weight.shape[-2], weight.shape[-1]
        W = weight.detach()
        if W.dtype != torch.float32 or not W.is_contiguous():
            W = W.contiguous().float()
        W = W.view(-1, Cin, Cout)
        b = bias.contiguous().float() if bias is not None else None
        if _use_fwd2(W.shape[0], Cin, Cout, X.shape[0]):
            out = _conv_t(X, nbr_f, order_f, _weight_t(W, 0), 0, b, None, M_out)
        else:
            out = _conv(X, nbr_f, order_f, W, b, None, M_out)
        ctx.save_for_backward(X, W)
        ctx.aux = (nbr_f, order_f, nbr_b, order_b, flip, weight.shape, bias is not None)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        X, W = ctx.saved_tensors
        nbr_f, order_f, nbr_b, order_b, flip, wshape, has_bias = ctx.aux
        dY = grad_out if (grad_out.dtype == torch.float32 and grad_out.is_contiguous()) else grad_out.contiguous().float()
        K, Cin, Cout = W.shape
        dX = dW = db = None
        if ctx.needs_input_grad[0]:
            if _use_fwd2(K, Cout, Cin, dY.shape[0]):    # dIn: the weight itself is the B^T operand ([K, Cin, Cout]: rows = dIn outputs)
                dX = _conv_t(dY, nbr_b, order_b, W, flip, None, None, X.shape[0])
            else:
                WT = _weight_t(W, flip)
                dX = _conv(dY, nbr_b, order_b, WT, None, None, X.shape[0])
        if ctx.needs_input_grad[1]:
            dW = _dw(X, nbr_f, order_f, dY, K, Cin, Cout).view(wshape)
        if has_bias and ctx.needs_input_grad[2]:
            db = dY.sum(0)
        return dX, dW, db, None, None, None, None, None, None


def sparse_conv(features, weight, bias, nbr_f, order_f, nbr_b, order_b, flip, M_out):
    return SparseConvFunction.apply(features, weight, bias, nbr_f, order_f, nbr_b, order_b, flip, M_out)


_SIDE_STREAMS = {}


def _side_stream(device):
    key = (device.type, device.index)
    st = _SIDE_STREAMS.get(key)
    if st is None:
        st = torch.cuda.Stream(device=device)
        _SIDE_STREAMS[key] = st
    return st


def _rulebook_tensors(rb):
    out = [rb.nbr, rb.nbr_up, rb.order, rb.order_up, rb.nbr_p, rb.nbr_up_p, rb.out_indices]
    if rb.out_hash is not None:
        out += [rb.out_hash[0], rb.out_hash[1]]
    return [t for t in out if t is not None]


def _build_pyramid(tensor, n_levels, subm_key, down_key, first_id):
    """the rulebook chain itself (current stream); returns the tensors it allocated"""
    gen = _build_pyramid_gen(tensor, n_levels, subm_key, down_key, first_id)
    try:
        while True:
            next(gen)
    except StopIteration as done:
        return done.value


def _build_pyramid_gen(tensor, n_levels, subm_key, down_key, first_id):
    """generator form of the chain: yields before every host sync (one per strided level)"""
    built = []
    indices, shape = tensor.indices, [int(s) for s in tensor.spatial_shape]
    hash_tab = tensor._hash
    batch_size = int(getattr(tensor, "batch_size", 0) or 0)
    deferred = [] if _tile_batch_enabled(batch_size) else None
    new_rbs = []
    hints = getattr(tensor, "_level_counts", None)
    if hints is not None and (len(hints) != n_levels - 1 or os.environ.get("WSIS_LEVEL_COUNTS", "1") == "0"):
        hints = None
    if batch_size > 0 and indices.is_cuda and indices.shape[0] > 0:
        # batch indices outside [0, batch_size) are a caller error upstream (they index past its dense grid).  The
        # native pyramid drops such rows and its count check raises; this walk must not build tables over them either:
        # with host counts the check rides the pending-count read of this pass, without them the chain stops the
        # host per level anyway and the check is one more small read.
        b = indices[:, 0]
        bad = ((b < 0) | (b >= batch_size)).sum(dtype=torch.int32).reshape(1)
        if hints is None:
            if int(bad) != 0:
                raise _n.WsisError("SparseConvTensor.batch_size = %d but %d voxel(s) carry a batch index outside "
                                   "[0, batch_size)" % (batch_size, int(bad)))
        else:
            ev = torch.cuda.Event()
            ev.record()
            _PENDING_COUNTS.append((bad, 0, ev))
    for lvl in range(n_levels):
        kid = first_id + lvl
        key = subm_key.format(kid)
        if key not in tensor.indice_dict:
            if hash_tab is None:
                hash_tab = build_hash(indices, shape)
                if lvl == 0:
                    tensor._hash = hash_tab
                    built += [hash_tab[0], hash_tab[1]]
            rb = build_subm_rulebook(indices, shape, [3, 3, 3], [1, 1, 1], hash_tab, deferred=deferred)
            tensor.indice_dict[key] = rb
            new_rbs.append(rb)
        if lvl + 1 < n_levels:
            dkey = down_key.format(kid)
            rb = tensor.indice_dict.get(dkey)
            if rb is None:
                rb = yield from _down_rulebook_gen(indices, shape, [2, 2, 2], [2, 2, 2], [0, 0, 0], deferred=deferred,
                                                   m_out=hints[lvl] if hints is not None else None)
                tensor.indice_dict[dkey] = rb
                new_rbs.append(rb)
            indices, shape, hash_tab = rb.out_indices, rb.out_shape, rb.out_hash
    if deferred:
        finish_tile_orders(deferred, batch_size)
    for rb in new_rbs:
        built += _rulebook_tensors(rb)
    return built


# field numbers of wsis_rulebook_pyramid_layout (include/wsis_hip.h WSIS_PYR_*)
(PYR_ROWS, PYR_CAP, PYR_INDICES, PYR_KEYS, PYR_VALS, PYR_SUBM_NBR, PYR_SUBM_MASK, PYR_SUBM_ORDER, PYR_SUBM_NBR_P, PYR_CAND,
 PYR_OUT_KEYS, PYR_COUNT, PYR_DOWN_WS, PYR_DOWN_NBR, PYR_UP_NBR, PYR_DOWN_MASK, PYR_UP_MASK, PYR_DOWN_ORDER, PYR_UP_ORDER,
 PYR_DOWN_NBR_P, PYR_UP_NBR_P, PYR_FIELDS) = range(22)


class Pyramid(object):
    """All rulebooks of a UBlock pyramid in ONE device arena, built by one native call (wsis_rulebook_pyramid) from the
    batch's host-side level counts.  ``ptr(level, field)`` gives device pointers without creating tensors (what the
    op-list executor needs); ``rulebook(key)`` materialises the ``Rulebook`` views of one table on demand (module walk,
    tests, the profiler's pair counts)."""

    def __init__(self, arena, layout, indices0, shapes, n_levels, subm_key, down_key, first_id):
        self.arena, self.layout, self.indices0, self.shapes, self.n_levels = arena, layout, indices0, shapes, n_levels
        self.base = arena.data_ptr()
        self.keys = {}
        for l in range(n_levels):
            self.keys[subm_key.format(first_id + l)] = ("subm", l)
            if l + 1 < n_levels:
                self.keys[down_key.format(first_id + l)] = ("down", l)
        self._made = {}

    def rows(self, level):
        return int(self.layout[level, PYR_ROWS])

    def ptr(self, level, field):
        off = int(self.layout[level, field])
        return 0 if off < 0 else self.base + off

    def view(self, level, field, dtype, shape):
        off = int(self.layout[level, field])
        n = int(np.prod(shape)) * torch.empty(0, dtype=dtype).element_size()
        return self.arena[off:off + n].view(dtype).view(shape)

    def indices(self, level):
        return self.indices0 if level == 0 else self.view(level, PYR_INDICES, torch.int32, (self.rows(level), 4))

    def hash_tab(self, level):
        cap = int(self.layout[level, PYR_CAP])
        return (self.view(level, PYR_KEYS, torch.int64, (cap,)), self.view(level, PYR_VALS, torch.int32, (cap,)), cap)

    def rulebook(self, key):
        rb = self._made.get(key)
        if rb is not None:
            return rb
        kind, l = self.keys[key]
        M = self.rows(l)
        if kind == "subm":
            rb = Rulebook("subm", [3, 3, 3], [1, 1, 1], [1, 1, 1], self.indices(l), self.indices(l), list(self.shapes[l]),
                          list(self.shapes[l]))
            rb.nbr = self.view(l, PYR_SUBM_NBR, torch.int32, (27, M))
            rb.order = self.view(l, PYR_SUBM_ORDER, torch.int32, (M,))
            rb.nbr_p = self.view(l, PYR_SUBM_NBR_P, torch.int32, (27, M))
            rb.out_hash = self.hash_tab(l)
        else:
            Mo = self.rows(l + 1)
            rb = Rulebook("down", [2, 2, 2], [2, 2, 2], [0, 0, 0], self.indices(l), self.indices(l + 1),
                          list(self.shapes[l]), list(self.shapes[l + 1]))
            rb.nbr = self.view(l, PYR_DOWN_NBR, torch.int32, (8, Mo))
            rb.nbr_up = self.view(l, PYR_UP_NBR, torch.int32, (8, M))
            rb.order = self.view(l, PYR_DOWN_ORDER, torch.int32, (Mo,))
            rb.order_up = self.view(l, PYR_UP_ORDER, torch.int32, (M,))
            rb.nbr_p = self.view(l, PYR_DOWN_NBR_P, torch.int32, (8, Mo))
            rb.nbr_up_p = self.view(l, PYR_UP_NBR_P, torch.int32, (8, M))
            rb.out_hash = self.hash_tab(l + 1)
        self._made[key] = rb
        return rb


class PyramidDict(dict):
    """``indice_dict`` of a tensor whose rulebooks live in a ``Pyramid``: the keys are there at once, the ``Rulebook``
    objects are made when somebody asks for one"""

    def __init__(self, pyramid):
        super().__init__()
        self.pyramid = pyramid
        for k in pyramid.keys:
            dict.__setitem__(self, k, None)

    def __getitem__(self, k):
        v = dict.__getitem__(self, k)
        if v is None and k in self.pyramid.keys:
            v = self.pyramid.rulebook(k)
            dict.__setitem__(self, k, v)
        return v

    def get(self, k, default=None):
        return self[k] if k in self else default

    def values(self):
        return [self[k] for k in self]

    def items(self):
        return [(k, self[k]) for k in self]


def _pyramid_native_ok(tensor, n_levels, keys):
    hints = getattr(tensor, "_level_counts", None)
    return (os.environ.get("WSIS_PYRAMID_NATIVE", "1") != "0" and os.environ.get("WSIS_LEVEL_COUNTS", "1") != "0"
            and hints is not None and len(hints) == n_levels - 1 and all(int(h) > 0 for h in hints)
            and tensor.indices.shape[0] > 0 and not any(k in tensor.indice_dict for k in keys)
            and getattr(tensor, "_hash", None) is None and _tile_batch_enabled(int(getattr(tensor, "batch_size", 0) or 0))
            and 3 * n_levels - 2 <= 16)


def _build_pyramid_native(tensor, n_levels, subm_key, down_key, first_id):
    """wsis_rulebook_pyramid on the current stream -> Pyramid (and the tensors it allocated)"""
    import ctypes
    lib = _n.hip()
    indices = tensor.indices
    _check_indices(indices)
    M0 = int(indices.shape[0])
    hints = [int(h) for h in tensor._level_counts]
    h_counts = (ctypes.c_int64 * max(len(hints), 1))(*hints)
    layout = np.zeros((n_levels, PYR_FIELDS), dtype=np.int64)
    total = lib.wsis_rulebook_pyramid_layout(M0, h_counts, n_levels, layout.ctypes.data)
    if total < 0:
        raise _n.WsisError("rulebook pyramid layout query failed")
    arena = torch.empty(total + 256, dtype=torch.uint8, device=indices.device)
    skip = (-arena.data_ptr()) % 256
    arena = arena[skip:skip + total]
    shapes = [[int(s) for s in tensor.spatial_shape]]
    for _ in range(n_levels - 1):
        shapes.append(get_conv_output_size(shapes[-1], [2] * 3, [2] * 3, [0] * 3, [1] * 3))
    _n.check(lib.wsis_rulebook_pyramid(_n.ptr(indices), M0, _n.i32x3(shapes[0]), h_counts, n_levels,
                                       int(tensor.batch_size), _tile_block_shift(), _n.ptr(arena), total,
                                       _n.stream_ptr()), "rulebook_pyramid")
    pyr = Pyramid(arena, layout, indices, shapes, n_levels, subm_key, down_key, first_id)
    ev = torch.cuda.Event()
    ev.record()                          # behind every count of the chain on the building stream
    for l in range(n_levels - 1):
        _PENDING_COUNTS.append((pyr.view(l, PYR_COUNT, torch.int32, (1,)), hints[l], ev))
    return pyr


def hint_batch_rows(rows):
    """``wsis_hint_batch_rows``: the weight-gradient launch plan follows the rows of the batch being trained.  The hint
    is ONE word per process (all devices, streams and the weight-gradient worker thread read it) and stays until the
    next call: every forward pass of the model sets it (``prebuild_unet_rulebooks`` or, with ``WSIS_PREBUILD=0``,
    ``Network.forward``); a caller that drives ``wsis_spconv_dw`` itself sets it itself (0 = no hint) before runs whose
    bits it compares."""
    _n.check(_n.hip().wsis_hint_batch_rows(int(rows)), "hint_batch_rows")


def prebuild_unet_rulebooks(tensor, n_levels, subm_key="subm{}", down_key="spconv{}", first_id=1, side_stream=None):
    """Builds every rulebook of a UBlock pyramid (SubM k3 p1 per level, k2 s2 between levels) up front and
    stores them in ``tensor.indice_dict`` under the keys the modules will look up.

    The rulebook chain only depends on the coordinates.  It runs on a SIDE stream: its per-level host syncs
    (output row counts) then wait for that stream only, while the main stream keeps executing whatever is still
    queued (the previous step's backward / optimizer), so the pipeline is never drained; the main stream joins
    the side stream before the first convolution.  ``tensor.indices`` must be complete before the call
    (``tensor._ready_event``, if set, is waited for on the side stream)."""
    keys = [subm_key.format(first_id + l) for l in range(n_levels)] + \
           [down_key.format(first_id + l) for l in range(n_levels - 1)]
    if tensor.indices.is_cuda:
        # launch-plan hint for this batch's weight-gradient products (wsis_hip.h: wsis_hint_batch_rows): every path that
        # trains on the tensor -- the native executor and the module walk -- passes here with the same row count
        hint_batch_rows(int(tensor.indices.shape[0]))
    if all(k in tensor.indice_dict for k in keys):
        return                   # e.g. attached from a RulebookPrefetcher
    if side_stream is None:
        side_stream = os.environ.get("WSIS_RULEBOOK_STREAM", "1") != "0"
    dev = tensor.indices.device
    main = torch.cuda.current_stream(dev)
    side = _side_stream(dev) if side_stream else None
    ctx = torch.cuda.stream(side) if side is not None else _NullCtx()
    if side is not None:
        # the side stream must see complete coordinates: an explicit ``_ready_event`` (recorded right after the
        # producer of ``indices``, lets the build overlap whatever else is queued on the main stream) or, without
        # one, everything enqueued on the current stream so far (a drop-in caller that built the tensor the
        # reference's way: ``SparseConvTensor(feats, coords.int(), ...)`` on the main stream)
        ev = getattr(tensor, "_ready_event", None)
        if ev is not None:
            side.wait_event(ev)
        else:
            side.wait_stream(main)
    with ctx:
        verify_pending_counts()      # counts of the PREVIOUS build that was sized from host values (long finished)
        if _pyramid_native_ok(tensor, n_levels, keys):
            # every level's row count is known on the host: the whole chain is ONE native call into one arena
            pyr = _build_pyramid_native(tensor, n_levels, subm_key, down_key, first_id)
            tensor.indice_dict = PyramidDict(pyr)
            built = [pyr.arena]
        else:
            built = _build_pyramid(tensor, n_levels, subm_key, down_key, first_id)
    if side is not None:
        main.wait_stream(side)
        for t in built:          # allocated on the side stream, consumed on the main stream
            t.record_stream(main)


class RulebookSet(object):
    """Rulebooks of one batch built ahead of its forward pass (see RulebookPrefetcher): the indice_dict / hash a
    SparseConvTensor of the same coordinates adopts, plus the event that orders them before their consumers."""

    def __init__(self, indices, spatial_shape):
        self.indices = indices
        self.spatial_shape = spatial_shape
        self.indice_dict = {}
        self._hash = None
        self.done = None       # recorded on the building stream after the last rulebook kernel
        self.tensors = []

    def attach(self, tensor):
        """hand the rulebooks to ``tensor`` (same indices); the current stream waits for the build"""
        main = torch.cuda.current_stream(tensor.indices.device)
        if self.done is not None:
            main.wait_event(self.done)
        for t in self.tensors:   # allocated on the building stream, consumed on this one
            t.record_stream(main)
        tensor.indice_dict.update(self.indice_dict)
        tensor._hash = self._hash


class RulebookPipeline(object):
    """Builds the rulebook pyramid of the NEXT batch on the side stream in slices, from the issuing thread itself:
    ``start`` issues the first slice, every ``pump`` (called between the phases of the current step: after the
    forward, after the backward, after the optimizer) reads one output row count -- long computed by then, so the
    host does not block -- and issues the next slice, ``finish`` hands out the RulebookSet.  The chain's ~60 small
    kernels then run next to the current step's kernels instead of in front of the next forward pass, where the
    main stream had to wait for them (~1.4 ms per step), and nothing needs a helper thread.

        pipe.start(coords_i+1, shape, ready_event);  forward_i;  pipe.pump();  backward_i;  pipe.pump(); ...
        rulebooks_i+1 = pipe.finish()
    """

    def __init__(self, n_levels, subm_key="subm{}", down_key="spconv{}", first_id=1):
        self.args = (n_levels, subm_key, down_key, first_id)
        self._gen = None
        self._rs = None

    def _advance(self):
        """one slice on the side stream; returns False when the chain is complete"""
        rs = self._rs
        side = _side_stream(rs.indices.device)
        with torch.cuda.stream(side):
            try:
                next(self._gen)
                return True
            except StopIteration as done:
                rs.tensors = done.value
                rs.done = torch.cuda.Event()
                rs.done.record(side)
                self._gen = None
                return False

    def start(self, indices, spatial_shape, ready_event=None):
        assert self._gen is None and self._rs is None, "one batch in flight at a time"
        _check_indices(indices)
        self._rs = RulebookSet(indices, spatial_shape)
        if ready_event is None:     # no explicit event: order behind everything queued on the current stream
            ready_event = torch.cuda.Event()
            ready_event.record(torch.cuda.current_stream(indices.device))
        _side_stream(indices.device).wait_event(ready_event)
        self._gen = _build_pyramid_gen(self._rs, *self.args)
        self._advance()

    def pump(self):
        if self._gen is not None:
            self._advance()

    def finish(self):
        assert self._rs is not None, "nothing started"
        while self._gen is not None:
            self._advance()
        rs, self._rs = self._rs, None
        return rs


class RulebookPrefetcher(object):
    """Builds the rulebook pyramid of the NEXT batch on a side stream from a background thread while the current
    step is still being issued -- the data-loader stage of the sparse-conv path.  The per-level host syncs of the
    chain (output row counts of the strided levels) then block only the helper thread (torch releases the GIL
    while it waits), neither the issuing thread nor the main stream.

        pre = RulebookPrefetcher(n_levels)
        pre.submit(coords_int32, spatial_shape, ready_event)      # batch i+1
        ... train_step(batch i) ...
        rulebooks = pre.result()                                  # RulebookSet for batch i+1
    """

    def __init__(self, n_levels, subm_key="subm{}", down_key="spconv{}", first_id=1):
        self.args = (n_levels, subm_key, down_key, first_id)
        self._thread = None
        self._out = None
        self._err = None

    def _work(self, rs, ready_event, device):
        try:
            torch.cuda.set_device(device)
            side = _side_stream(device)
            if ready_event is not None:
                side.wait_event(ready_event)
            with torch.cuda.stream(side):
                rs.tensors = _build_pyramid(rs, *self.args)
                rs.done = torch.cuda.Event()
                rs.done.record(side)
            self._out = rs
        except BaseException as e:   # surfaced by result()
            self._err = e

    def submit(self, indices, spatial_shape, ready_event=None):
        import threading
        assert self._thread is None, "one batch in flight at a time"
        _check_indices(indices)
        rs = RulebookSet(indices, spatial_shape)
        if ready_event is None:     # recorded HERE, on the submitting thread's current stream (the producer's)
            ready_event = torch.cuda.Event()
            ready_event.record(torch.cuda.current_stream(indices.device))
        self._out = self._err = None
        self._thread = threading.Thread(target=self._work, args=(rs, ready_event, indices.device), daemon=True)
        self._thread.start()

    def result(self):
        assert self._thread is not None, "nothing submitted"
        self._thread.join()
        self._thread = None
        if self._err is not None:
            raise self._err
        return self._out


class _NullCtx(object):
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False
