"""Drop-in for ``import pointgroup_ops`` (PointGroup ``lib/pointgroup_ops`` [UPSTREAM]).

Same call signatures as the reference's call sites:
  voxelization_idx  modules/datasets/scannetv2_dataset.py:449,528  test_scannetv2.py:389
  voxelization      train_scannetv2.py:189  test_scannetv2.py:182
  ballquery_batch_p / bfs_cluster   (exported upstream, named by BASELINE.json north_star)

voxelization_idx / bfs_cluster are host operators (libwsis_host.so, CPU tensors in and out, exactly
like upstream); voxelization / ballquery_batch_p are HIP kernels (libwsis_hip.so).  Extension: voxelization_idx
also accepts a CUDA LongTensor and then runs on the device with the same first-occurrence contract.
"""
import torch
from torch.autograd import Function

import wsis_native as _n

__all__ = ["voxelization_idx", "voxelization", "ballquery_batch_p", "bfs_cluster",
           "Voxelization_Idx", "Voxelization", "BallQueryBatchP", "BFSCluster"]


def _voxelization_idx_gpu(coords):
    lib = _n.hip()
    dev = coords.device
    N = coords.size(0)
    ws_bytes = lib.wsis_voxelize_idx_workspace_bytes(N)
    if ws_bytes < 0:
        raise _n.WsisError("voxelize_idx workspace query failed")
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    p2v = torch.empty(N, dtype=torch.int32, device=dev)
    counts = torch.zeros(2, dtype=torch.int32, device=dev)
    st = _n.stream_ptr()
    _n.check(lib.wsis_voxelize_idx_map(_n.ptr(coords), N, _n.ptr(p2v), _n.ptr(counts), _n.ptr(ws), ws_bytes, st),
             "voxelize_idx_map")
    M, ma = (int(v) for v in counts.tolist())
    locs = torch.empty((M, 4), dtype=torch.int64, device=dev)
    v2p = torch.empty((M, ma + 1), dtype=torch.int32, device=dev)
    _n.check(lib.wsis_voxelize_idx_fill(_n.ptr(coords), N, M, ma, _n.ptr(locs), _n.ptr(v2p), _n.ptr(ws), ws_bytes, st),
             "voxelize_idx_fill")
    return locs, p2v, v2p


class Voxelization_Idx(Function):
    @staticmethod
    def forward(ctx, coords, batchsize, mode=4):
        """coords: LongTensor [N,4] (batch, x, y, z), CPU, contiguous.
        returns (output_coords Long [M,4], input_map Int [N], output_map Int [M, maxActive+1])"""
        assert coords.dtype == torch.int64 and coords.dim() == 2 and coords.size(1) == 4
        if mode != 4:
            raise NotImplementedError("only mode=4 (mean) is used by 3D-WSIS (config data.mode: 4)")
        if coords.is_cuda:
            # extension: device tensors in -> device tensors out, same first-occurrence contract
            return _voxelization_idx_gpu(coords.contiguous())
        assert coords.is_contiguous()
        N = coords.size(0)
        lib = _n.host()
        input_map = torch.empty(N, dtype=torch.int32)
        import ctypes
        M = ctypes.c_int64(0)
        ma = ctypes.c_int32(0)
        _n.check_host(lib.wsis_host_voxelize_idx_map(_n.ptr(coords), N, _n.ptr(input_map),
                                                     ctypes.addressof(M), ctypes.addressof(ma)),
                      "voxelize_idx_map")
        M, ma = M.value, ma.value
        output_coords = torch.empty((M, 4), dtype=torch.int64)
        output_map = torch.empty((M, ma + 1), dtype=torch.int32)
        _n.check_host(lib.wsis_host_voxelize_idx_fill(_n.ptr(coords), N, _n.ptr(input_map), M, ma,
                                                      _n.ptr(output_coords), _n.ptr(output_map)),
                      "voxelize_idx_fill")
        return output_coords, input_map, output_map

    @staticmethod
    def backward(ctx, a=None, b=None, c=None):
        return None


voxelization_idx = Voxelization_Idx.apply


class Voxelization(Function):
    @staticmethod
    def forward(ctx, feats, map_rule, mode=4):
        """feats: cuda float [N,C]; map_rule: cuda int [M, maxActive+1]; -> cuda float [M,C]"""
        _n.require_cuda(feats, map_rule)
        assert map_rule.dtype == torch.int32 and map_rule.is_contiguous()
        feats = feats.contiguous().float()
        N, C = feats.shape
        M, stride = map_rule.shape
        out = torch.empty((M, C), dtype=torch.float32, device=feats.device)
        _n.check(_n.hip().wsis_voxelize_fwd(_n.ptr(feats), _n.ptr(map_rule), _n.ptr(out), M, C, stride,
                                            int(mode), _n.stream_ptr()), "voxelize_fwd")
        ctx.for_backwards = (map_rule, int(mode), N, C)
        return out

    @staticmethod
    def backward(ctx, d_output_feats):
        map_rule, mode, N, C = ctx.for_backwards
        M, stride = map_rule.shape
        d = d_output_feats.contiguous().float()
        d_feats = torch.zeros((N, C), dtype=torch.float32, device=d.device)
        _n.check(_n.hip().wsis_voxelize_bwd(_n.ptr(d), _n.ptr(map_rule), _n.ptr(d_feats), M, C, stride,
                                            mode, _n.stream_ptr()), "voxelize_bwd")
        return d_feats, None, None


voxelization = Voxelization.apply


class BallQueryBatchP(Function):
    @staticmethod
    def forward(ctx, coords, batch_idxs, batch_offsets, radius, meanActive):
        """coords cuda float [N,3]; batch_idxs cuda int [N]; batch_offsets cuda int [B+1]
        -> idx cuda int [nActive], start_len cuda int [N,2].
        ``meanActive`` is accepted for signature parity; the count-then-fill implementation never
        overflows, so the upstream retry loop is not needed."""
        _n.require_cuda(coords, batch_idxs, batch_offsets)
        assert coords.is_contiguous() and coords.dtype == torch.float32
        assert batch_idxs.dtype == torch.int32 and batch_offsets.dtype == torch.int32
        N = coords.size(0)
        B = batch_offsets.numel() - 1
        lib = _n.hip()
        dev = coords.device
        ws_bytes = lib.wsis_ballquery_workspace_bytes(N)
        if ws_bytes < 0:
            raise _n.WsisError("ballquery workspace query failed")
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        start_len = torch.zeros((N, 2), dtype=torch.int32, device=dev)
        total = torch.zeros(1, dtype=torch.int32, device=dev)
        st = _n.stream_ptr()
        _n.check(lib.wsis_ballquery_count(_n.ptr(coords), _n.ptr(batch_idxs), _n.ptr(batch_offsets), N, B,
                                          float(radius), _n.ptr(start_len), _n.ptr(total), _n.ptr(ws),
                                          ws_bytes, st), "ballquery_count")
        n_active = int(total.item())
        idx = torch.empty(n_active, dtype=torch.int32, device=dev)
        _n.check(lib.wsis_ballquery_fill(_n.ptr(coords), _n.ptr(batch_idxs), _n.ptr(batch_offsets), N, B,
                                         float(radius), _n.ptr(start_len), _n.ptr(idx), n_active, _n.ptr(ws),
                                         ws_bytes, st), "ballquery_fill")
        return idx, start_len

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None, None, None, None


ballquery_batch_p = BallQueryBatchP.apply


def _bfs_cluster_device(sem, idx, start_len, threshold):
    _n.require_cuda(sem, idx, start_len)
    for t in (sem, idx, start_len):
        assert t.dtype == torch.int32 and t.is_contiguous()
    lib, dev, N = _n.hip(), sem.device, sem.size(0)
    if N and int(start_len[:, 1].max()) >= 1000:
        # a list truncated at upstream's 1000-neighbour cap makes the graph asymmetric (p lists q, q does not list p);
        # the union-find components assume symmetric lists, so that (pathological) case takes the host walk
        ci, co = BFSCluster.apply(sem.cpu(), idx.cpu(), start_len.cpu(), threshold)
        return ci.to(dev), co.to(dev)
    st = _n.stream_ptr()
    parent = torch.empty(N, dtype=torch.int32, device=dev)
    root = torch.empty(N, dtype=torch.int32, device=dev)
    size = torch.empty(N, dtype=torch.int32, device=dev)
    _n.check(lib.wsis_cc_same_label(_n.ptr(sem), _n.ptr(idx), _n.ptr(start_len), N, _n.ptr(parent), _n.ptr(root),
                                    _n.ptr(size), st), "cc_same_label")
    ar = torch.arange(N, dtype=torch.int32, device=dev)
    seeds = torch.nonzero((root == ar) & (size >= threshold)).flatten().int()   # ascending = the host's seed order
    n_c = int(seeds.numel())                                                      # the one host read
    sizes = size[seeds.long()]
    offsets = torch.zeros(n_c + 1, dtype=torch.int32, device=dev)
    offsets[1:] = torch.cumsum(sizes, 0).int()
    total = int(offsets[-1].item()) if n_c else 0
    cluster_idxs = torch.empty((total, 2), dtype=torch.int32, device=dev)
    if n_c:
        pos = torch.full((N,), -1, dtype=torch.int32, device=dev)
        stamp = torch.full((N,), 2 ** 31 - 1, dtype=torch.int32, device=dev)
        _n.check(lib.wsis_bfs_order(_n.ptr(idx), _n.ptr(start_len), _n.ptr(sem), _n.ptr(seeds), _n.ptr(offsets), n_c,
                                    _n.ptr(pos), _n.ptr(stamp), _n.ptr(cluster_idxs), st), "bfs_order")
    return cluster_idxs, offsets


class BFSCluster(Function):
    @staticmethod
    def forward(ctx, semantic_label, ball_query_idxs, start_len, threshold):
        """int tensors -> cluster_idxs int [sumNPoint,2], cluster_offsets int [nCluster+1].  CPU tensors (the upstream
        contract) run the host walk of libwsis_host.so; CUDA tensors run the device version (csrc/cluster.hip: union-find
        components + one workgroup per kept cluster replaying the FIFO order) -- same output, element for element."""
        if semantic_label.is_cuda:
            return _bfs_cluster_device(semantic_label, ball_query_idxs, start_len, int(threshold))
        for t in (semantic_label, ball_query_idxs, start_len):
            if t.is_cuda:
                raise _n.WsisError("bfs_cluster: all tensors on one device")
            assert t.dtype == torch.int32 and t.is_contiguous()
        import ctypes
        N = semantic_label.size(0)
        lib = _n.host()
        point_cluster = torch.empty(N, dtype=torch.int32)
        order = torch.empty(N, dtype=torch.int32)
        nc = ctypes.c_int64(0)
        npnt = ctypes.c_int64(0)
        _n.check_host(lib.wsis_host_bfs_cluster_count(_n.ptr(semantic_label), _n.ptr(ball_query_idxs),
                                                      _n.ptr(start_len), N, int(threshold),
                                                      _n.ptr(point_cluster), _n.ptr(order),
                                                      ctypes.addressof(nc), ctypes.addressof(npnt)),
                      "bfs_cluster_count")
        cluster_idxs = torch.empty((npnt.value, 2), dtype=torch.int32)
        cluster_offsets = torch.empty(nc.value + 1, dtype=torch.int32)
        _n.check_host(lib.wsis_host_bfs_cluster_fill(_n.ptr(point_cluster), _n.ptr(order), N, nc.value,
                                                     npnt.value, _n.ptr(cluster_idxs),
                                                     _n.ptr(cluster_offsets)), "bfs_cluster_fill")
        return cluster_idxs, cluster_offsets

    @staticmethod
    def backward(ctx, a=None):
        return None


bfs_cluster = BFSCluster.apply
