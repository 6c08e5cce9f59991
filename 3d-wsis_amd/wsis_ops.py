"""Fused operators of the hot path that have no third-party name in the reference:

* ``edge_affinity``      -- modules/model/backbone_3D_WSIS.py:218-244 (q.k dot, pos-enc scaling, segment
                            softmax over the out-edges of u, weighted V sum) as one HIP kernel + backward.
* ``affinity_matrix`` / ``propagate_class`` / ``weak_label_propagation`` -- the dense S x S fp64 affinity
  product of train_scannetv2.py:562-570 and modules/datasets/scannetv2_dataset.py:664-736 on the f64 MFMA.
"""
import ctypes
import os

import numpy as np
import torch
from torch.autograd import Function

import wsis_native as _n
from torch_scatter import SegmentCSR


def branch_stream(device, i=0):
    """the stream for work that is independent of the main chain of a step (the filter net beside the UNet, the point-level
    head beside the superpoint recurrence); autograd runs the backward of such work on the same stream.  It is the
    rulebook side stream of spconv.ops (idle once the pyramid of the step is built): a process keeps to four streams
    -- main, this one, the weight-gradient stream, the count check -- because HIP maps streams onto four hardware queues
    and a fifth stream shares its queue with one of the others (measured: 8.52 or 8.74 ms per step depending on which).
    WSIS_BRANCH=0: None (everything on the current stream)."""
    if os.environ.get("WSIS_BRANCH", "1") == "0" or device.type != "cuda":
        return None
    from spconv import ops as sp_ops
    return sp_ops._side_stream(device)


class EdgeGraph(object):
    """CSR over sources and over targets of a directed edge list (built once per batch)."""

    def __init__(self, edge_u, edge_v, num_nodes, num_src=None, csr_u=None, csr_v=None):
        """``num_src`` = edge_u.max() + 1, the row count of the reference's ``scatter(..., edge_u, dim=0)``
        (backbone_3D_WSIS.py:232).  Read from the device when not given: that D2H copy waits for everything queued
        on the stream (a whole training step when the host runs ahead), so loaders pass the value they know from
        the host-side edge list."""
        _n.require_cuda(edge_u, edge_v)
        self.eu = edge_u.contiguous().long()
        self.ev = edge_v.contiguous().long()
        self.E = self.eu.numel()
        self.S = int(num_nodes)
        if num_src is not None:
            self.Su = int(num_src) if self.E > 0 else 0
            assert 0 <= self.Su <= self.S
        else:
            self.Su = int(self.eu.max().item()) + 1 if self.E > 0 else 0
        # (a caller that builds all CSRs of a batch with one sort hands them in: torch_scatter.segment_csr_batch)
        self.csr_u = csr_u if csr_u is not None else SegmentCSR(self.eu, self.Su)
        self.csr_v = csr_v if csr_v is not None else SegmentCSR(self.ev, self.S)
        assert self.csr_u.S == self.Su and self.csr_v.S == self.S


class _EdgeAffinity(Function):
    @staticmethod
    def forward(ctx, q, k, v, pos, graph, scale):
        _n.require_cuda(q, k, v, pos)
        q, k, v = q.contiguous().float(), k.contiguous().float(), v.contiguous().float()
        pos = pos.contiguous().float().view(-1)
        S, D = q.shape
        assert S == graph.S and pos.numel() == graph.E
        aff = torch.empty(graph.E, dtype=torch.float32, device=q.device)
        res = torch.empty((graph.Su, D), dtype=torch.float32, device=q.device)
        _n.check(_n.hip().wsis_edge_affinity_fwd(_n.ptr(q), _n.ptr(k), _n.ptr(v), _n.ptr(pos), _n.ptr(graph.eu),
                                                 _n.ptr(graph.ev), _n.ptr(graph.csr_u.perm),
                                                 _n.ptr(graph.csr_u.offsets), float(scale), _n.ptr(aff),
                                                 _n.ptr(res), graph.E, graph.Su, D, _n.stream_ptr()),
                 "edge_affinity_fwd")
        ctx.save_for_backward(q, k, v, pos, aff)
        ctx.graph, ctx.scale = graph, float(scale)
        return aff, res

    @staticmethod
    def backward(ctx, d_aff, d_res):
        q, k, v, pos, aff = ctx.saved_tensors
        g = ctx.graph
        S, D = q.shape
        dev = q.device
        d_res = d_res.contiguous().float() if d_res is not None else torch.zeros((g.Su, D), device=dev)
        d_aff = d_aff.contiguous().float() if d_aff is not None else None
        dq = torch.empty((S, D), dtype=torch.float32, device=dev)
        dk = torch.empty((S, D), dtype=torch.float32, device=dev)
        dv = torch.empty((S, D), dtype=torch.float32, device=dev)
        dpos = torch.empty(max(g.E, 1), dtype=torch.float32, device=dev)
        tmp = torch.empty(max(2 * g.E, 1), dtype=torch.float32, device=dev)
        _n.check(_n.hip().wsis_edge_affinity_bwd(
            _n.ptr(q), _n.ptr(k), _n.ptr(v), _n.ptr(pos), _n.ptr(aff), _n.ptr(g.eu), _n.ptr(g.ev),
            _n.ptr(g.csr_u.perm), _n.ptr(g.csr_u.offsets), _n.ptr(g.csr_v.perm), _n.ptr(g.csr_v.offsets),
            ctx.scale, _n.ptr(d_aff), _n.ptr(d_res), _n.ptr(dq), _n.ptr(dk), _n.ptr(dv), _n.ptr(dpos),
            _n.ptr(tmp), g.E, S, g.Su, D, _n.stream_ptr()), "edge_affinity_bwd")
        return dq, dk, dv, dpos[:g.E], None, None


def edge_affinity(q, k, v, pos_enc, graph, scale):
    """returns (edge_affinity [E], res [max(u)+1, D])."""
    return _EdgeAffinity.apply(q, k, v, pos_enc, graph, scale)


class _PosEnc(Function):
    """pos_e = fc_position(centre[u_e] - centre[v_e]) (backbone_3D_WSIS.py:54-58, 222-224) as one launch forward, two
    backward (csrc/affinity.hip); the centres are data (no gradient)"""

    @staticmethod
    def forward(ctx, centre, eu, ev, W1, b1, W2, b2):
        _n.require_cuda(centre, eu, ev, W1)
        centre = centre.contiguous().float()
        E = eu.numel()
        pos = torch.empty(E, dtype=torch.float32, device=centre.device)
        _n.check(_n.hip().wsis_pos_enc_fwd(_n.ptr(centre), _n.ptr(eu), _n.ptr(ev), _n.ptr(W1), _n.ptr(b1), _n.ptr(W2),
                                           _n.ptr(b2), _n.ptr(pos), E, _n.stream_ptr()), "pos_enc_fwd")
        ctx.save_for_backward(centre, eu, ev, W1, b1, W2)
        return pos

    @staticmethod
    def backward(ctx, dpos):
        centre, eu, ev, W1, b1, W2 = ctx.saved_tensors
        E = eu.numel()
        need = ctx.needs_input_grad
        dW1 = torch.empty_like(W1) if need[3] else None
        db1 = torch.empty_like(b1) if need[4] else None
        dW2 = torch.empty_like(W2) if need[5] else None
        db2 = torch.empty(1, dtype=torch.float32, device=W1.device) if need[6] else None
        if E == 0:
            return (None, None, None) + tuple(None if t is None else t.zero_() for t in (dW1, db1, dW2, db2))
        lib = _n.hip()
        ws_bytes = lib.wsis_pos_enc_workspace_bytes(E)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=W1.device)
        _n.check(lib.wsis_pos_enc_bwd(_n.ptr(centre), _n.ptr(eu), _n.ptr(ev), _n.ptr(W1), _n.ptr(b1), _n.ptr(W2),
                                      _n.ptr(dpos.contiguous().float()), _n.ptr(dW1), _n.ptr(db1), _n.ptr(dW2), _n.ptr(db2),
                                      E, _n.ptr(ws), ws_bytes, _n.stream_ptr()), "pos_enc_bwd")
        return None, None, None, dW1, db1, dW2, db2


def edge_position_encoding(fc_position, centre, edge_u, edge_v):
    """[E] position encodings, or None where the fused form does not apply (the caller runs the modules)"""
    if (not centre.is_cuda or centre.requires_grad or len(fc_position) != 3 or centre.dim() != 2 or centre.shape[1] != 3
            or os.environ.get("WSIS_FUSE_POS_ENC", "1") == "0"):
        return None
    l1, act, l2 = fc_position[0], fc_position[1], fc_position[2]
    if not (type(l1) is torch.nn.Linear and type(l2) is torch.nn.Linear and isinstance(act, torch.nn.ReLU)
            and l1.in_features == 3 and l1.out_features == 16 and l2.in_features == 16 and l2.out_features == 1
            and l1.bias is not None and l2.bias is not None):
        return None
    eu, ev = edge_u.contiguous().long(), edge_v.contiguous().long()
    return _PosEnc.apply(centre, eu, ev, l1.weight, l1.bias, l2.weight, l2.bias)


# ---- a17: dense affinity matrix + label propagation -------------------------------------------------

def affinity_matrix(edge_u, edge_v, edge_affinity_vals, S):
    """A [S,S] float64 with A[u,v] = affinity of edge (u,v) (train_scannetv2.py:567-570)."""
    _n.require_cuda(edge_u, edge_v, edge_affinity_vals)
    eu, ev = edge_u.contiguous().long(), edge_v.contiguous().long()
    aff = edge_affinity_vals.detach().contiguous().float()
    A = torch.empty((S, S), dtype=torch.float64, device=aff.device)
    _n.check(_n.hip().wsis_affinity_dense_build(_n.ptr(eu), _n.ptr(ev), _n.ptr(aff), eu.numel(), _n.ptr(A), S,
                                                _n.stream_ptr()), "affinity_dense_build")
    return A


def dgemm(A, B):
    _n.require_cuda(A, B)
    assert A.dtype == torch.float64 and B.dtype == torch.float64
    A, B = A.contiguous(), B.contiguous()
    M, Kd = A.shape
    N = B.shape[1]
    C = torch.empty((M, N), dtype=torch.float64, device=A.device)
    _n.check(_n.hip().wsis_dgemm(_n.ptr(A), _n.ptr(B), _n.ptr(C), M, N, Kd, _n.stream_ptr()), "dgemm")
    return C


def propagate_class(A, adj_u8, pred, conf, label, cls, iterations_num, thr=0.7):
    """One class of modules/datasets/scannetv2_dataset.py:689-721 on the device.
    returns (instance_scores fp64 [S], instance_pseudo_label int32 [S])."""
    S = A.shape[0]
    lib = _n.hip()
    st = _n.stream_ptr()
    T0 = torch.empty_like(A)
    _n.check(lib.wsis_affinity_transition(_n.ptr(A), _n.ptr(adj_u8), _n.ptr(pred), _n.ptr(conf), _n.ptr(label),
                                          int(cls), float(thr), _n.ptr(T0), S, st), "affinity_transition")
    T = T0
    for _ in range(int(iterations_num)):
        T = dgemm(T, T0)
    scores = torch.empty(S, dtype=torch.float64, device=A.device)
    arg = torch.empty(S, dtype=torch.int32, device=A.device)
    _n.check(lib.wsis_affinity_colmax(_n.ptr(T), _n.ptr(label), int(cls), _n.ptr(scores), _n.ptr(arg), S, st),
             "affinity_colmax")
    return scores, arg


def propagate_sparse(A, adj_u8, pred, conf, label, present, class_num, iterations_num, thr=0.7):
    """All present classes of modules/datasets/scannetv2_dataset.py:689-721 in ONE call on the sparse structure
    (``wsis_affinity_propagate_sparse``): T0_c has the sparsity of the edge list and only the rows of the superpoints
    labelled c of T0_c^(n+1) are used, so the S^3 dense product per class and iteration is a chain of sparse
    row-vector products.  returns (scores fp64 [n_present, S], arg int32 [n_present, S])."""
    S = A.shape[0]
    dev = A.device
    lib = _n.hip()
    n_present = len(present)
    cls_of = torch.tensor(list(present), dtype=torch.int32, device=dev)
    ci = np.full(class_num, -1, dtype=np.int32)
    ci[np.asarray(present, dtype=np.int64)] = np.arange(n_present, dtype=np.int32)
    ci_of_cls = torch.from_numpy(ci).to(dev)
    nnz = int(torch.count_nonzero(A))          # (stage boundary, once per scene: one read-back)
    col = torch.empty(max(nnz, 1), dtype=torch.int32, device=dev)
    val = torch.empty(max(nnz, 1), dtype=torch.float64, device=dev)
    scores = torch.empty((n_present, S), dtype=torch.float64, device=dev)
    arg = torch.empty((n_present, S), dtype=torch.int32, device=dev)
    ws_bytes = lib.wsis_affinity_propagate_sparse_workspace_bytes(S, n_present)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    _n.check(lib.wsis_affinity_propagate_sparse(_n.ptr(A), _n.ptr(adj_u8), _n.ptr(pred), _n.ptr(conf), _n.ptr(label),
                                                _n.ptr(cls_of), _n.ptr(ci_of_cls), n_present, int(class_num), float(thr),
                                                int(iterations_num), S, _n.ptr(col), _n.ptr(val), nnz, _n.ptr(scores),
                                                _n.ptr(arg), _n.ptr(ws), ws_bytes, _n.stream_ptr()),
             "affinity_propagate_sparse")
    return scores, arg


PROP_SPARSE_MAX_S = 8188       # two rows of S doubles + the non-zero list in 160 KB of LDS


def weak_label_propagation(A, adjacency, sp_semantic_value, superpoint_pred_semantic, superpoint_semantic_label,
                           iterations_num, class_num, dense=None):
    """Device version of ScanNetV2Inst_spg.weak_label_propagation up to ``pseudo_label_final``
    (modules/datasets/scannetv2_dataset.py:664-736).  ``adjacency`` is the [S,S] adjacency (without the
    identity; it is added here like ``:681-682``).  Returns (pseudo_label_final [S] float64 numpy with -100
    for unknown, pseudo_label_scores [S]).  Default: the sparse form (``propagate_sparse``); ``dense=True`` (or
    WSIS_PROP_DENSE=1, or S > 8188) runs the per-class dense products on the f64 matrix cores."""
    dev = A.device
    S = A.shape[0]
    if dense is None:
        dense = os.environ.get("WSIS_PROP_DENSE", "0") != "0"
    dense = dense or S > PROP_SPARSE_MAX_S
    adj = torch.as_tensor(adjacency, device=dev)
    adj_u8 = (adj.to(torch.int32) + torch.eye(S, dtype=torch.int32, device=dev)).to(torch.uint8).contiguous()
    pred = torch.as_tensor(superpoint_pred_semantic, device=dev).to(torch.int32).contiguous()
    conf = torch.as_tensor(sp_semantic_value, device=dev).to(torch.float32).contiguous()
    label_np = np.asarray(superpoint_semantic_label)
    label = torch.as_tensor(label_np, device=dev).to(torch.int32).contiguous()
    present = [c for c in range(class_num) if (label_np == c).sum() != 0]
    if not present:
        return np.ones(S) * -100, np.zeros(S)
    if dense:
        scores_list, pseudo_list = [], []
        for c in present:
            s, a = propagate_class(A, adj_u8, pred, conf, label, c, iterations_num)
            scores_list.append(s)
            pseudo_list.append(a)
        scores = torch.stack(scores_list).cpu().numpy()
        pseudo = torch.stack(pseudo_list).cpu().numpy()
    else:
        s, a = propagate_sparse(A.contiguous(), adj_u8, pred, conf, label, present, class_num, iterations_num)
        scores, pseudo = s.cpu().numpy(), a.cpu().numpy()
    _ind = np.argmax(scores, axis=0)
    pseudo_label = np.choose(_ind, pseudo)
    pseudo_label_scores = np.choose(_ind, scores)
    final = np.ones(S) * -100
    unknown = (pseudo_label_scores != 0) & (label_np == -100)
    final[unknown] = pseudo_label[unknown]
    return final, pseudo_label_scores


# ---- a12: fused BatchNorm1d(+ReLU) -------------------------------------------------------------------------

class _BatchNormReLU(Function):
    """y = relu?(batch_norm(x)) with torch.nn.BatchNorm1d semantics (training: batch statistics + running-stat
    update; eval: running statistics).  sparse_unet3d.py:128-137 runs these as separate torch modules."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, training, momentum, eps, relu):
        _n.require_cuda(x)
        x = x.contiguous().float()
        M, C = x.shape
        lib = _n.hip()
        st = _n.stream_ptr()
        dev = x.device
        use_batch_stats = training or running_mean is None
        ws = None
        if use_batch_stats:
            assert M >= 1
            ws_bytes = lib.wsis_bn_workspace_bytes(M, C)
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            mean = torch.empty(C, dtype=torch.float32, device=dev)
            var = torch.empty(C, dtype=torch.float32, device=dev)
            upd = training and running_mean is not None
            _n.check(lib.wsis_bn_stats(_n.ptr(x), M, C, _n.ptr(mean), _n.ptr(var),
                                       _n.ptr(running_mean) if upd else None,
                                       _n.ptr(running_var) if upd else None, float(momentum), _n.ptr(ws),
                                       ws_bytes, st), "bn_stats")
        else:
            mean, var = running_mean, running_var
        y = torch.empty_like(x)
        _n.check(lib.wsis_bn_apply(_n.ptr(x), _n.ptr(mean), _n.ptr(var), _n.ptr(weight), _n.ptr(bias), float(eps),
                                   int(relu), _n.ptr(y), M, C, st), "bn_apply")
        ctx.save_for_backward(x, mean, var, weight, bias)
        ctx.cfg = (float(eps), int(relu), int(use_batch_stats))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, var, weight, bias = ctx.saved_tensors
        eps, relu, batch_stats = ctx.cfg
        dy = dy.contiguous().float()
        M, C = x.shape
        lib = _n.hip()
        dev = x.device
        ws_bytes = lib.wsis_bn_workspace_bytes(M, C)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        dgamma = torch.empty(C, dtype=torch.float32, device=dev)
        dbeta = torch.empty(C, dtype=torch.float32, device=dev)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        _n.check(lib.wsis_bn_bwd(_n.ptr(x), _n.ptr(dy), _n.ptr(mean), _n.ptr(var), _n.ptr(weight), _n.ptr(bias), eps,
                                 relu, batch_stats, _n.ptr(dx), _n.ptr(dgamma), _n.ptr(dbeta), None, M, C, _n.ptr(ws),
                                 ws_bytes, _n.stream_ptr()), "bn_bwd")
        gw = dgamma if (weight is not None and ctx.needs_input_grad[1]) else None
        gb = dbeta if (bias is not None and ctx.needs_input_grad[2]) else None
        return dx, gw, gb, None, None, None, None, None, None


class _SyncBatchNormReLU(Function):
    """_BatchNormReLU with the batch statistics taken over ALL ranks' rows: what torch.nn.SyncBatchNorm computes, which
    the reference converts every BatchNorm to when num_gpus > 1 (train_scannetv2.py:734-736).  Forward: local
    (mean, var) -> (count, sum, sum of squares) in fp64 -> one all-reduce -> global mean / biased var (running
    statistics from the global unbiased var) -> apply.  Backward: local (sum dz, sum dz*xhat) -> one all-reduce ->
    dx with the global sums and count; dgamma / dbeta stay the LOCAL sums (the gradient exchange averages them, as
    DistributedDataParallel does behind SyncBatchNorm).  Two small collectives per layer and pass: opt-in
    (wsis_parallel.convert_sync_batchnorm), and the UNet then runs as the module walk."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, momentum, eps, relu, group):
        import torch.distributed as dist
        _n.require_cuda(x)
        x = x.contiguous().float()
        M, C = x.shape
        lib, st, dev = _n.hip(), _n.stream_ptr(), x.device
        mean_l = torch.zeros(C, dtype=torch.float32, device=dev)
        var_l = torch.zeros(C, dtype=torch.float32, device=dev)
        if M > 0:
            ws_bytes = lib.wsis_bn_workspace_bytes(M, C)
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            _n.check(lib.wsis_bn_stats(_n.ptr(x), M, C, _n.ptr(mean_l), _n.ptr(var_l), None, None, float(momentum),
                                       _n.ptr(ws), ws_bytes, st), "bn_stats")
        # every rank's (mean, biased var, count) gathered in ONE collective and combined in fp64 with Chan's formula:
        # var = sum n_i (var_i + (mean_i - mean)^2) / N.  (count, sum, sum of squares) from the fp32-rounded local
        # statistics cancels when |mean| >> sigma: the rounding of mean_i alone is ~1e-7 mean^2 against var.)
        t = torch.cat((mean_l.double(), var_l.double(), torch.full((1,), float(M), dtype=torch.float64, device=dev)))
        world = dist.get_world_size(group)
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(parts, t, group=group)
        g = torch.stack(parts)                                   # [world, 2C + 1]
        n_i = g[:, 2 * C:2 * C + 1]                              # [world, 1]
        N = n_i.sum()
        mean64 = (g[:, :C] * n_i).sum(0) / N
        dm = g[:, :C] - mean64
        var64 = ((g[:, C:2 * C] + dm * dm) * n_i).sum(0) / N
        mean, var = mean64.float(), var64.float()
        if running_mean is not None:
            unb = var64 * (N / (N - 1.0).clamp_min(1.0))
            running_mean.mul_(1.0 - momentum).add_(mean, alpha=momentum)
            running_var.mul_(1.0 - momentum).add_(unb.float(), alpha=momentum)
        y = torch.empty_like(x)
        if M > 0:
            _n.check(lib.wsis_bn_apply(_n.ptr(x), _n.ptr(mean), _n.ptr(var), _n.ptr(weight), _n.ptr(bias), float(eps),
                                       int(relu), _n.ptr(y), M, C, st), "bn_apply")
        ctx.save_for_backward(x, mean, var, weight, bias, N)
        ctx.cfg = (float(eps), int(relu), group)
        return y

    @staticmethod
    def backward(ctx, dy):
        import torch.distributed as dist
        x, mean, var, weight, bias, N = ctx.saved_tensors
        eps, relu, group = ctx.cfg
        dy = dy.contiguous().float()
        M, C = x.shape
        lib, st, dev = _n.hip(), _n.stream_ptr(), x.device
        dgamma = torch.zeros(C, dtype=torch.float32, device=dev)
        dbeta = torch.zeros(C, dtype=torch.float32, device=dev)
        if M > 0:
            ws_bytes = lib.wsis_bn_workspace_bytes(M, C)
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            _n.check(lib.wsis_bn_bwd(_n.ptr(x), _n.ptr(dy), _n.ptr(mean), _n.ptr(var), _n.ptr(weight), _n.ptr(bias), eps,
                                     relu, 1, None, _n.ptr(dgamma), _n.ptr(dbeta), None, M, C, _n.ptr(ws), ws_bytes, st),
                     "bn_bwd")
        g = torch.cat((dgamma, dbeta)).double()
        dist.all_reduce(g, group=group)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            if M > 0:
                sc = (g * (float(M) / N)).float()             # the kernel divides by the local row count
                _n.check(lib.wsis_bn_bwd_apply(_n.ptr(x), _n.ptr(dy), _n.ptr(mean), _n.ptr(var), _n.ptr(weight),
                                               _n.ptr(bias), _n.ptr(sc[:C]), _n.ptr(sc[C:]), eps, relu, _n.ptr(dx), None,
                                               M, C, st), "bn_bwd_apply")
        gw = dgamma if (weight is not None and ctx.needs_input_grad[1]) else None
        gb = dbeta if (bias is not None and ctx.needs_input_grad[2]) else None
        return dx, gw, gb, None, None, None, None, None, None


def sync_group(bn):
    """the process group a BatchNorm module shares its batch statistics with (None: it does not).  Set by
    wsis_parallel.convert_sync_batchnorm; only in training mode, with a group of more than one rank."""
    g = getattr(bn, "_wsis_sync", None)
    if g is None or not bn.training:
        return None
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized():
        return None
    grp = dist.group.WORLD if g is True else g
    return grp if dist.get_world_size(grp) > 1 else None


def _flush_batch_count(bn, *args):
    n = getattr(bn, "_wsis_pending_batches", 0)
    if n:
        bn.num_batches_tracked += n
        bn._wsis_pending_batches = 0


def _defer_batch_count(bn):
    """num_batches_tracked only feeds momentum=None; with a fixed momentum it is bookkeeping, so the +1 is
    accumulated on the host and written back when the module's state is read (state_dict) or on demand."""
    if not hasattr(bn, "_wsis_pending_batches"):
        bn._wsis_pending_batches = 0
        bn.register_state_dict_pre_hook(lambda module, prefix, keep_vars: _flush_batch_count(module))
    bn._wsis_pending_batches += 1


def flush_bn_counters(model):
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            _flush_batch_count(m)


def batch_norm_relu(x, bn, relu=True):
    """fused forward of an nn.BatchNorm1d module (its parameters/buffers) optionally followed by ReLU"""
    if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
        if bn.momentum is None:
            bn.num_batches_tracked += 1          # the cumulative-average mode needs the live value
        else:
            _defer_batch_count(bn)               # one tiny kernel per BN per step otherwise: flushed lazily
    momentum = 0.1 if bn.momentum is None else bn.momentum
    grp = sync_group(bn)
    if grp is not None:
        return _SyncBatchNormReLU.apply(x, bn.weight, bn.bias, bn.running_mean if bn.track_running_stats else None,
                                        bn.running_var if bn.track_running_stats else None, momentum, bn.eps, relu, grp)
    return _BatchNormReLU.apply(x, bn.weight, bn.bias, bn.running_mean if bn.track_running_stats else None,
                                bn.running_var if bn.track_running_stats else None, bn.training, momentum, bn.eps,
                                relu)


# ---- a21: edge-conditioned message passing (ECC GNN) ----------------------------------------------------------

class _EccMessage(Function):
    """inp[s] = mean over edges (s -> t) of x[t] @ W_(s->t)   (spg_modules.py:97-121, aggr='mean')"""

    @staticmethod
    def forward(ctx, x, weights, src, dst, csr_src, csr_dst):
        _n.require_cuda(x, weights)
        x = x.contiguous().float()
        w = weights.contiguous().float()
        S, C = x.shape
        E = src.numel()
        assert w.shape == (E, C, C) and csr_src.S == S and csr_dst.S == S
        out = torch.empty((S, C), dtype=torch.float32, device=x.device)
        _n.check(_n.hip().wsis_ecc_message_fwd(_n.ptr(x), _n.ptr(w), _n.ptr(dst), _n.ptr(csr_src.perm),
                                               _n.ptr(csr_src.offsets), _n.ptr(out), S, E, C, _n.stream_ptr()),
                 "ecc_message_fwd")
        ctx.save_for_backward(x, w)
        ctx.graph = (src, dst, csr_src, csr_dst)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, w = ctx.saved_tensors
        src, dst, csr_src, csr_dst = ctx.graph
        S, C = x.shape
        E = src.numel()
        dout = dout.contiguous().float()
        dx = torch.empty_like(x)
        dw = torch.empty_like(w)
        _n.check(_n.hip().wsis_ecc_message_bwd(_n.ptr(x), _n.ptr(w), _n.ptr(dout), _n.ptr(src), _n.ptr(csr_dst.perm),
                                               _n.ptr(csr_dst.offsets), _n.ptr(csr_src.offsets), _n.ptr(dx),
                                               _n.ptr(dw), S, E, C, _n.stream_ptr()), "ecc_message_bwd")
        return dx, dw, None, None, None, None


def ecc_message(x, weights, src, dst, csr_src, csr_dst):
    return _EccMessage.apply(x, weights, src, dst, csr_src, csr_dst)


class _EccContract(Function):
    """m[e] = sum_c h[e,c] U[dst_e, c, :] + U[dst_e, 64, :]  -- the edge-conditioned message x_t @ W_e without the
    per-edge filter tensor (csrc/ecc.hip; reference spg_modules.py:97-121 with the fnet of
    graphnet.py:19-36)."""

    @staticmethod
    def forward(ctx, h, U, csr_dst):
        _n.require_cuda(h, U)
        h, U = h.contiguous().float(), U.contiguous().float()
        E, S = h.shape[0], U.shape[0]
        assert h.shape[1] == 64 and U.shape[1] == 65 * 32 and csr_dst.S == S
        m = torch.empty((E, 32), dtype=torch.float32, device=h.device)
        _n.check(_n.hip().wsis_ecc_contract_fwd(_n.ptr(h), _n.ptr(U), _n.ptr(csr_dst.perm), _n.ptr(csr_dst.offsets),
                                                _n.ptr(m), S, E, _n.stream_ptr()), "ecc_contract_fwd")
        ctx.save_for_backward(h, U)
        ctx.csr = csr_dst
        return m

    @staticmethod
    def backward(ctx, dm):
        h, U = ctx.saved_tensors
        csr = ctx.csr
        dm = dm.contiguous().float()
        dU, dh = torch.empty_like(U), torch.empty_like(h)
        _n.check(_n.hip().wsis_ecc_contract_bwd(_n.ptr(h), _n.ptr(U), _n.ptr(dm), _n.ptr(csr.perm), _n.ptr(csr.offsets),
                                                _n.ptr(dU), _n.ptr(dh), U.shape[0], h.shape[0], _n.stream_ptr()),
                 "ecc_contract_bwd")
        return dh, dU, None


def ecc_contract(h, U, csr_dst):
    return _EccContract.apply(h, U, csr_dst)


# ---- a21: fused GRUCellEx -------------------------------------------------------------------------------------

class _GruCellEx(Function):
    """hy = GRUCellEx(x, h) with input gate and per-row normalised gates (spg_modules.py:226-253), C == 32."""

    @staticmethod
    def forward(ctx, x, h, w_ig, b_ig, w_ih, w_hh, b_ih, b_hh):
        _n.require_cuda(x, h)
        args = [t.contiguous().float() for t in (x, h, w_ig, b_ig, w_ih, w_hh, b_ih, b_hh)]
        x, h = args[0], args[1]
        S, C = x.shape
        hy = torch.empty_like(x)
        _n.check(_n.hip().wsis_gru_cell_fwd(*[_n.ptr(t) for t in args], _n.ptr(hy), S, C, _n.stream_ptr()),
                 "gru_cell_fwd")
        ctx.save_for_backward(*args)
        return hy

    @staticmethod
    def backward(ctx, dhy):
        args = ctx.saved_tensors
        x, h, w_ig, b_ig, w_ih, w_hh, b_ih, b_hh = args
        S, C = x.shape
        lib = _n.hip()
        dhy = dhy.contiguous().float()
        dx, dh = torch.empty_like(x), torch.empty_like(h)
        dwig, dbig = torch.empty_like(w_ig), torch.empty_like(b_ig)
        dwih, dwhh = torch.empty_like(w_ih), torch.empty_like(w_hh)
        dbih, dbhh = torch.empty_like(b_ih), torch.empty_like(b_hh)
        ws_bytes = lib.wsis_gru_cell_workspace_bytes(S)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
        _n.check(lib.wsis_gru_cell_bwd(*[_n.ptr(t) for t in args], _n.ptr(dhy), _n.ptr(dx), _n.ptr(dh), _n.ptr(dwig),
                                       _n.ptr(dbig), _n.ptr(dwih), _n.ptr(dwhh), _n.ptr(dbih), _n.ptr(dbhh), S, C,
                                       _n.ptr(ws), ws_bytes, _n.stream_ptr()), "gru_cell_bwd")
        return dx, dh, dwig, dbig, dwih, dwhh, dbih, dbhh


def gru_cell_ex(x, h, cell):
    """fused forward of a graphnet.GRUCellEx module (layernorm + ingate configuration, 32 channels)"""
    ig = cell._modules["ig"]
    return _GruCellEx.apply(x, h, ig.weight, ig.bias, cell.weight_ih, cell.weight_hh, cell.bias_ih, cell.bias_hh)


# ---- the whole recurrent graph convolution as ONE autograd node -------------------------------------------------

class _EccGruLoop(Function):
    """R x { U = hx @ W' -> m_e = h_e . U_t (ecc_contract) -> mean over the out-edges -> GRUCellEx }
    (spg_modules.py:152-185) recorded as one node: the same kernels in the same order as the per-op Functions above,
    but the 4 R autograd nodes, their Python dispatch and the gradient-accumulation kernels of the shared parameters
    (W', the edge features h, the six GRU tensors, all used R times) collapse into one forward and one backward call
    with in-place accumulation.  Returns cat([hx_0 .. hx_R], 1) or hx_R."""

    @staticmethod
    def forward(ctx, hx, h, Waug, w_ig, b_ig, w_ih, w_hh, b_ih, b_hh, csr_src, csr_dst, repeats, cat_all):
        _n.require_cuda(hx, h)
        lib = _n.hip()
        f = lambda t: t.contiguous().float()
        hx, h, Waug = f(hx), f(h), f(Waug)
        gp = [f(t) for t in (w_ig, b_ig, w_ih, w_hh, b_ih, b_hh)]
        S, E = hx.shape[0], h.shape[0]
        assert hx.shape[1] == 32 and h.shape[1] == 64 and Waug.shape == (32, 65 * 32) and csr_dst.S == S and csr_src.S == S
        st = _n.stream_ptr()
        # hx_0 .. hx_R stacked in one buffer: the backward reduces dW' over all R iterations in ONE launch
        hx_all = torch.empty(((repeats + 1) * S, 32), dtype=torch.float32, device=hx.device)
        hxs = [hx_all[i * S:(i + 1) * S] for i in range(repeats + 1)]
        hxs[0].copy_(hx)
        inps, Us = [], []
        own = os.environ.get("WSIS_ECC_OWN_GEMM", "1") != "0"     # the two K / N = 32 products on csrc/ecc.hip's kernels
        for i in range(repeats):
            if own:
                U = torch.empty((S, Waug.shape[1]), dtype=torch.float32, device=hx.device)
                _n.check(lib.wsis_ecc_u_fwd(_n.ptr(hxs[i]), _n.ptr(Waug), _n.ptr(U), S, st), "ecc_u_fwd")
            else:
                U = hxs[i] @ Waug
            m = torch.empty((E, 32), dtype=torch.float32, device=hx.device)
            _n.check(lib.wsis_ecc_contract_fwd(_n.ptr(h), _n.ptr(U), _n.ptr(csr_dst.perm), _n.ptr(csr_dst.offsets),
                                               _n.ptr(m), S, E, st), "ecc_contract_fwd")
            inp = torch.empty((S, 32), dtype=torch.float32, device=hx.device)
            if own:      # the mean over the in-edges is formed by the cell's launch (same order of additions)
                _n.check(lib.wsis_gru_cell_fwd_mean(_n.ptr(m), _n.ptr(csr_src.perm), _n.ptr(csr_src.offsets), _n.ptr(inp),
                                                    _n.ptr(hxs[i]), *[_n.ptr(t) for t in gp], _n.ptr(hxs[i + 1]), S, 32,
                                                    st), "gru_cell_fwd_mean")
            else:
                _n.check(lib.wsis_segment_reduce_fwd(_n.ptr(m), _n.ptr(csr_src.perm), _n.ptr(csr_src.offsets),
                                                     _n.ptr(inp), None, E, S, 32, 1, st), "segment_reduce_fwd")
                _n.check(lib.wsis_gru_cell_fwd(_n.ptr(inp), _n.ptr(hxs[i]), *[_n.ptr(t) for t in gp], _n.ptr(hxs[i + 1]),
                                               S, 32, st), "gru_cell_fwd")
            Us.append(U)
            inps.append(inp)
        ctx.save_for_backward(h, Waug, *gp, hx_all, *inps, *Us)
        ctx.meta = (csr_src, csr_dst, repeats, cat_all)
        return torch.cat(hxs, 1) if cat_all else hxs[-1].clone()     # never a view of the saved buffer

    @staticmethod
    def backward(ctx, dout):
        csr_src, csr_dst, R, cat_all = ctx.meta
        sv = ctx.saved_tensors
        h, Waug, gp = sv[0], sv[1], list(sv[2:8])
        hx_all, inps, Us = sv[8], sv[9:9 + R], sv[9 + R:9 + 2 * R]
        lib = _n.hip()
        st = _n.stream_ptr()
        E = h.shape[0]
        S = hx_all.shape[0] // (R + 1)
        hxs = [hx_all[i * S:(i + 1) * S] for i in range(R + 1)]
        dev = h.device
        dout = dout.contiguous().float()
        d_slices = [dout[:, 32 * i:32 * (i + 1)] for i in range(R + 1)] if cat_all else None
        own = os.environ.get("WSIS_ECC_OWN_GEMM", "1") != "0"
        # the gradient that reaches hx_{i+1}: what the later iterations pass down (d_carry; none for the last state) plus,
        # with cat_all, its own slice of the output gradient -- summed by the cell's backward as it loads them (own) or by
        # a launch per iteration
        pitch = dout.shape[1]
        d_carry = None if cat_all else dout
        dh = torch.empty_like(h) if R > 0 else torch.zeros_like(h)      # written by the first evaluated iteration
        # the six GRU parameter gradients: every iteration leaves its slabs in its own region, ONE reduce at the end
        dgp = [torch.empty_like(t) for t in gp] if R > 0 else [torch.zeros_like(t) for t in gp]
        ws_bytes = (lib.wsis_gru_cell_workspace_bytes(S) - 256) * max(R, 1) + 256
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        # dU of every iteration kept: dW' = sum_i hx_i^T dU_i = [hx_0; ..; hx_{R-1}]^T [dU_0; ..; dU_{R-1}] is one
        # row-split MFMA reduction (the sparse-conv dW kernel, K = 1, dense rows) instead of R one-workgroup GEMMs
        dU_all = torch.empty((R * S, Waug.shape[1]), dtype=torch.float32, device=dev)
        WaugT = Waug.t()
        for i in reversed(range(R)):
            d_inp, d_hprev = torch.empty((S, 32), dtype=torch.float32, device=dev), torch.empty((S, 32), dtype=torch.float32,
                                                                                                device=dev)
            if own:
                extra = (dout.data_ptr() + 4 * 32 * (i + 1)) if cat_all else None
            else:
                extra = None
                if cat_all:      # materialise carry + slice
                    d_carry = d_slices[i + 1].contiguous() if d_carry is None else d_carry.add_(d_slices[i + 1])
            _n.check(lib.wsis_gru_cell_bwd_seq(_n.ptr(inps[i]), _n.ptr(hxs[i]), *[_n.ptr(t) for t in gp], _n.ptr(d_carry),
                                               extra, pitch, _n.ptr(d_inp), _n.ptr(d_hprev), *[_n.ptr(t) for t in dgp], S,
                                               32, R - 1 - i, R, 1 if i == 0 else 0, _n.ptr(ws), ws_bytes, st),
                     "gru_cell_bwd_seq")
            dU = dU_all[i * S:(i + 1) * S]
            if own:
                # the mean's backward (d_m[e] = d_inp[src_e] / out-degree) is formed inside the contraction's backward
                _n.check(lib.wsis_ecc_contract_bwd_mean(_n.ptr(h), _n.ptr(Us[i]), _n.ptr(d_inp), _n.ptr(csr_src.index),
                                                        _n.ptr(csr_src.offsets), _n.ptr(csr_dst.perm),
                                                        _n.ptr(csr_dst.offsets), _n.ptr(dU), _n.ptr(dh), S, E,
                                                        0 if i == R - 1 else 1, st), "ecc_contract_bwd_mean")
            else:
                d_m = torch.empty((E, 32), dtype=torch.float32, device=dev)
                _n.check(lib.wsis_segment_reduce_bwd(_n.ptr(d_inp), _n.ptr(csr_src.index), _n.ptr(csr_src.offsets), None,
                                                     _n.ptr(d_m), E, S, 32, 1, st), "segment_reduce_bwd")
                _n.check(lib.wsis_ecc_contract_bwd_acc(_n.ptr(h), _n.ptr(Us[i]), _n.ptr(d_m), _n.ptr(csr_dst.perm),
                                                       _n.ptr(csr_dst.offsets), _n.ptr(dU), _n.ptr(dh), S, E,
                                                       0 if i == R - 1 else 1, st), "ecc_contract_bwd")
            d_carry = d_hprev.addmm_(dU, WaugT)          # in place: no copy of d_hprev into a new output first
        if R == 0:
            d_hx = d_slices[0].contiguous() if cat_all else dout
        else:
            d_hx = d_carry.add_(d_slices[0]) if cat_all else d_carry
        dWaug = None
        if ctx.needs_input_grad[2] and R == 0:
            dWaug = torch.zeros_like(Waug)
        elif ctx.needs_input_grad[2]:
            from spconv import ops as sp_ops
            dWaug = sp_ops._dw(hx_all[:R * S], None, None, dU_all, 1, 32, Waug.shape[1]).view(32, Waug.shape[1])
        need = ctx.needs_input_grad
        return (d_hx if need[0] else None, dh if need[1] else None, dWaug,
                *[g if need[3 + j] else None for j, g in enumerate(dgp)], None, None, None, None)


def ecc_gru_loop(hx, h, Waug, cell, csr_src, csr_dst, repeats, cat_all):
    ig = cell._modules["ig"]
    return _EccGruLoop.apply(hx, h, Waug, ig.weight, ig.bias, cell.weight_ih, cell.weight_hh, cell.bias_ih,
                             cell.bias_hh, csr_src, csr_dst, int(repeats), bool(cat_all))


# ---- point-level Linear with a split-K weight gradient ----------------------------------------------------------

_TALL_MIN_ROWS = int(__import__("os").environ.get("WSIS_TALL_MIN_ROWS", "1024"))


def colsum(x):
    """x.sum(0) of a contiguous fp32 [M, C] device tensor in one fixed-order launch (C % 4 == 0, C <= 1024)"""
    M, C = x.shape
    if not x.is_cuda or C % 4 != 0 or C > 1024 or C < 4 or x.dtype != torch.float32 or not x.is_contiguous():
        return x.sum(0)
    lib = _n.hip()
    ws_bytes = lib.wsis_colsum_workspace_bytes(M, C)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
    out = torch.empty(C, dtype=torch.float32, device=x.device)
    _n.check(lib.wsis_colsum(_n.ptr(x), M, C, _n.ptr(out), _n.ptr(ws), ws_bytes, _n.ptr(_n.sync_block(x.device)),
                             _n.stream_ptr()), "colsum")
    return out


class _TallLinear(Function):
    """y = x @ W^T + b for x [N, Cin] with N ~ 2*10^5 points (backbone_3D_WSIS.py:59-64, 182).  hipBLASLt runs
    the weight gradient X^T dY (a [Cin x Cout] output reduced over N rows) as ONE workgroup (~0.4 ms); here it
    goes through the chunked MFMA reduction of the sparse-conv dW kernel (K = 1, dense rows)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return torch.addmm(bias, x, weight.t()) if bias is not None else x @ weight.t()

    @staticmethod
    def backward(ctx, dy):
        from spconv import ops as sp_ops
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        dx = dy @ weight if ctx.needs_input_grad[0] else None
        dw = None
        if ctx.needs_input_grad[1]:
            # the dW kernel takes Cin % 32 == 0 and Cout % 4 == 0: narrow layers (the 3 -> 16 -> 1 position MLP, the
            # 13-wide first layer of the filter net over the edge rows) are zero-padded for this product only --
            # hipBLASLt runs their [Cin x rows] @ [rows x Cout] gradient as one 16x16 tile per workgroup (43-87 us)
            cout, cin = weight.shape
            if cout % 32 == 0 and cin % 4 == 0:
                # dW in the weight's own [out, in] layout straight from the kernel (dy as the "input" side): no transposed
                # view for autograd to copy into the .grad tensor
                dw = sp_ops._dw(dy, None, None, x.contiguous(), 1, cout, cin).view(cout, cin)
            else:
                cin_p, cout_p = (cin + 31) // 32 * 32, (cout + 3) // 4 * 4
                xp = x.contiguous() if cin_p == cin else torch.nn.functional.pad(x, (0, cin_p - cin))
                dyp = dy if cout_p == cout else torch.nn.functional.pad(dy, (0, cout_p - cout))
                dw = sp_ops._dw(xp, None, None, dyp, 1, cin_p, cout_p).view(cin_p, cout_p)[:cin, :cout].t()
        db = None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = colsum(dy)
        return dx, dw, db


def tall_linear(x, linear):
    # rows from which the row-split dW reduction beats the one-workgroup GEMM (measured at 2.3 k rows: 15 -> 7 us)
    if not x.is_cuda or x.dim() != 2 or x.shape[0] < _TALL_MIN_ROWS or x.dtype != torch.float32:
        return linear(x)
    return _TallLinear.apply(x, linear.weight, linear.bias)


def tall_sequential(seq, x):
    """an nn.Sequential of Linear / activation layers over many rows: the Linear layers through tall_linear"""
    for m in seq:
        x = tall_linear(x, m) if type(m) is torch.nn.Linear else m(x)
    return x


# ---- the superpoint-level heads as one operator ----------------------------------------------------------------------

class _SpHeads(Function):
    """All heads Linear(64,64) -> BatchNorm1d -> ReLU -> Linear(64,cout) and bias-free Linear(64,64) layers that read the
    same [S,64] rows (backbone_3D_WSIS.py:59-64, 195-216, 253) as ONE autograd node: three launches forward, four
    backward (csrc/heads.hip) instead of a module chain per head (4 launches forward and ~12 backward each, plus the
    gradient accumulation of the shared input)."""

    @staticmethod
    def forward(ctx, x, meta, *tensors):
        n_heads, n_lin, couts, eps, momentum, training, running = meta
        _n.require_cuda(x, *tensors)
        ctx.set_materialize_grads(False)       # an output nobody consumed arrives as None (NULL = zero in the library)
        lib = _n.hip()
        x = x.contiguous().float()
        S = x.shape[0]
        nb = n_heads + n_lin
        hidden = torch.empty((nb, S, 64), dtype=torch.float32, device=x.device)
        saved = torch.empty((max(n_heads, 1), 2, 64), dtype=torch.float32, device=x.device)
        outs = [torch.empty((S, couts[p]), dtype=torch.float32, device=x.device) for p in range(n_heads)]
        h = _n.Heads()
        h.n_heads, h.n_lin = n_heads, n_lin
        for p in range(n_heads):
            W1, b1, g, b, W2, b2 = tensors[6 * p:6 * p + 6]
            h.cout[p] = couts[p]
            h.W1[p], h.b1[p], h.gamma[p], h.beta[p] = _n.ptr(W1), _n.ptr(b1), _n.ptr(g), _n.ptr(b)
            h.W2[p], h.b2[p] = _n.ptr(W2), _n.ptr(b2)
            rm, rv = running[p]
            h.running_mean[p], h.running_var[p] = _n.ptr(rm), _n.ptr(rv)
            h.out[p] = _n.ptr(outs[p])
        for q in range(n_lin):
            h.W1[n_heads + q] = _n.ptr(tensors[6 * n_heads + q])
        for p in range(nb):
            h.hidden[p] = hidden[p].data_ptr()
        ws_bytes = lib.wsis_heads_workspace_bytes(S, n_heads, n_lin)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
        _n.check(lib.wsis_heads_fwd(ctypes.byref(h), _n.ptr(x), S, float(eps), float(momentum), int(training),
                                    _n.ptr(saved), _n.ptr(ws), ws_bytes, _n.stream_ptr()), "heads_fwd")
        ctx.save_for_backward(x, hidden, saved, *tensors)
        ctx.meta = (n_heads, n_lin, couts, training)
        return tuple(outs) + tuple(hidden[n_heads + q] for q in range(n_lin))

    @staticmethod
    def backward(ctx, *grads):
        n_heads, n_lin, couts, training = ctx.meta
        x, hidden, saved = ctx.saved_tensors[:3]
        tensors = ctx.saved_tensors[3:]
        lib = _n.hip()
        S = x.shape[0]
        nb = n_heads + n_lin
        keep = []
        h = _n.Heads()
        h.n_heads, h.n_lin = n_heads, n_lin
        out_grads = [None] * len(tensors)

        def want(i):
            if not ctx.needs_input_grad[2 + i]:
                return None
            out_grads[i] = torch.empty_like(tensors[i], memory_format=torch.contiguous_format)
            return out_grads[i].data_ptr()

        for p in range(nb):
            g = grads[p]
            if g is not None:
                g = g.contiguous().float()
                keep.append(g)
            h.dout[p] = _n.ptr(g)
            h.hidden[p] = hidden[p].data_ptr()
        for p in range(n_heads):
            W1, b1, gam, bet, W2, b2 = tensors[6 * p:6 * p + 6]
            h.cout[p] = couts[p]
            h.W1[p], h.b1[p], h.gamma[p], h.beta[p] = _n.ptr(W1), _n.ptr(b1), _n.ptr(gam), _n.ptr(bet)
            h.W2[p], h.b2[p] = _n.ptr(W2), _n.ptr(b2)
            h.dW1[p], h.db1[p], h.dgamma[p] = want(6 * p), want(6 * p + 1), want(6 * p + 2)
            h.dbeta[p], h.dW2[p], h.db2[p] = want(6 * p + 3), want(6 * p + 4), want(6 * p + 5)
        for q in range(n_lin):
            h.W1[n_heads + q] = _n.ptr(tensors[6 * n_heads + q])
            h.dW1[n_heads + q] = want(6 * n_heads + q)
        dx = torch.empty_like(x)
        ws_bytes = lib.wsis_heads_workspace_bytes(S, n_heads, n_lin)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
        _n.check(lib.wsis_heads_bwd(ctypes.byref(h), _n.ptr(x), S, int(training), _n.ptr(saved), _n.ptr(dx), _n.ptr(ws),
                                    ws_bytes, _n.stream_ptr()), "heads_bwd")
        return (dx if ctx.needs_input_grad[0] else None, None) + tuple(out_grads)


def _head_parts(head):
    """(Linear, BatchNorm1d, Linear) of a ``head(cin, cout)`` Sequential that the fused operator covers, else None"""
    if len(head) != 4:
        return None
    l1, bn, act, l2 = head[0], head[1], head[2], head[3]
    if not (type(l1) is torch.nn.Linear and type(l2) is torch.nn.Linear and isinstance(bn, torch.nn.BatchNorm1d)
            and isinstance(act, torch.nn.ReLU)):
        return None
    if l1.in_features != 64 or l1.out_features != 64 or l1.bias is None or l2.in_features != 64 or l2.out_features > 32:
        return None
    if not bn.affine or bn.num_features != 64 or bn.momentum is None or sync_group(bn) is not None:
        return None
    if not bn.training and not bn.track_running_stats:
        return None
    return l1, bn, l2


def sp_heads(x, heads, linears=()):
    """outputs of the ``head`` Sequentials and of the bias-free Linear(64,64) layers over x [S,64]; None where the fused
    operator does not apply (CPU tensors, other widths, SyncBatchNorm, mixed train / eval heads, WSIS_FUSE_HEADS=0): the
    caller then runs the modules one by one"""
    import os
    if (not x.is_cuda or x.dim() != 2 or x.shape[1] != 64 or x.dtype != torch.float32 or x.shape[0] < 1
            or len(heads) + len(linears) > _n.HEADS_MAX or os.environ.get("WSIS_FUSE_HEADS", "1") == "0"):
        return None
    parts = [_head_parts(hd) for hd in heads]
    if any(p is None for p in parts):
        return None
    if any(type(l) is not torch.nn.Linear or l.bias is not None or l.in_features != 64 or l.out_features != 64
           for l in linears):
        return None
    modes = {bn.training for _, bn, _ in parts}
    if len(modes) > 1 or len({(bn.eps, bn.momentum) for _, bn, _ in parts}) > 1:
        return None
    training = bool(modes.pop()) if modes else False
    eps, momentum = (parts[0][1].eps, parts[0][1].momentum) if parts else (1e-5, 0.1)
    tensors, running, couts = [], [], []
    for l1, bn, l2 in parts:
        tensors += [l1.weight, l1.bias, bn.weight, bn.bias, l2.weight, l2.bias]
        couts.append(l2.out_features)
        track = bn.track_running_stats and bn.running_mean is not None
        running.append((bn.running_mean, bn.running_var) if track else (None, None))
        if training and track and bn.num_batches_tracked is not None:
            _defer_batch_count(bn)
    tensors += [l.weight for l in linears]
    meta = (len(parts), len(linears), tuple(couts), eps, momentum, training, tuple(running))
    outs = _SpHeads.apply(x, meta, *tensors)
    return list(outs[:len(parts)]), list(outs[len(parts):])


# ---- a14: voxel -> point gather with a deterministic backward ---------------------------------------------------

class _GatherRows(Function):
    """out[p] = src[idx[p]]  (backbone_3D_WSIS.py:179 ``output.features[input_map.long()]``).  Backward is the
    segmented sum over the points of each voxel (CSR of idx, ascending point order) instead of an atomic
    index-add: deterministic."""

    @staticmethod
    def forward(ctx, src, idx, csr):
        _n.require_cuda(src, idx)
        src = src.contiguous().float()
        idx = idx.contiguous()
        assert idx.dtype in (torch.int32, torch.int64)
        N, C = idx.numel(), src.shape[1]
        out = torch.empty((N, C), dtype=torch.float32, device=src.device)
        _n.check(_n.hip().wsis_gather_rows(_n.ptr(src), _n.ptr(idx), int(idx.dtype == torch.int64), _n.ptr(out), N, C,
                                           _n.stream_ptr()), "gather_rows")
        ctx.csr = csr
        ctx.rows = src.shape[0]
        return out

    @staticmethod
    def backward(ctx, dout):
        csr = ctx.csr
        dout = dout.contiguous().float()
        C = dout.shape[1]
        dsrc = torch.empty((csr.S, C), dtype=torch.float32, device=dout.device)
        _n.check(_n.hip().wsis_segment_reduce_fwd(_n.ptr(dout), _n.ptr(csr.perm), _n.ptr(csr.offsets), _n.ptr(dsrc),
                                                  None, csr.N, csr.S, C, 0, _n.stream_ptr()), "segment_reduce_fwd")
        assert csr.S == ctx.rows
        return dsrc, None, None


def gather_rows(src, idx, csr=None):
    """src[idx] with a deterministic backward; ``csr`` = SegmentCSR(idx, src.shape[0]) may be passed for reuse"""
    if csr is None:
        csr = SegmentCSR(idx, src.shape[0])
    return _GatherRows.apply(src, idx, csr)


class _SemanticPointLoss(Function):
    """CrossEntropy(ignore_index) + mean per-class dice of the point-level scores in two passes over [N, C]
    (``csrc/loss.hip``; reference ``losses_3D_WSIS.py:52-67``).  Returns (loss, number of kept rows)."""

    @staticmethod
    def forward(ctx, scores, labels, ignore_label):
        _n.require_cuda(scores, labels)
        lib = _n.hip()
        scores = scores.contiguous().float()
        labels = labels.contiguous().long()
        N, C = scores.shape
        out = torch.empty(2, dtype=torch.float32, device=scores.device)
        saved = torch.empty(2 * C + 1, dtype=torch.float32, device=scores.device)
        ws_bytes = lib.wsis_semantic_loss_workspace_bytes(N)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=scores.device)
        _n.check(lib.wsis_semantic_loss_fwd(_n.ptr(scores), _n.ptr(labels), N, C, int(ignore_label), _n.ptr(out),
                                            _n.ptr(saved), _n.ptr(ws), ws_bytes, _n.stream_ptr()),
                 "semantic_loss_fwd")
        ctx.save_for_backward(scores, labels, saved)
        ctx.ignore_label = int(ignore_label)
        loss, n_valid = out[0], out[1]
        ctx.mark_non_differentiable(n_valid)
        return loss, n_valid

    @staticmethod
    def backward(ctx, g_loss, _g_n):
        scores, labels, saved = ctx.saved_tensors
        N, C = scores.shape
        d = torch.empty_like(scores)
        g = g_loss.contiguous().float()
        _n.check(_n.hip().wsis_semantic_loss_bwd(_n.ptr(scores), _n.ptr(labels), N, C, ctx.ignore_label,
                                                 _n.ptr(saved), _n.ptr(g), _n.ptr(d), _n.stream_ptr()),
                 "semantic_loss_bwd")
        return d, None, None


def semantic_point_loss(scores, labels, ignore_label=-100):
    return _SemanticPointLoss.apply(scores, labels, ignore_label)


class _SpCrossEntropy(Function):
    """CrossEntropyLoss(ignore_index) of the superpoint scores + the logged scores.sum() in one launch each way
    (csrc/loss.hip; reference losses_3D_WSIS.py:72-74).  Returns (loss, sum of the scores)."""

    @staticmethod
    def forward(ctx, scores, labels, ignore_label):
        _n.require_cuda(scores, labels)
        scores = scores.contiguous().float()
        labels = labels.contiguous().long()
        S, C = scores.shape
        out = torch.empty(3, dtype=torch.float32, device=scores.device)
        lib = _n.hip()
        ws_bytes = lib.wsis_sp_ce_loss_workspace_bytes(S)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=scores.device)
        _n.check(lib.wsis_sp_ce_loss_fwd(_n.ptr(scores), _n.ptr(labels), S, C, int(ignore_label), _n.ptr(out), _n.ptr(ws),
                                         ws_bytes, _n.ptr(_n.sync_block(scores.device)), _n.stream_ptr()), "sp_ce_loss_fwd")
        ctx.save_for_backward(scores, labels, out)
        ctx.ignore_label = int(ignore_label)
        loss, total = out[0], out[1]
        ctx.mark_non_differentiable(total)
        return loss, total

    @staticmethod
    def backward(ctx, g_loss, _g_total):
        scores, labels, out = ctx.saved_tensors
        S, C = scores.shape
        d = torch.empty_like(scores)
        _n.check(_n.hip().wsis_sp_ce_loss_bwd(_n.ptr(scores), _n.ptr(labels), S, C, ctx.ignore_label, _n.ptr(out),
                                              _n.ptr(g_loss.contiguous().float()), _n.ptr(d), _n.stream_ptr()),
                 "sp_ce_loss_bwd")
        return d, None, None


def superpoint_cross_entropy(scores, labels, ignore_label=-100):
    return _SpCrossEntropy.apply(scores, labels, ignore_label)


class _LossSum(Function):
    """t_0 + t_1 + ... of device scalars in order, one launch (each `loss = loss + term` is a launch otherwise); the
    gradient of a sum is the upstream scalar for every term"""

    @staticmethod
    def forward(ctx, paired, *terms):
        ts = [t.reshape(()).float() for t in terms]
        _n.require_cuda(*ts)
        out = torch.empty((), dtype=torch.float32, device=ts[0].device)
        arr = (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
        _n.check(_n.hip().wsis_loss_sum(arr, len(ts), int(paired), _n.ptr(out), _n.stream_ptr()), "loss_sum")
        ctx.meta = [(t.shape, t.dtype) for t in terms]      # a term may be [1] or another float type: its gradient too
        return out

    @staticmethod
    def backward(ctx, g):
        return (None,) + tuple(g.reshape(shape).to(dtype) for shape, dtype in ctx.meta)


def loss_sum(terms, paired=0):
    """sum of 1..8 device scalars in list order; bit i of ``paired`` groups (t_i + t_{i+1}) first"""
    return _LossSum.apply(paired, *terms)


class _SpRegressionLoss(Function):
    """offset L1 + cosine, occupancy L1 and instance-size L1 over the labelled superpoints in one launch
    (``csrc/loss.hip``; reference ``losses_3D_WSIS.py:79-96,113-127``).  Returns the four losses and the row count."""

    @staticmethod
    def forward(ctx, pred_off, gt_off, pred_occ, gt_occ, pred_size, gt_size, sem_label, ins_label, ignore_label):
        _n.require_cuda(pred_off, gt_off, pred_occ, gt_occ, pred_size, gt_size, sem_label, ins_label)
        f = lambda t: t.contiguous().float()
        pred_off, gt_off, pred_occ, gt_occ, pred_size, gt_size = (f(pred_off), f(gt_off), f(pred_occ), f(gt_occ),
                                                                  f(pred_size), f(gt_size))
        sem_label, ins_label = sem_label.contiguous().long(), ins_label.contiguous().long()
        S = pred_off.shape[0]
        assert pred_off.shape == (S, 3) and gt_off.shape == (S, 3) and pred_occ.numel() == S and pred_size.numel() == S
        out = torch.empty(5, dtype=torch.float32, device=pred_off.device)
        _n.check(_n.hip().wsis_sp_regression_loss_fwd(
            _n.ptr(pred_off), _n.ptr(gt_off), _n.ptr(pred_occ), _n.ptr(gt_occ), _n.ptr(pred_size), _n.ptr(gt_size),
            _n.ptr(sem_label), _n.ptr(ins_label), S, int(ignore_label), _n.ptr(out), _n.stream_ptr()),
            "sp_regression_loss_fwd")
        ctx.save_for_backward(pred_off, gt_off, pred_occ, gt_occ, pred_size, gt_size, sem_label, ins_label, out)
        ctx.ignore_label = int(ignore_label)
        ctx.shapes = (pred_occ.shape, pred_size.shape)
        n_valid = out[4]
        ctx.mark_non_differentiable(n_valid)
        return out[0], out[1], out[2], out[3], n_valid

    @staticmethod
    def backward(ctx, g_norm, g_dir, g_occ, g_size, _g_n):
        pred_off, gt_off, pred_occ, gt_occ, pred_size, gt_size, sem_label, ins_label, out = ctx.saved_tensors
        S = pred_off.shape[0]
        d_off = torch.empty_like(pred_off)
        d_occ = torch.empty_like(pred_occ)
        d_size = torch.empty_like(pred_size)
        g = [t.contiguous().float() for t in (g_norm, g_dir, g_occ, g_size)]
        _n.check(_n.hip().wsis_sp_regression_loss_bwd(
            _n.ptr(pred_off), _n.ptr(gt_off), _n.ptr(pred_occ), _n.ptr(gt_occ), _n.ptr(pred_size), _n.ptr(gt_size),
            _n.ptr(sem_label), _n.ptr(ins_label), S, ctx.ignore_label, _n.ptr(out), _n.ptr(g[0]), _n.ptr(g[1]),
            _n.ptr(g[2]), _n.ptr(g[3]), _n.ptr(d_off), _n.ptr(d_occ), _n.ptr(d_size), _n.stream_ptr()),
            "sp_regression_loss_bwd")
        return d_off, None, d_occ, None, d_size, None, None, None, None


def sp_regression_losses(pred_off, gt_off, pred_occ, gt_occ, pred_size, gt_size, sem_label, ins_label,
                         ignore_label=-100):
    return _SpRegressionLoss.apply(pred_off, gt_off, pred_occ, gt_occ, pred_size, gt_size, sem_label, ins_label,
                                   ignore_label)


DISC_MAX_ROWS, DISC_MAX_SLOTS = 4096, 64


class _DiscriminativeLoss(Function):
    """pull / push / regularisation loss of one scene's superpoint embeddings in one launch each way
    (``csrc/loss.hip``; reference ``losses_3D_WSIS.py:157-230``), instances in host-bounded slots."""

    @staticmethod
    def forward(ctx, x, ins_label, sem_label, n_slots, ignore_label, delta_v, delta_d, p_var, p_dist, p_reg):
        _n.require_cuda(x, ins_label, sem_label)
        lib = _n.hip()
        x = x.contiguous().float()
        ins_label, sem_label = ins_label.contiguous().long(), sem_label.contiguous().long()
        S, D = x.shape
        out = torch.empty(1, dtype=torch.float32, device=x.device)
        saved = torch.empty(lib.wsis_disc_loss_saved_floats(), dtype=torch.float32, device=x.device)
        ctx.args = (S, D, int(n_slots), int(ignore_label), float(delta_v), float(delta_d), float(p_var), float(p_dist),
                    float(p_reg))
        _n.check(lib.wsis_disc_loss_fwd(_n.ptr(x), _n.ptr(ins_label), _n.ptr(sem_label), *ctx.args, _n.ptr(out),
                                        _n.ptr(saved), _n.stream_ptr()), "disc_loss_fwd")
        ctx.save_for_backward(x, ins_label, sem_label, saved)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        x, ins_label, sem_label, saved = ctx.saved_tensors
        dx = torch.empty_like(x)
        g = g.contiguous().float()
        _n.check(_n.hip().wsis_disc_loss_bwd(_n.ptr(x), _n.ptr(ins_label), _n.ptr(sem_label), *ctx.args,
                                             _n.ptr(saved), _n.ptr(g), _n.ptr(dx), _n.stream_ptr()), "disc_loss_bwd")
        return (dx,) + (None,) * 9


def discriminative_loss(x, ins_label, sem_label, n_slots, ignore_label=-100, delta_v=0.1, delta_d=1.5, p_var=1.0,
                        p_dist=1.0, p_reg=0.001):
    return _DiscriminativeLoss.apply(x, ins_label, sem_label, n_slots, ignore_label, delta_v, delta_d, p_var, p_dist,
                                     p_reg)
