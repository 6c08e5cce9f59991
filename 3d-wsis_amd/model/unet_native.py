"""Native execution of the sparse UNet (input_conv -> UBlock -> output_layer) through the op-list executor.

The module tree of ``Network`` (same parameters, same state-dict) is walked ONCE per pass on the host to record the
forward pass as ``wsis_op`` records (include/wsis_hip.h: device pointers + sizes) into a numpy array; one C call
(``wsis_run_ops``) then issues every kernel.  The backward pass is recorded the same way from the closures the
forward walk leaves behind (a hand-written tape for the four op kinds of the UNet) and runs as a second C call.
Numerically this is the same kernel sequence the per-module path (spconv.SubMConv3d / BatchNorm / ...) launches,
so both paths agree bit for bit; what disappears is ~450 Python-dispatched autograd nodes per step
(sparse_unet3d.py:103-350 walked by torch in the reference).

Activations live in one arena per pass (bump allocation, nothing is freed before the backward: 288 GB of HBM make
rematerialisation pointless at these sizes), parameter gradients in one flat buffer whose views become ``p.grad``.
"""
import numpy as np
import torch
from torch import nn
from torch.autograd import Function

import spconv
import wsis_native as _n
import wsis_ops
from spconv import ops as sp_ops

OP_CONV, OP_BN_RELU, OP_CAT, OP_SPLIT, OP_ADD, OP_CONV_BWD, OP_BN_RELU_BWD = 1, 2, 3, 4, 5, 6, 7
F_RELU, F_TRAINING, F_UPDATE, F_FLIP = 1, 2, 4, 8

OP_DTYPE = np.dtype([("kind", "<i4"), ("flags", "<i4"), ("M_in", "<i8"), ("M_out", "<i8"), ("K", "<i4"),
                     ("Cin", "<i4"), ("Cout", "<i4"), ("reserved", "<i4"), ("eps", "<f4"), ("momentum", "<f4"),
                     ("inp", "<u8", (8,)), ("out", "<u8", (4,))])
assert OP_DTYPE.itemsize == 144          # sizeof(wsis_op)

_REL = 1 << 62                            # tag: byte offset into the FORWARD arena, resolved at launch
_GREL = 1 << 61                           # tag: byte offset into the BACKWARD arena
_PREL = 1 << 60                           # tag: byte offset into the flat PARAMETER-GRADIENT buffer


class _Recorder(object):
    """op list + bump allocator of one pass (``tag`` marks the arena its allocations live in)"""

    def __init__(self, tag, align=256):
        self.rows = []
        self.bytes = 0
        self.tag = tag
        self.align = align

    def alloc(self, n_floats):
        off = self.bytes
        self.bytes += (int(n_floats) * 4 + self.align - 1) // self.align * self.align
        return self.tag | off

    def op(self, kind, flags=0, M_in=0, M_out=0, K=0, Cin=0, Cout=0, eps=0.0, momentum=0.0, inp=(), out=()):
        self.rows.append((kind, flags, M_in, M_out, K, Cin, Cout, eps, momentum, inp, out))

    def finish(self, bases):
        """-> wsis_op array with the tagged pointers resolved; ``bases`` = {tag: arena base address}"""
        n = len(self.rows)
        arr = np.zeros(n, dtype=OP_DTYPE)
        inp = np.zeros((n, 8), dtype=np.uint64)
        out = np.zeros((n, 4), dtype=np.uint64)
        for i, r in enumerate(self.rows):
            arr[i] = (r[0], r[1], r[2], r[3], r[4], r[5], r[6], 0, r[7], r[8], 0, 0)
            if r[9]:
                inp[i, :len(r[9])] = r[9]
            if r[10]:
                out[i, :len(r[10])] = r[10]
        for a in (inp, out):
            for tag, base in bases.items():
                t = np.uint64(tag)
                m = (a & t) != 0
                a[m] = (a[m] ^ t) + np.uint64(base)
        arr["inp"] = inp
        arr["out"] = out
        return arr


def _ptr(t):
    return 0 if t is None else t.data_ptr()


class _Table(object):
    """gather tables of one conv: forward (rows = outputs) and dIn (rows = inputs) + flip flag"""

    def __init__(self, nbr_f, order_f, nbr_b, order_b, flip):
        self.nbr_f, self.order_f, self.nbr_b, self.order_b, self.flip = nbr_f, order_f, nbr_b, order_b, flip


class UNetProgram(object):
    """records forward and backward of input_conv + unet + output_layer of a ``Network``"""

    def __init__(self, net):
        self.net = net
        self.params = [p for m in (net.input_conv, net.unet, net.output_layer) for p in m.parameters()]
        self.flat_grad, self.flat_params = None, []

    # ---- forward recording: every helper returns (out_handle, backward_closure) -----------------------------
    def _conv(self, rec, x, conv, table, M_in, M_out, residual=0):
        K = int(np.prod(conv.kernel_size))
        Cin, Cout = conv.in_channels, conv.out_channels
        W = conv.weight
        assert conv.bias is None, "the UNet convolutions carry no bias (sparse_unet3d.py)"
        y = rec.alloc(M_out * Cout)
        t = table if table is not None else _Table(None, None, None, None, 0)
        rec.op(OP_CONV, 0, M_in, M_out, K, Cin, Cout, inp=(x, _ptr(t.nbr_f), _ptr(t.order_f), W.data_ptr(), 0, residual),
               out=(y,))
        self._account("spconv_fwd_kernel", t.nbr_f, M_out, Cin, Cout)

        def bwd(recb, dy, need_dx=True):
            dx = recb.alloc(M_in * Cin) if need_dx else 0
            dW = self._grad_handle(recb, W)
            recb.op(OP_CONV_BWD, F_FLIP if t.flip else 0, M_in, M_out, K, Cin, Cout,
                    inp=(x, W.data_ptr(), dy, _ptr(t.nbr_f), _ptr(t.order_f), _ptr(t.nbr_b), _ptr(t.order_b)),
                    out=(dx, dW))
            if need_dx:
                self._account("spconv_fwd_kernel", t.nbr_b, M_in, Cout, Cin)
            self._account("spconv_dw_kernel", t.nbr_f, M_out, Cin, Cout)
            return dx
        return y, bwd

    def _bn_relu(self, rec, x, bn, M, relu=True):
        C = bn.num_features
        training = bn.training or not bn.track_running_stats
        update = bn.training and bn.track_running_stats
        flags = (F_RELU if relu else 0) | (F_TRAINING if training else 0) | (F_UPDATE if update else 0)
        if update and bn.num_batches_tracked is not None:
            assert bn.momentum is not None, "cumulative-average BatchNorm is not used by 3D-WSIS"
            wsis_ops._defer_batch_count(bn)
        y = rec.alloc(M * C)
        mean = rec.alloc(C) if training else bn.running_mean.data_ptr()
        var = rec.alloc(C) if training else bn.running_var.data_ptr()
        rec.op(OP_BN_RELU, flags, M, M, 0, C, C, bn.eps, bn.momentum if bn.momentum is not None else 0.1,
               inp=(x, _ptr(bn.weight), _ptr(bn.bias), _ptr(bn.running_mean), _ptr(bn.running_var)),
               out=(y, mean if training else 0, var if training else 0))

        def bwd(recb, dy, addend=0):
            # ``addend``: gradient arriving at x over a second path (residual skip / UNet skip connection), added
            # in the same pass instead of a separate accumulation kernel
            dx = recb.alloc(M * C)
            dg = self._grad_handle(recb, bn.weight) if bn.weight is not None else recb.alloc(C)
            db = self._grad_handle(recb, bn.bias) if bn.bias is not None else recb.alloc(C)
            recb.op(OP_BN_RELU_BWD, flags, M, M, 0, C, C, bn.eps, 0.0,
                    inp=(x, dy, mean, var, _ptr(bn.weight), _ptr(bn.bias), addend), out=(dx, dg, db))
            return dx
        return y, bwd

    def _residual_block(self, rec, x, blk, table, M):
        seq = blk.conv_branch
        bn1, conv1, bn2, conv2 = seq[0], seq[2], seq[3], seq[5]
        a1, b_bn1 = self._bn_relu(rec, x, bn1, M)
        z1, b_c1 = self._conv(rec, a1, conv1, table, M, M)
        a2, b_bn2 = self._bn_relu(rec, z1, bn2, M)
        first = blk.i_branch[0]
        if isinstance(first, nn.Identity):
            res, b_i = x, None
        else:
            res, b_i = self._conv(rec, x, first, None, M, M)          # 1x1 projection of the skip path
        out, b_c2 = self._conv(rec, a2, conv2, table, M, M, residual=res)

        def bwd(recb, d_out):
            d_a2 = b_c2(recb, d_out)
            d_z1 = b_bn2(recb, d_a2)
            d_a1 = b_c1(recb, d_z1)
            d_skip = d_out if b_i is None else b_i(recb, d_out)
            return b_bn1(recb, d_a1, addend=d_skip)
        return out, bwd

    def _ublock(self, rec, x, ub, lvl):
        M = self.M[lvl]
        subm = self.subm[lvl]
        bwds = []
        for blk in ub.blocks:
            x, b = self._residual_block(rec, x, blk, subm, M)
            bwds.append(b)
        if len(ub.nPlanes) == 1:
            def bwd_leaf(recb, d):
                for b in reversed(bwds):
                    d = b(recb, d)
                return d
            return x, bwd_leaf
        C0, C1 = ub.nPlanes[0], ub.nPlanes[1]
        M1 = self.M[lvl + 1]
        identity = x
        a, b_bn = self._bn_relu(rec, x, ub.conv[0], M)
        d, b_down = self._conv(rec, a, ub.conv[2], self.down[lvl], M, M1)
        u, b_u = self._ublock(rec, d, ub.u, lvl + 1)
        a2, b_bn2 = self._bn_relu(rec, u, ub.deconv[0], M1)
        up, b_up = self._conv(rec, a2, ub.deconv[2], self.up[lvl], M1, M)
        cat = rec.alloc(M * 2 * C0)
        rec.op(OP_CAT, 0, M, M, 0, C0, C0, inp=(identity, up), out=(cat,))
        x = cat
        tails = []
        for blk in ub.blocks_tail:
            x, b = self._residual_block(rec, x, blk, subm, M)
            tails.append(b)

        def bwd(recb, dcur):
            for b in reversed(tails):
                dcur = b(recb, dcur)
            d_id = recb.alloc(M * C0)
            d_up = recb.alloc(M * C0)
            recb.op(OP_SPLIT, 0, M, M, 0, C0, C0, inp=(dcur,), out=(d_id, d_up))
            d_a2 = b_up(recb, d_up)
            d_u = b_bn2(recb, d_a2)
            d_d = b_u(recb, d_u)
            d_a = b_down(recb, d_d)
            d_x = b_bn(recb, d_a, addend=d_id)
            for b in reversed(bwds):
                d_x = b(recb, d_x)
            return d_x
        return x, bwd

    # ---- bookkeeping ----------------------------------------------------------------------------------------
    def _grad_handle(self, recb, p):
        # parameter gradients live densely in their own flat buffer (one all-reduce for data parallelism)
        h = self._grad.get(id(p))
        if h is None:
            h = self._palloc.alloc(p.numel())
            self._grad[id(p)] = h
        return h

    def _account(self, name, nbr, M_out, Cin, Cout):
        prof = sp_ops.PROFILER
        if prof is not None:
            P = prof.pairs(nbr, M_out)
            prof.end(name, None, P * (Cin + Cout) * 4 + P * 8, 2 * P * Cin * Cout)

    def bind(self, tensor):
        """rulebooks of the pyramid of ``tensor`` (built if missing) -> per-level tables and row counts"""
        net = self.net
        sp_ops.prebuild_unet_rulebooks(tensor, net.blocks)
        self.M, self.subm, self.down, self.up = [], [], [], []
        for lvl in range(net.blocks):
            rb = tensor.indice_dict["subm%d" % (lvl + 1)]
            self.M.append(int(rb.in_indices.shape[0]))
            self.subm.append(_Table(rb.nbr_p, rb.order, rb.nbr_p, rb.order, 1))
            if lvl + 1 < net.blocks:
                rd = tensor.indice_dict["spconv%d" % (lvl + 1)]
                self.down.append(_Table(rd.nbr_p, rd.order, rd.nbr_up_p, rd.order_up, 0))
                self.up.append(_Table(rd.nbr_up_p, rd.order_up, rd.nbr_p, rd.order, 0))
        self.keep = [tensor.indice_dict]     # the tables must outlive the backward pass

    def record_forward(self, x_ptr, Cin0):
        net = self.net
        rec = _Recorder(_REL)
        M0 = self.M[0]
        y, b_in = self._conv(rec, x_ptr, net.input_conv[0], self.subm[0], M0, M0)
        y, b_u = self._ublock(rec, y, net.unet, 0)
        out, b_out = self._bn_relu(rec, y, net.output_layer[0], M0)

        def backward(recb, d_out, need_dx):
            d = b_out(recb, d_out)
            d = b_u(recb, d)
            return b_in(recb, d, need_dx)
        return rec, out, backward


def _arena(nbytes, device):
    t = torch.empty(nbytes + 256, dtype=torch.uint8, device=device)
    return t, (t.data_ptr() + 255) // 256 * 256


def _view(arena, base, tag, handle, shape):
    numel = int(np.prod(shape))
    off = (handle ^ tag) + base - arena.data_ptr()
    return arena[off:off + numel * 4].view(torch.float32).view(shape)


class UNetFunction(Function):
    """features [M0, Cin] -> [M0, m]; ``prog`` is a bound UNetProgram, ``params`` its parameters (so that autograd
    routes their gradients)."""

    @staticmethod
    def forward(ctx, x, prog, *params):
        _n.require_cuda(x)
        x = x if (x.dtype == torch.float32 and x.is_contiguous()) else x.contiguous().float()
        rec, out_h, backward = prog.record_forward(x.data_ptr(), x.shape[1])
        arena, base = _arena(rec.bytes, x.device)
        _run(_n.hip(), rec.finish({_REL: base}), x.device)
        out = _view(arena, base, _REL, out_h, (prog.M[0], prog.net.output_layer[0].num_features))
        ctx.prog, ctx.record_backward, ctx.arena, ctx.base, ctx.x = prog, backward, arena, base, x
        ctx.keep = prog.keep
        return out

    @staticmethod
    def backward(ctx, d_out):
        prog = ctx.prog
        d_out = d_out if (d_out.dtype == torch.float32 and d_out.is_contiguous()) else d_out.contiguous().float()
        recb = _Recorder(_GREL)
        prog._grad = {}
        prog._palloc = _Recorder(_PREL, align=16)
        need_dx = ctx.needs_input_grad[0]
        dx_h = ctx.record_backward(recb, d_out.data_ptr(), need_dx)
        garena, gbase = _arena(recb.bytes, d_out.device)
        pflat = torch.zeros(prog._palloc.bytes // 4 + 64, dtype=torch.float32, device=d_out.device)
        pbase = (pflat.data_ptr() + 255) // 256 * 256
        _run(_n.hip(), recb.finish({_REL: ctx.base, _GREL: gbase, _PREL: pbase}), d_out.device)
        first = (pbase - pflat.data_ptr()) // 4
        grads, covered = [], []
        for i, p in enumerate(prog.params):
            h = prog._grad.get(id(p))
            if h is None or not ctx.needs_input_grad[2 + i]:
                grads.append(None)
            else:
                off = first + (h ^ _PREL) // 4
                grads.append(pflat[off:off + p.numel()].view(p.shape))
                covered.append(p)
        # data-parallel hook: the gradients of ``flat_params`` are views of ``flat_grad`` (parallel.GradSync)
        prog.flat_grad, prog.flat_params = pflat, covered
        dx = _view(garena, gbase, _GREL, dx_h, tuple(ctx.x.shape)) if need_dx else None
        return (dx, None) + tuple(grads)


def _run(lib, ops, device):
    n = len(ops)
    p = ops.ctypes.data
    ws_bytes = lib.wsis_run_ops_workspace_bytes(p, n)
    if ws_bytes < 0:
        raise _n.WsisError("run_ops workspace query failed")
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device)
    _n.check(lib.wsis_run_ops(p, n, _n.ptr(ws), ws_bytes, _n.stream_ptr()), "run_ops")


def run_unet(net, input_tensor):
    """input_conv + unet + output_layer of ``net`` on ``input_tensor`` (SparseConvTensor) -> features [M0, m]"""
    prog = getattr(net, "_native_prog", None)
    if prog is None:
        prog = UNetProgram(net)
        net._native_prog = prog
    prog.bind(input_tensor)
    return UNetFunction.apply(input_tensor.features, prog, *prog.params)
