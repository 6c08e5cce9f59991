"""Native execution of the sparse UNet (input_conv -> UBlock -> output_layer) through the op-list executor.

The module tree of ``Network`` (same parameters, same state-dict) is walked ONCE per (model, mode) to record the
forward pass and -- from the closures the forward walk leaves behind, a hand-written tape for the four op kinds of
the UNet -- the backward pass as SYMBOLIC ``wsis_op`` templates (include/wsis_hip.h: device pointers + sizes): row
counts are level indices, activations are arena allocation ids, gather tables are slots.  Per step the templates are
instantiated with a handful of vectorised numpy assignments (row counts of the scene's pyramid, table pointers,
arena bases) and ONE C call per pass (``wsis_run_ops``) issues every kernel.
Numerically this is the same kernel sequence the per-module path (spconv.SubMConv3d / BatchNorm / ...) launches,
so both paths agree bit for bit; what disappears is ~450 Python-dispatched autograd nodes per step
(sparse_unet3d.py:103-350 walked by torch in the reference).

Activations live in one arena per pass (bump allocation, nothing is freed before the backward: 288 GB of HBM make
rematerialisation pointless at these sizes), parameter gradients densely in one flat buffer whose views become
``p.grad`` (and which ``parallel.GradSync`` all-reduces in place).
"""
import os

import numpy as np
import torch
from torch import nn
from torch.autograd import Function

import wsis_native as _n
import wsis_ops
from spconv import ops as sp_ops

OP_CONV, OP_BN_RELU, OP_CAT, OP_SPLIT, OP_ADD, OP_CONV_BWD, OP_BN_RELU_BWD = 1, 2, 3, 4, 5, 6, 7
F_RELU, F_TRAINING, F_UPDATE, F_FLIP, F_STATS, F_BN_IN, F_STAT_FIN = 1, 2, 4, 8, 16, 32, 64
N_IN, N_OUT = 12, 8

OP_DTYPE = np.dtype([("kind", "<i4"), ("flags", "<i4"), ("M_in", "<i8"), ("M_out", "<i8"), ("K", "<i4"),
                     ("Cin", "<i4"), ("Cout", "<i4"), ("reserved", "<i4"), ("eps", "<f4"), ("momentum", "<f4"),
                     ("momentum2", "<f4"), ("reserved2", "<f4"), ("inp", "<u8", (N_IN,)), ("out", "<u8", (N_OUT,))])
assert OP_DTYPE.itemsize == 216          # sizeof(wsis_op)

# symbolic pointer = tag | id; plain device pointers (parameters, BN buffers) carry no tag
_FWD = 1 << 62      # allocation id in the FORWARD arena
_BWD = 1 << 61      # allocation id in the BACKWARD arena
_PAR = 1 << 60      # allocation id in the flat PARAMETER-GRADIENT buffer
_TBL = 1 << 59      # gather-table slot: level*6 + {0 subm.nbr, 1 subm.order, 2 down.nbr, 3 down.order, 4 up.nbr, 5 up.order}
_EXT = 1 << 58      # external tensor: 0 = input features, 1 = gradient of the output
_TAGS = (_FWD, _BWD, _PAR, _TBL, _EXT)
_ID_MASK = (1 << 32) - 1       # allocation id / slot
_OFF_SHIFT = 32                # bits 32..57: offset into the allocation, in floats (a channel range of a statistics vector)
_OFF_MASK = (1 << 26) - 1


def _at(handle, floats):
    """symbolic pointer ``floats`` floats into an allocation; a plain device pointer is simply advanced"""
    if floats == 0 or handle == 0:
        return handle
    if any(handle & t for t in _TAGS):
        return handle + (int(floats) << _OFF_SHIFT)
    return handle + 4 * int(floats)


def _fuse_fin_enabled():
    """BatchNorm statistics finished inside the producing convolution's launch (last-arrival tickets,
    WSIS_FUSE_BN_FIN=1; default off: finish + apply in one launch behind the convolution).  Measured on the C2 scene:
    40 launches fewer, but every workgroup of the convolution then waits for its partial stores and a ticket before it
    retires -- the level-0 layers run 59 -> 71 us, which is what the shorter apply pass (25 -> 10 us) gives back:
    10.56 / 10.77 against 10.52 / 10.70 ms per step in alternating runs."""
    on = os.environ.get("WSIS_FUSE_BN_FIN", "0") != "0"
    if on:
        _n.require_experimental("WSIS_FUSE_BN_FIN (statistics finished inside the convolution)")
    return on


def _fuse_apply_level():
    """pyramid level from which the BatchNorm(+ReLU) in front of a convolution is applied by that convolution as it reads
    its input (WSIS_FUSE_BN_APPLY=<level>; default off).  Measured on the C2 scene: the 64 VALU + 13 LDS instructions per
    (offset, chunk) step are not hidden under the MFMAs -- the fused layers run 12-20 % longer (level 0: +11.5 us per
    launch against a 10 us apply pass, level 3: +4 us against 3 us) and the own-rows weight gradient 30 % longer, so the
    apply pass stays a launch of its own; the fused form remains for memory-tight runs (no activation copy)."""
    v = os.environ.get("WSIS_FUSE_BN_APPLY", "")
    lvl = int(v) if v.lstrip("-").isdigit() else 99
    if lvl < 99:
        _n.require_experimental("WSIS_FUSE_BN_APPLY (BatchNorm applied by the consuming convolution)")
    return lvl


def _fuse_stats_enabled():
    """BatchNorm statistics from the producing convolution's epilogue (WSIS_FUSE_BN_STATS=0: separate pass over x,
    bit-identical to the per-module walk)"""
    return os.environ.get("WSIS_FUSE_BN_STATS", "1") != "0"


def _lvl(level):
    """symbolic row count of a pyramid level (negative code, resolved per scene)"""
    return -(level + 1)


class _Recorder(object):
    """symbolic op list + allocation table of one arena"""

    def __init__(self, tag, align=256):
        self.tag, self.align = tag, align
        self.rows = []
        self.alloc_level, self.alloc_mult, self.alloc_slices = [], [], []

    def alloc(self, level, mult, per_slice=False):
        """``mult`` floats per row of pyramid level ``level`` (level < 0: ``mult`` floats in total); ``per_slice``:
        per 32-row slice of the level instead (BatchNorm partials of a convolution output)"""
        self.alloc_level.append(level)
        self.alloc_mult.append(int(mult))
        self.alloc_slices.append(bool(per_slice))
        return self.tag | (len(self.alloc_level) - 1)

    def op(self, kind, flags=0, M_in=0, M_out=0, K=0, Cin=0, Cout=0, eps=0.0, momentum=0.0, inp=(), out=()):
        """appends an op; rows are lists (later ops patch earlier ones: statistics targets, epilogue reductions):
        [kind, flags, M_in, M_out, K, Cin, Cout, eps, momentum, inp[N_IN], out[N_OUT], momentum2]"""
        inp = list(inp) + [0] * (N_IN - len(inp))
        out = list(out) + [0] * (N_OUT - len(out))
        self.rows.append([kind, flags, M_in, M_out, K, Cin, Cout, eps, momentum, inp, out, 0.0])
        return len(self.rows) - 1


class _Arena(object):
    """allocation table -> per-scene byte offsets"""

    def __init__(self, rec):
        self.level = np.asarray(rec.alloc_level, dtype=np.int64)
        self.mult = np.asarray(rec.alloc_mult, dtype=np.int64)
        self.slices = np.asarray(rec.alloc_slices, dtype=bool)
        self.align = rec.align

    def layout(self, Mvec):
        if len(self.level) == 0:
            return np.zeros(0, dtype=np.int64), 0
        rows = np.where(self.level >= 0, Mvec[np.maximum(self.level, 0)], 1)
        rows = np.where(self.slices, (rows + 31) // 32, rows)
        size = (rows * self.mult * 4 + self.align - 1) // self.align * self.align
        end = np.cumsum(size)
        return end - size, int(end[-1])


class _Template(object):
    """op records with every static field filled in; row counts and tagged pointers are patched per scene"""

    def __init__(self, rec):
        n = len(rec.rows)
        self.arr = np.zeros(n, dtype=OP_DTYPE)
        ptr = np.zeros((n, N_IN + N_OUT), dtype=np.uint64)          # inp | out
        m_in = np.zeros(n, dtype=np.int64)
        m_out = np.zeros(n, dtype=np.int64)
        for i, r in enumerate(rec.rows):
            self.arr[i] = (r[0], r[1], 0, 0, r[4], r[5], r[6], 0, r[7], r[8], r[11], 0, 0, 0)
            m_in[i], m_out[i] = r[2], r[3]
            ptr[i, :N_IN] = r[9]
            ptr[i, N_IN:] = r[10]
        self.m_in_lit, self.m_out_lit = np.maximum(m_in, 0), np.maximum(m_out, 0)
        self.m_in_lvl, self.m_out_lvl = np.maximum(-m_in - 1, 0), np.maximum(-m_out - 1, 0)
        self.m_in_sym, self.m_out_sym = m_in < 0, m_out < 0
        flat = ptr.reshape(-1)
        self.static = flat.copy()
        self.patch = {}
        for tag in _TAGS:
            pos = np.nonzero((flat & np.uint64(tag)) != 0)[0]
            off = ((flat[pos] >> np.uint64(_OFF_SHIFT)) & np.uint64(_OFF_MASK)) * np.uint64(4)      # bytes
            self.patch[tag] = (pos, (flat[pos] & np.uint64(_ID_MASK)).astype(np.int64), off)
            self.static[pos] = 0

    def instantiate(self, Mvec, luts):
        arr = self.arr.copy()
        arr["M_in"] = np.where(self.m_in_sym, Mvec[self.m_in_lvl], self.m_in_lit)
        arr["M_out"] = np.where(self.m_out_sym, Mvec[self.m_out_lvl], self.m_out_lit)
        flat = self.static.copy()
        for tag, (pos, idx, off) in self.patch.items():
            if len(pos):
                flat[pos] = luts[tag][idx] + off
        ptr = flat.reshape(-1, N_IN + N_OUT)
        arr["inp"] = ptr[:, :N_IN]
        arr["out"] = ptr[:, N_IN:]
        return arr


def _ptr(t):
    return 0 if t is None else t.data_ptr()


class _Table(object):
    """gather tables of one conv as slots: forward (rows = outputs) and dIn (rows = inputs) + flip flag"""

    def __init__(self, nbr_f=0, order_f=0, nbr_b=0, order_b=0, flip=0):
        self.nbr_f, self.order_f, self.nbr_b, self.order_b, self.flip = nbr_f, order_f, nbr_b, order_b, flip


def _subm(l):
    return _Table(_TBL | (l * 6), _TBL | (l * 6 + 1), _TBL | (l * 6), _TBL | (l * 6 + 1), 1)


def _down(l):
    return _Table(_TBL | (l * 6 + 2), _TBL | (l * 6 + 3), _TBL | (l * 6 + 4), _TBL | (l * 6 + 5), 0)


def _up(l):
    return _Table(_TBL | (l * 6 + 4), _TBL | (l * 6 + 5), _TBL | (l * 6 + 2), _TBL | (l * 6 + 3), 0)


class _Compiled(object):
    """forward + backward templates of one (model, mode)"""
    pass


class UNetProgram(object):
    """records forward and backward of input_conv + unet + output_layer of a ``Network``"""

    def __init__(self, net):
        self.net = net
        self.params = [p for m in (net.input_conv, net.unet, net.output_layer) for p in m.parameters()]
        self.bns = [m for mod in (net.unet, net.output_layer) for m in mod.modules() if isinstance(m, nn.BatchNorm1d)]
        self.flat_grad, self.flat_params = None, []
        self.flat_tail = None        # spare floats behind the gradients (parallel.GradSync packs the other
        self.tail_floats = 0         # parameters' gradients there: ONE collective per step)
        self.overlap = None          # wsis_parallel.GradSync: early all-reduce of the finished first part (see backward)
        self._garena = None          # _GradArena of the last backward program
        # True (default): ``.grad`` of the UNet's parameters are views of ONE buffer that the next backward pass
        # overwrites once the gradients were cleared -- a tensor the caller kept from the previous step changes under
        # it.  False (or WSIS_PERSISTENT_GRADS=0): every backward pass writes a buffer of its own (the aliasing is
        # gone; a 44 MB allocation + the cached views rebuilt per step, data-parallel exchange still in place).
        self.persistent_grads = os.environ.get("WSIS_PERSISTENT_GRADS", "1") != "0"
        self.bn_sync = None          # _BnSync while the model's BatchNorm layers share their statistics across ranks
        self._cache = {}

    # ---- symbolic recording: every helper returns (out_handle, backward_closure) -----------------------------
    def _conv(self, rec, x, conv, table, lvl_in, lvl_out, residual=0, stats=True):
        K = int(np.prod(conv.kernel_size))
        Cin, Cout = conv.in_channels, conv.out_channels
        W = conv.weight
        assert conv.bias is None, "the UNet convolutions carry no bias (sparse_unet3d.py)"
        y = rec.alloc(lvl_out, Cout)
        t = table if table is not None else _Table()
        # BatchNorm statistics of the output from the convolution's own epilogue (training passes, layers on the
        # wave-autonomous kernel): per-slice (sum, sum of squares) partials next to the output
        part = 0
        if stats and self._fuse_stats and sp_ops._use_fwd2(K, Cin, Cout):
            part = rec.alloc(lvl_out, 2 * Cout, per_slice=True)
        # the input may be a BatchNorm(+ReLU) that is applied while this convolution reads it (see _bn_relu)
        v = self._virt.get(x)
        flags, eps, bn_in = (F_STATS if part else 0), 0.0, (0, 0, 0, 0)
        if v is not None:
            assert sp_ops._use_fwd2(K, Cin, Cout) and v["C"] == Cin
            x = v["x"]
            flags |= F_BN_IN | (F_RELU if v["relu"] else 0)
            eps, bn_in = v["eps"], (v["mean"], v["var"], v["gamma"], v["beta"])
        row = rec.op(OP_CONV, flags, _lvl(lvl_in), _lvl(lvl_out), K, Cin, Cout, eps=eps,
                     inp=(x, t.nbr_f, t.order_f, W.data_ptr(), 0, residual) + bn_in, out=(y, part))
        if part:
            self._stats_src[y] = [(part, Cout, row)]
        self._acc_f.append(("spconv_fwd_kernel", t.nbr_f, lvl_out, Cin, Cout))

        def bwd(recb, dy, need_dx=True):
            dx = recb.alloc(lvl_in, Cin) if need_dx else 0
            dW = self._grad_handle(W)
            bflags = (F_FLIP if t.flip else 0) | ((F_BN_IN | (F_RELU if v["relu"] else 0)) if v is not None else 0)
            recb.op(OP_CONV_BWD, bflags, _lvl(lvl_in), _lvl(lvl_out), K, Cin, Cout, eps=eps,
                    inp=(x, W.data_ptr(), dy, t.nbr_f, t.order_f, t.nbr_b, t.order_b) + bn_in, out=(dx, dW))
            if need_dx:
                self._acc_b.append(("spconv_fwd_kernel", t.nbr_b, lvl_in, Cout, Cin))
            self._acc_b.append(("spconv_dw_kernel", t.nbr_f, lvl_out, Cin, Cout))
            return dx
        return y, bwd

    def _bn_relu(self, rec, x, bn, lvl, relu=True, fuse=False):
        """BatchNorm1d(+ReLU).  ``fuse``: the only consumer is a convolution on the wave-autonomous kernel, which applies
        the BatchNorm while it reads its input (the returned handle is virtual: nothing is written).  Training
        statistics come from the producing convolutions' epilogue partials where they exist -- finished inside those
        launches (F_STAT_FIN) or by a finalize op -- otherwise from a pass over x."""
        C = bn.num_features
        training = bn.training or not bn.track_running_stats
        update = bn.training and bn.track_running_stats
        flags = (F_RELU if relu else 0) | (F_TRAINING if training else 0) | (F_UPDATE if update else 0)
        if update and bn.num_batches_tracked is not None:
            assert bn.momentum is not None, "cumulative-average BatchNorm is not used by 3D-WSIS"
            self._count.append(bn)
        fuse = fuse and lvl >= self._fuse_apply_lvl and self._fuse_stats and C % 32 == 0
        momentum = bn.momentum if bn.momentum is not None else 0.1
        mean = rec.alloc(-1, C) if training else bn.running_mean.data_ptr()
        var = rec.alloc(-1, C) if training else bn.running_var.data_ptr()
        src = self._stats_src.get(x) if training else None
        have_parts = src is not None and sum(c for _, c, _ in src) == C and len(src) <= 2
        stats_done = False
        if have_parts and self._fuse_fin and lvl >= self._fuse_fin_lvl and all(rec.rows[r][10][4] == 0 for _, _, r in src):
            # every producer finishes its channel range of this BatchNorm's statistics inside its own launch
            c0 = 0
            rm = _ptr(bn.running_mean) if update else 0
            rv = _ptr(bn.running_var) if update else 0
            for _, c, r in src:
                row = rec.rows[r]
                tgt = (_at(mean, c0), _at(var, c0), _at(rm, c0), _at(rv, c0))
                if not (row[1] & F_STAT_FIN):
                    row[1] |= F_STAT_FIN
                    row[10][2], row[10][3], row[9][10], row[9][11] = tgt
                    row[8] = momentum
                else:
                    row[10][4:8] = tgt
                    row[11] = momentum
                c0 += c
            stats_done = True
        y = rec.alloc(lvl, 0 if fuse else C)
        if fuse:
            self._virt[y] = dict(x=x, C=C, mean=mean, var=var, gamma=_ptr(bn.weight), beta=_ptr(bn.bias), eps=bn.eps,
                                 relu=relu)
        if training and not stats_done:
            K0, parts, sflags = 0, (0, 0), flags
            if have_parts:
                sflags |= F_STATS            # the producers' epilogues wrote the partials: no statistics pass over x
                K0 = src[0][1]
                parts = (src[0][0], src[1][0] if len(src) == 2 else 0)
            rec.op(OP_BN_RELU, sflags, _lvl(lvl), _lvl(lvl), K0, C, C, bn.eps, momentum,
                   inp=(x, _ptr(bn.weight), _ptr(bn.bias), _ptr(bn.running_mean), _ptr(bn.running_var), parts[0], parts[1]),
                   out=(0 if fuse else y, mean, var))
            # algorithmic bytes of the op (bench.py's BatchNorm line): x read + y written; a statistics pass over x
            # (no epilogue partials) reads x once more
            self._acc_f.append(("bn_op", 0, lvl, C, (0 if fuse else 2) + (0 if have_parts else 1)))
        elif not fuse:
            # statistics known (finished by the producers, or evaluation mode): apply pass only
            rec.op(OP_BN_RELU, flags & ~(F_TRAINING | F_UPDATE), _lvl(lvl), _lvl(lvl), 0, C, C, bn.eps, momentum,
                   inp=(x, _ptr(bn.weight), _ptr(bn.bias), mean, var), out=(y,))
            self._acc_f.append(("bn_op", 0, lvl, C, 2))

        def bwd(recb, dy, addend=0):
            # ``addend``: gradient arriving at x over a second path (residual skip / UNet skip connection), added
            # in the same pass instead of a separate accumulation kernel
            dx = recb.alloc(lvl, C)
            dg = self._grad_handle(bn.weight) if bn.weight is not None else recb.alloc(-1, C)
            db = self._grad_handle(bn.bias) if bn.bias is not None else recb.alloc(-1, C)
            recb.op(OP_BN_RELU_BWD, flags & ~F_STATS, _lvl(lvl), _lvl(lvl), 0, C, C, bn.eps, 0.0,      # (F_STATS: _fuse_bn_bwd)
                    inp=(x, dy, mean, var, _ptr(bn.weight), _ptr(bn.bias), addend), out=(dx, dg, db))
            # x and dy read, dx written, the addend read where there is one (a separate reduction pass over x and dy,
            # taken when the producing dIn pass left no partials, is not counted: algorithmic = the fused form)
            self._acc_b.append(("bn_op", 0, lvl, C, 3 + (1 if addend else 0)))
            return dx
        return y, bwd

    def _fuse_bn_bwd(self, recb):
        """dIn passes whose output is the dy of a BatchNorm backward a few ops later (training mode, layer on the
        wave-autonomous kernel) also write that op's (sum dz, sum dz*xhat) slice partials from their epilogue: both
        ops get F_STATS and the BatchNorm op the partial buffer in inp[7] (WSIS_FUSE_BN_BWD=0: separate pass)"""
        if os.environ.get("WSIS_FUSE_BN_BWD", "1") == "0" or not self._fuse_stats:
            return
        rows = recb.rows
        for i, r in enumerate(rows):
            if r[0] != OP_CONV_BWD or not r[10][0]:
                continue
            K, Cin, Cout = r[4], r[5], r[6]
            if not sp_ops._use_fwd2(K, Cout, Cin):          # the dIn product gathers dY: roles of the channels swap
                continue
            dx = r[10][0]
            for j in range(i + 1, min(i + 5, len(rows))):
                b = rows[j]
                if b[0] == OP_BN_RELU_BWD and b[9][1] == dx and (b[1] & F_TRAINING) and b[5] == Cin and b[2] == r[2]:
                    lvl = -b[2] - 1
                    b[9][7] = recb.alloc(lvl, 2 * Cin, per_slice=True)
                    b[1] |= F_STATS
                    r[1] |= F_STATS
                    break

    def _residual_block(self, rec, x, blk, lvl):
        seq = blk.conv_branch
        bn1, conv1, bn2, conv2 = seq[0], seq[2], seq[3], seq[5]
        table = _subm(lvl)
        a1, b_bn1 = self._bn_relu(rec, x, bn1, lvl, fuse=True)
        z1, b_c1 = self._conv(rec, a1, conv1, table, lvl, lvl)
        a2, b_bn2 = self._bn_relu(rec, z1, bn2, lvl, fuse=True)
        first = blk.i_branch[0]
        if isinstance(first, nn.Identity):
            res, b_i = x, None
        else:
            res, b_i = self._conv(rec, x, first, None, lvl, lvl, stats=False)   # 1x1 projection of the skip path
        out, b_c2 = self._conv(rec, a2, conv2, table, lvl, lvl, residual=res)

        def bwd(recb, d_out):
            d_a2 = b_c2(recb, d_out)
            d_z1 = b_bn2(recb, d_a2)
            d_a1 = b_c1(recb, d_z1)
            d_skip = d_out if b_i is None else b_i(recb, d_out)
            return b_bn1(recb, d_a1, addend=d_skip)
        return out, bwd

    def _ublock(self, rec, x, ub, lvl):
        bwds = []
        for blk in ub.blocks:
            x, b = self._residual_block(rec, x, blk, lvl)
            bwds.append(b)
        if len(ub.nPlanes) == 1:
            def bwd_leaf(recb, d):
                for b in reversed(bwds):
                    d = b(recb, d)
                return d
            return x, bwd_leaf
        C0 = ub.nPlanes[0]
        identity = x
        a, b_bn = self._bn_relu(rec, x, ub.conv[0], lvl, fuse=True)
        d, b_down = self._conv(rec, a, ub.conv[2], _down(lvl), lvl, lvl + 1)
        u, b_u = self._ublock(rec, d, ub.u, lvl + 1)
        a2, b_bn2 = self._bn_relu(rec, u, ub.deconv[0], lvl + 1, fuse=True)
        up, b_up = self._conv(rec, a2, ub.deconv[2], _up(lvl), lvl + 1, lvl)
        cat = rec.alloc(lvl, 2 * C0)
        rec.op(OP_CAT, 0, _lvl(lvl), _lvl(lvl), 0, C0, C0, inp=(identity, up), out=(cat,))
        if identity in self._stats_src and up in self._stats_src:      # statistics of a concatenation = both halves'
            self._stats_src[cat] = self._stats_src[identity] + self._stats_src[up]
        x = cat
        tails = []
        for blk in ub.blocks_tail:
            x, b = self._residual_block(rec, x, blk, lvl)
            tails.append(b)

        def bwd(recb, dcur):
            for b in reversed(tails):
                dcur = b(recb, dcur)
            d_id = recb.alloc(lvl, C0)
            d_up = recb.alloc(lvl, C0)
            recb.op(OP_SPLIT, 0, _lvl(lvl), _lvl(lvl), 0, C0, C0, inp=(dcur,), out=(d_id, d_up))
            d_a2 = b_up(recb, d_up)
            d_u = b_bn2(recb, d_a2)
            d_d = b_u(recb, d_u)
            d_a = b_down(recb, d_d)
            d_x = b_bn(recb, d_a, addend=d_id)
            for b in reversed(bwds):
                d_x = b(recb, d_x)
            return d_x
        return x, bwd

    def _grad_handle(self, p):
        # parameter gradients live densely in their own flat buffer (one all-reduce for data parallelism)
        h = self._grad.get(id(p))
        if h is None:
            h = self._prec.alloc(-1, p.numel())
            self._grad[id(p)] = h
        return h

    # ---- compile (once per mode) / bind (per scene) ------------------------------------------------------------
    def _mode_key(self, need_dx):
        # bn_sync: the synced pass intercepts training-mode BatchNorm ops only -- the forms that finish the statistics
        # inside the producing convolution or apply them inside the consuming one leave no such op, so a program
        # compiled with them must never serve a synced pass (and the other way round)
        return (need_dx, self.bn_sync is not None, _fuse_stats_enabled(), _fuse_fin_enabled(), _fuse_apply_level(),
                os.environ.get("WSIS_FUSE_BN_FIN_LVL", "0"),
                os.environ.get("WSIS_FWD2", "1"), tuple(bn.training for bn in self.bns), tuple(p.data_ptr() for p in self.params),
                tuple(bn.running_mean.data_ptr() if bn.running_mean is not None else 0 for bn in self.bns))

    def compiled(self, need_dx):
        key = self._mode_key(need_dx)
        c = self._cache.get(key)
        if c is not None:
            return c
        if len(self._cache) > 8:
            self._cache.clear()
        net = self.net
        c = _Compiled()
        rec, recb = _Recorder(_FWD), _Recorder(_BWD)
        self._prec, self._grad = _Recorder(_PAR, align=16), {}
        self._acc_f, self._acc_b, self._count = [], [], []
        self._stats_src, self._fuse_stats = {}, _fuse_stats_enabled()
        self._virt, self._fuse_apply_lvl = {}, _fuse_apply_level()
        self._fuse_fin = _fuse_fin_enabled() and self._fuse_stats
        self._fuse_fin_lvl = int(os.environ.get("WSIS_FUSE_BN_FIN_LVL", "0"))      # (EXPERIMENTAL build: from this level on)
        if self.bn_sync is not None:      # every BatchNorm layer stays an op of its own: _run_synced runs it between parts
            self._fuse_fin, self._fuse_apply_lvl = False, 99
        y, b_in = self._conv(rec, _EXT | 0, net.input_conv[0], _subm(0), 0, 0)
        y, b_u = self._ublock(rec, y, net.unet, 0)
        out, b_out = self._bn_relu(rec, y, net.output_layer[0], 0)
        d = b_out(recb, _EXT | 1)
        d = b_u(recb, d)
        dx = b_in(recb, d, need_dx)
        self._fuse_bn_bwd(recb)
        c.fwd, c.bwd = _Template(rec), _Template(recb)
        c.fwd_arena, c.bwd_arena, c.par_arena = _Arena(rec), _Arena(recb), _Arena(self._prec)
        c.out_id, c.dx_id = out & _ID_MASK, (dx & _ID_MASK) if need_dx else -1
        c.grad_ids = [(self._grad[id(p)] & _ID_MASK) if id(p) in self._grad else -1 for p in self.params]
        c.acc_f, c.acc_b, c.count = self._acc_f, self._acc_b, self._count
        # milestone of the backward list: the op after which the first ~half of the flat parameter-gradient buffer is
        # final (gradients are allocated in the order the backward pass produces them)
        done = np.zeros(len(recb.rows), dtype=np.int64)
        hw = 0
        for i, r in enumerate(recb.rows):
            for h in r[10]:
                if h & _PAR:
                    hw = max(hw, (h & _ID_MASK) + 1)
            done[i] = hw
        c.bwd_grads_done = done
        c.out_channels = net.output_layer[0].num_features
        self._cache[key] = c
        return c

    def bind(self, tensor):
        """rulebooks of the pyramid of ``tensor`` (built if missing) -> row counts and the table-pointer LUT"""
        net = self.net
        sp_ops.prebuild_unet_rulebooks(tensor, net.blocks)
        L = net.blocks
        self.Mvec = np.zeros(L, dtype=np.int64)
        self.table_lut = np.zeros(L * 6, dtype=np.uint64)
        self.table_tensors = [None] * (L * 6)
        pyr = getattr(tensor.indice_dict, "pyramid", None)
        if pyr is not None and pyr.n_levels == L:
            # the pyramid came from one native call into one arena: row counts and table pointers straight from its
            # layout, no per-table tensor views (they are made only if the profiler asks for pair counts)
            lay, base = pyr.layout, np.uint64(pyr.base)
            self.Mvec[:] = lay[:, sp_ops.PYR_ROWS]
            fields = (sp_ops.PYR_SUBM_NBR_P, sp_ops.PYR_SUBM_ORDER, sp_ops.PYR_DOWN_NBR_P, sp_ops.PYR_DOWN_ORDER,
                      sp_ops.PYR_UP_NBR_P, sp_ops.PYR_UP_ORDER)
            off = lay[:, fields]                                   # [L, 6]
            self.table_lut[:] = np.where(off >= 0, off.astype(np.uint64) + base, np.uint64(0)).reshape(-1)
            self.table_tensors = _LazyTables(tensor.indice_dict, L)
            self.keep = [tensor.indice_dict]
            return
        for lvl in range(L):
            rb = tensor.indice_dict["subm%d" % (lvl + 1)]
            self.Mvec[lvl] = rb.in_indices.shape[0]
            ts = [rb.nbr_p, rb.order, None, None, None, None]
            if lvl + 1 < L:
                rd = tensor.indice_dict["spconv%d" % (lvl + 1)]
                ts[2:] = [rd.nbr_p, rd.order, rd.nbr_up_p, rd.order_up]
            for j, t in enumerate(ts):
                self.table_lut[lvl * 6 + j] = _ptr(t)
                self.table_tensors[lvl * 6 + j] = t
        self.keep = [tensor.indice_dict]     # the tables must outlive the backward pass

    def account(self, entries, Mvec, tensors):
        prof = sp_ops.PROFILER
        if prof is None:
            return
        for name, nbr, lvl, Cin, Cout in entries:
            if name == "bn_op":      # (Cin = channels, Cout = tensors of [rows, channels] the op reads + writes)
                M = int(Mvec[lvl])
                prof.end(name, None, M * Cin * Cout * 4, 0, (M, Cin, Cout, M))
                continue
            t = tensors[nbr & _ID_MASK] if nbr else None
            P = prof.pairs(t, int(Mvec[lvl]))
            prof.end(name, None, P * (Cin + Cout) * 4 + P * 8, 2 * P * Cin * Cout, (int(Mvec[lvl]), Cin, Cout, P))


class _LazyTables(object):
    """table_tensors[level*6 + slot] made on demand from the pyramid's rulebooks (the profiler's pair counts)"""

    def __init__(self, indice_dict, L):
        self.d, self.L = indice_dict, L

    def __getitem__(self, i):
        lvl, slot = divmod(int(i), 6)
        if slot < 2:
            rb = self.d["subm%d" % (lvl + 1)]
            return (rb.nbr_p, rb.order)[slot]
        rd = self.d["spconv%d" % (lvl + 1)]
        return (rd.nbr_p, rd.order, rd.nbr_up_p, rd.order_up)[slot - 2]


def _arena_tensor(nbytes, device, zero=False):
    make = torch.zeros if zero else torch.empty
    t = make(nbytes + 256, dtype=torch.uint8, device=device)
    return t, (t.data_ptr() + 255) // 256 * 256


def _view(arena, base, offset, shape):
    numel = int(np.prod(shape))
    off = base + int(offset) - arena.data_ptr()
    return arena[off:off + numel * 4].view(torch.float32).view(shape)


class _GradArena(object):
    """the flat parameter-gradient buffer of a compiled backward program, with the per-parameter views of it.  Kept by
    the program from one backward pass to the next (the executor overwrites every gradient in it; the alignment gaps and
    the tail were zeroed once): a pass costs the host no allocation, no 230 slice / view calls."""

    @staticmethod
    def key_of(prog, c, ptotal, tail, dev):
        return (id(c), int(ptotal), int(tail), dev, tuple(p.requires_grad for p in prog.params))

    def __init__(self, prog, c, poffs, ptotal, tail, dev):
        self.key = self.key_of(prog, c, ptotal, tail, dev)
        self.c = c                                   # (keeps id(c) from being reused while this buffer lives)
        self.arena, self.base = _arena_tensor(ptotal + tail, dev, zero=True)   # alignment gaps are all-reduced too
        self.flat = self.arena.view(torch.float32)
        self.first = (self.base - self.arena.data_ptr()) // 4
        self.lut = poffs.astype(np.uint64) + np.uint64(self.base)
        self.params, self.views = [], []
        for i, p in enumerate(prog.params):
            gid = c.grad_ids[i]
            if gid >= 0 and p.requires_grad:
                off = self.first + int(poffs[gid]) // 4
                self.params.append(p)
                self.views.append(self.flat[off:off + p.numel()].view(p.shape))
        t0 = self.first + (ptotal + 3) // 4
        self.tail = self.flat[t0:t0 + prog.tail_floats] if prog.tail_floats else None


class UNetFunction(Function):
    """features [M0, Cin] -> [M0, m]; ``prog`` is a bound UNetProgram.  Its parameters are NOT inputs of the node:
    ``backward`` stores their gradients (views of the program's flat buffer) in ``.grad`` itself -- what 230
    AccumulateGrad nodes did, ~1 ms of host time per step.  ``anchor`` (any parameter that requires a gradient) only
    makes autograd run the node when the input features need no gradient."""

    @staticmethod
    def forward(ctx, x, prog, anchor=None):
        _n.require_cuda(x)
        x = x if (x.dtype == torch.float32 and x.is_contiguous()) else x.contiguous().float()
        need_dx = bool(x.requires_grad)
        c = prog.compiled(need_dx)
        for bn in c.count:
            wsis_ops._defer_batch_count(bn)
        Mvec, table_lut, tensors = prog.Mvec, prog.table_lut, prog.table_tensors
        offs, total = c.fwd_arena.layout(Mvec)
        arena, base = _arena_tensor(total, x.device)
        fwd_lut = offs.astype(np.uint64) + np.uint64(base)
        none = fwd_lut[:0]
        luts = {_FWD: fwd_lut, _TBL: table_lut, _EXT: np.array([x.data_ptr(), 0], dtype=np.uint64), _BWD: none,
                _PAR: none}
        sy = prog.bn_sync.new_pass() if prog.bn_sync is not None else None      # row counts / arenas belong to THIS pass
        if sy is not None:
            sy.arenas = [arena]
            _run_synced(_n.hip(), c.fwd.instantiate(Mvec, luts), x.device, sy)
        else:
            _run(_n.hip(), c.fwd.instantiate(Mvec, luts), x.device)
        ctx.sy = sy
        # tables sized from the batch's host-side level counts: the device's own counts are compared inside the SAME
        # pass -- an inference pass right here (its result is used next; the read waits for the rulebook chain on the
        # side stream only), a training pass at the end of its backward pass, i.e. before the optimizer can use a
        # gradient (reading here would park the issuing thread for the length of the rulebook chain, ~1 ms per step)
        if not (torch.is_grad_enabled() and (need_dx or anchor is not None)):
            sp_ops.verify_pending_counts()
        prog.account(c.acc_f, Mvec, tensors)
        out = _view(arena, base, offs[c.out_id], (int(Mvec[0]), c.out_channels))
        ctx.prog, ctx.c, ctx.arena, ctx.fwd_lut, ctx.x = prog, c, arena, fwd_lut, x
        ctx.Mvec, ctx.table_lut, ctx.tensors, ctx.keep = Mvec, table_lut, tensors, prog.keep
        return out

    @staticmethod
    def backward(ctx, d_out):
        prog, c, Mvec = ctx.prog, ctx.c, ctx.Mvec
        d_out = d_out if (d_out.dtype == torch.float32 and d_out.is_contiguous()) else d_out.contiguous().float()
        dev = d_out.device
        boffs, btotal = c.bwd_arena.layout(Mvec)
        poffs, ptotal = c.par_arena.layout(Mvec)
        garena, gbase = _arena_tensor(btotal, dev)
        tail = int(prog.tail_floats) * 4
        # the program's own gradient buffer while no parameter still holds a gradient (zero_grad(set_to_none=True), the
        # reference's loop); otherwise (gradient accumulation, a second backward pass) a buffer of this pass's own
        ga = prog._garena
        if ga is None or ga.key != _GradArena.key_of(prog, c, ptotal, tail, dev):
            ga = prog._garena = _GradArena(prog, c, poffs, ptotal, tail, dev)
        if not prog.persistent_grads or any(p.grad is not None for p in ga.params):
            ga = _GradArena(prog, c, poffs, ptotal, tail, dev)
        parena, pflat, first = ga.arena, ga.flat, ga.first
        luts = {_FWD: ctx.fwd_lut, _BWD: boffs.astype(np.uint64) + np.uint64(gbase), _PAR: ga.lut, _TBL: ctx.table_lut,
                _EXT: np.array([ctx.x.data_ptr(), d_out.data_ptr()], dtype=np.uint64)}
        hook = prog.overlap
        if ctx.sy is not None:
            ctx.sy.arenas = [ctx.arena, garena, parena]
            _run_synced(_n.hip(), c.bwd.instantiate(Mvec, luts), dev, ctx.sy)
        elif hook is not None and hook.ready() and len(poffs) > 1:
            # all-reduce of the first part of the gradient buffer starts while the rest of the pass still runs
            starts = np.append(poffs, ptotal)
            want = ptotal // 2
            op_i = int(np.searchsorted(starts[c.bwd_grads_done], want))          # first op with >= half final
            op_i = min(op_i, len(c.bwd_grads_done) - 2)
            split = int(starts[c.bwd_grads_done[op_i]]) // 4
            _run(_n.hip(), c.bwd.instantiate(Mvec, luts), dev, mark_op=op_i, waiter=hook.comm_stream_ptr())
            hook.early(pflat[first:first + split], first + split)
        else:
            _run(_n.hip(), c.bwd.instantiate(Mvec, luts), dev)
        prog.account(c.acc_b, Mvec, ctx.tensors)
        for p, v in zip(ga.params, ga.views):
            p.grad = v if p.grad is None else p.grad + v
        # data-parallel hook: the gradients of ``flat_params`` are views of ``flat_grad`` (parallel.GradSync)
        prog.flat_grad, prog.flat_params, prog.flat_tail = pflat, ga.params, ga.tail
        dx = _view(garena, gbase, boffs[c.dx_id], tuple(ctx.x.shape)) if c.dx_id >= 0 else None
        sp_ops.verify_pending_counts()       # (see forward: the counts of this pass's rulebooks, long written by now)
        return dx, None, None



class _BnSync(object):
    """SyncBatchNorm INSIDE the executor's pass (train_scannetv2.py:734-736 converts every BatchNorm when num_gpus > 1):
    the op list is issued in parts (``wsis_run_ops_part``) that end in front of every training-mode BatchNorm op; the
    layer itself runs here -- local statistics from the producers' epilogue partials, ONE all-gather of (mean, var,
    count) combined in fp64 (Chan), apply; backward: local (sum dz, sum dz xhat), ONE all-reduce, dx with the global sums
    -- and the next part follows.  Convolutions, concatenations and the weight-gradient side stream stay in the native
    executor; every rank issues the same collectives in the same order (same op list), zero-row ranks included."""

    def __init__(self, prog, group):
        self.group = group
        self.by_ptr = {}
        for bn in prog.bns:
            for t in (bn.running_mean, bn.running_var):
                if t is not None:
                    self.by_ptr[t.data_ptr()] = t
        self.counts = {}          # mean pointer of a layer -> global row count (fp64 device scalar), forward -> backward
        self.arenas = []

    def new_pass(self):
        """the state of one forward + backward pass: two forward passes before a backward (gradient accumulation, an
        extra training-mode forward) must not wipe each other's global row counts"""
        p = _BnSync.__new__(_BnSync)
        p.group, p.by_ptr, p.counts, p.arenas = self.group, self.by_ptr, {}, []
        return p

    def view(self, ptr, numel):
        ptr = int(ptr)
        for a in self.arenas:
            base = a.data_ptr()
            if base <= ptr < base + a.numel():
                return a[ptr - base:ptr - base + 4 * numel].view(torch.float32)
        raise _n.WsisError("synced BatchNorm: pointer outside the pass's arenas")


def _bn_fwd_synced(lib, op, sy, dev, st, slot):
    import torch.distributed as dist
    M, C, flags = int(op["M_in"]), int(op["Cin"]), int(op["flags"])
    inp, out = [int(v) for v in op["inp"]], [int(v) for v in op["out"]]
    mom, eps = float(op["momentum"]), float(op["eps"])
    mean, var = sy.view(out[1], C), sy.view(out[2], C)
    if M > 0 and (flags & F_STATS):
        n_part = (M + 31) // 32
        C0 = int(op["K"]) if inp[6] else C
        wsb = lib.wsis_bn_stats_finalize_workspace_bytes(n_part, C)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        _n.check(lib.wsis_bn_stats_finalize(inp[5], n_part, M, C0, out[1], out[2], None, None, mom, _n.ptr(ws), wsb, slot, st),
                 "bn_stats_finalize")
        if inp[6]:
            _n.check(lib.wsis_bn_stats_finalize(inp[6], n_part, M, C - C0, out[1] + 4 * C0, out[2] + 4 * C0, None, None, mom,
                                                _n.ptr(ws), wsb, slot, st), "bn_stats_finalize")
    elif M > 0:
        wsb = lib.wsis_bn_workspace_bytes(M, C)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        _n.check(lib.wsis_bn_stats(inp[0], M, C, out[1], out[2], None, None, mom, _n.ptr(ws), wsb, st), "bn_stats")
    else:
        mean.zero_()
        var.zero_()
    t = torch.cat((mean.double(), var.double(), torch.full((1,), float(M), dtype=torch.float64, device=dev)))
    parts = [torch.empty_like(t) for _ in range(dist.get_world_size(sy.group))]
    dist.all_gather(parts, t, group=sy.group)
    g = torch.stack(parts)
    n_i = g[:, 2 * C:2 * C + 1]
    N = n_i.sum()
    mean64 = (g[:, :C] * n_i).sum(0) / N
    dm = g[:, :C] - mean64
    var64 = ((g[:, C:2 * C] + dm * dm) * n_i).sum(0) / N
    mean.copy_(mean64.float())
    var.copy_(var64.float())
    if flags & F_UPDATE:
        rm, rv = sy.by_ptr[inp[3]], sy.by_ptr[inp[4]]
        unb = var64 * (N / (N - 1.0).clamp_min(1.0))
        rm.mul_(1.0 - mom).add_(mean, alpha=mom)
        rv.mul_(1.0 - mom).add_(unb.float(), alpha=mom)
    sy.counts[out[1]] = N
    if out[0] and M > 0:
        _n.check(lib.wsis_bn_apply(inp[0], out[1], out[2], inp[1] or None, inp[2] or None, eps, 1 if flags & F_RELU else 0,
                                   out[0], M, C, st), "bn_apply")


def _bn_bwd_synced(lib, op, sy, dev, st):
    import torch.distributed as dist
    M, C, flags = int(op["M_in"]), int(op["Cin"]), int(op["flags"])
    inp, out = [int(v) for v in op["inp"]], [int(v) for v in op["out"]]
    eps, relu = float(op["eps"]), 1 if flags & F_RELU else 0
    dg, db = sy.view(out[1], C), sy.view(out[2], C)
    if M > 0:
        wsb = lib.wsis_bn_workspace_bytes(M, C)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        _n.check(lib.wsis_bn_bwd(inp[0], inp[1], inp[2], inp[3], inp[4] or None, inp[5] or None, eps, relu, 1, None, out[1],
                                 out[2], None, M, C, _n.ptr(ws), wsb, st), "bn_bwd")
    else:
        dg.zero_()
        db.zero_()
    g = torch.cat((dg, db)).double()
    dist.all_reduce(g, group=sy.group)
    N = sy.counts[inp[2]]
    if M > 0:
        sc = (g * (float(M) / N)).float()      # the kernel divides by the local row count
        _n.check(lib.wsis_bn_bwd_apply(inp[0], inp[1], inp[2], inp[3], inp[4] or None, inp[5] or None, _n.ptr(sc[:C]),
                                       _n.ptr(sc[C:]), eps, relu, out[0], inp[6] or None, M, C, st), "bn_bwd_apply")


def _run_synced(lib, ops, device, sy):
    n = len(ops)
    base, item = ops.ctypes.data, ops.dtype.itemsize
    kinds = ops["kind"]
    # backward: the dIn epilogue's partials pair a CONV_BWD op with the BatchNorm op behind it INSIDE one list; the parts
    # end in front of that op and the synced layer makes its own reduction pass, so the pairing is switched off
    bwd_pair = (kinds == OP_CONV_BWD) | (kinds == OP_BN_RELU_BWD)
    ops["flags"] = np.where(bwd_pair, ops["flags"] & ~F_STATS, ops["flags"])
    flags = ops["flags"]
    is_bn = ((kinds == OP_BN_RELU) | (kinds == OP_BN_RELU_BWD)) & ((flags & F_TRAINING) != 0)
    sync = _n.ptr(_n.sync_block(device))
    st = _n.stream_ptr()
    keep = []                     # every part's workspace stays alive until the weight-gradient side stream is joined
    # whatever happens between the parts (a collective that times out, a failed check), the joining call must be issued:
    # it clears the side stream's pending join and orders the queued weight-gradient launches before `keep` is freed
    try:
        i = 0
        while i < n:
            j = i
            while j < n and not is_bn[j]:
                j += 1
            if j > i:
                p = base + i * item
                wsb = lib.wsis_run_ops_workspace_bytes(p, j - i)
                if wsb < 0:
                    raise _n.WsisError("run_ops workspace query failed")
                ws = torch.empty(wsb, dtype=torch.uint8, device=device)
                keep.append(ws)
                _n.check(lib.wsis_run_ops_part(p, j - i, _n.ptr(ws), wsb, sync, st, 0), "run_ops_part")
            if j < n:
                if int(kinds[j]) == OP_BN_RELU:
                    _bn_fwd_synced(lib, ops[j], sy, device, st, sync)
                else:
                    _bn_bwd_synced(lib, ops[j], sy, device, st)
                j += 1
            i = j
    finally:
        _n.check(lib.wsis_run_ops_part(None, 0, None, 0, sync, st, 1), "run_ops_part")
    return keep


def _run(lib, ops, device, mark_op=-1, waiter=None):
    n = len(ops)
    p = ops.ctypes.data
    ws_bytes = lib.wsis_run_ops_workspace_bytes(p, n)
    if ws_bytes < 0:
        raise _n.WsisError("run_ops workspace query failed")
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device)
    sync = _n.ptr(_n.sync_block(device))
    if mark_op >= 0:
        _n.check(lib.wsis_run_ops_marked(p, n, _n.ptr(ws), ws_bytes, sync, _n.stream_ptr(), int(mark_op), waiter),
                 "run_ops")
    else:
        _n.check(lib.wsis_run_ops(p, n, _n.ptr(ws), ws_bytes, sync, _n.stream_ptr()), "run_ops")


_EXPERIMENTAL_SWITCHES = ("WSIS_DEEP", "WSIS_FWD2P", "WSIS_RING", "WSIS_GRAPH", "WSIS_BN_FUSED_APPLY", "WSIS_BN_FUSED_FWD", "WSIS_BN_FUSED_BWD")


# launch-plan / tuning knobs the C layer reads through tune_env() (csrc/common.h): live in the EXPERIMENTAL build, compiled
# to their measured defaults in the default library (tests/test_abi.py keeps this list equal to the sources)
TUNE_KNOBS = (
    "WSIS_BN_APPLY_PT", "WSIS_BN_FUSED_GRID", "WSIS_BN_TICKET", "WSIS_DW2", "WSIS_DW2_HI", "WSIS_DW2_LO",
    "WSIS_DW2_WAVES", "WSIS_DW2_XCD", "WSIS_DW2_XSH", "WSIS_DW2_DENSE_WG", "WSIS_DW2_RED", "WSIS_DW2_PFLOOR", "WSIS_DW_BAL", "WSIS_DW_H16", "WSIS_DW_DIV", "WSIS_DW_THREAD", "WSIS_FWD2P_MIN",
    "WSIS_FWD2_BD", "WSIS_FWD2_DA_NW4", "WSIS_FWD2_DEAL", "WSIS_FWD2_NOSLAB", "WSIS_FWD2_NW", "WSIS_FWD2_NW_MAX",
    "WSIS_FWD2_NW_MAX_NOSLAB", "WSIS_FWD2_RG", "WSIS_FWD2_RGH", "WSIS_FWD2_RGH_LATE", "WSIS_FWD2_SNAKE",
    "WSIS_FWD2_TR", "WSIS_FWD2_TR_DA", "WSIS_FWD2_WAVES", "WSIS_FWD2_XCD", "WSIS_FWD2_ZS", "WSIS_FWD_NB_SMALL",
    "WSIS_FWD_SMALL_TILES", "WSIS_KSPLIT_TARGET", "WSIS_SL_BLOCKS", "WSIS_TILE_BAND", "WSIS_TILE_SCHED",
    "WSIS_TILE_SCHED_MIN", "WSIS_XCD_AWARE")
_WARNED_KNOBS = set()
_KNOB_SCAN = [0]


def _check_experimental_switches():
    """the switches of the retired designs exist in the EXPERIMENTAL build only: asking for one on
    the default library is an error, not a silent no-op; a tuning knob that is set while the default library is loaded
    (where it compiles to its default) gets ONE warning -- an A/B tool run on the wrong flavour must not report A == B
    silently"""
    for k in _EXPERIMENTAL_SWITCHES:
        if os.environ.get(k, "0") not in ("", "0"):
            _n.require_experimental(k + "=" + os.environ[k])
    _KNOB_SCAN[0] += 1
    if _KNOB_SCAN[0] > 4 and _KNOB_SCAN[0] % 64:       # (the scan is ~10 us of host time: the first passes, then 1 in 64)
        return
    ignored = [k for k in TUNE_KNOBS if k in os.environ and k not in _WARNED_KNOBS]
    if ignored and not _n.experimental():
        import warnings
        _WARNED_KNOBS.update(ignored)
        warnings.warn("tuning knob(s) %s are set but the DEFAULT libwsis_hip.so is loaded: they are live in the "
                      "EXPERIMENTAL build only (make -C 3d-wsis_amd/csrc EXPERIMENTAL=1) and have no effect here"
                      % ", ".join(ignored), RuntimeWarning, stacklevel=2)


def run_unet(net, input_tensor, sync_group=None):
    """input_conv + unet + output_layer of ``net`` on ``input_tensor`` (SparseConvTensor) -> features [M0, m].
    ``sync_group``: the process group the BatchNorm layers take their batch statistics over (None: per rank)"""
    _check_experimental_switches()
    prog = getattr(net, "_native_prog", None)
    if prog is None:
        prog = UNetProgram(net)
        net._native_prog = prog
    if sync_group is None:
        prog.bn_sync = None
    elif prog.bn_sync is None or prog.bn_sync.group is not sync_group:
        prog.bn_sync = _BnSync(prog, sync_group)
    prog.bind(input_tensor)
    anchor = next((p for p in prog.params if p.requires_grad), None) if torch.is_grad_enabled() else None
    return UNetFunction.apply(input_tensor.features, prog, anchor)
