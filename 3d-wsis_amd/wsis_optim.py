"""AdamW step of the training iteration (``train_scannetv2.py:251``, optimizer of
``config/ScanNet_v2_3D_WSIS.yaml:58-61``) as ONE launch of ``wsis_adamw_step`` over all parameters.

Same update rule as ``torch.optim.AdamW`` (tested against it step by step); the moments live in two flat buffers, a
small (pointer, size) table per step tells the kernel where each parameter, gradient and moment slice is.  The table
goes through a ring of pinned host buffers: the copy of step t is still queued on the stream while the host already
prepares step t+1 (gradient tensors of most parameters are new objects every backward pass)."""
import numpy as np
import torch

import wsis_native as _n

_RING = 4


class FlatAdamW(torch.optim.Optimizer):
    """A ``torch.optim.Optimizer`` (one parameter group): ``param_groups[0]`` is live -- the step reads ``lr``,
    ``betas``, ``eps`` and ``weight_decay`` from it, so the reference's ``PolyLR`` (an ``_LRScheduler``, stepped per
    epoch, ``utils/lr_scheduler.py``) and its ``save_checkpoint`` isinstance check work on it unchanged."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        params = list(params)
        if params and isinstance(params[0], dict):
            if len(params) != 1:
                raise ValueError("FlatAdamW takes ONE parameter group (one set of hyper-parameters per launch)")
            params = list(params[0]["params"])
        params = [p for p in params if p.requires_grad]
        if not params:
            raise ValueError("no parameters to optimise")
        super().__init__(params, dict(lr=float(lr), betas=(float(betas[0]), float(betas[1])), eps=float(eps),
                                      weight_decay=float(weight_decay)))
        self.params = self.param_groups[0]["params"]
        dev = self.params[0].device
        if dev.type != "cuda":
            raise _n.WsisError("FlatAdamW runs on the MI355X only (there is no CPU fallback)")
        for p in self.params:
            if p.dtype != torch.float32 or not p.is_contiguous() or p.device != dev:
                raise ValueError("FlatAdamW expects contiguous fp32 parameters on one device")
        lib = _n.hip()
        self.steps = np.zeros(len(self.params), dtype=np.int64)    # updates per parameter (torch counts per parameter)
        numel = [p.numel() for p in self.params]
        # moments: one flat buffer each; slices start at multiples of 4 floats so the float4 path applies
        starts, total = [], 0
        for n in numel:
            starts.append(total)
            total += (n + 3) // 4 * 4
        self.exp_avg = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(total, dtype=torch.float32, device=dev)
        seg_bytes = lib.wsis_adamw_segment_bytes()
        assert seg_bytes == 56
        chunk = lib.wsis_adamw_chunk()
        n = len(self.params)
        self._table = np.zeros((n, 7), dtype=np.int64)            # p, g, m, v, numel, (step_size, inv_sqrt_bc2), (g_clamp, 0) as 2 x fp32
        self._table[:, 0] = [p.data_ptr() for p in self.params]
        self._table[:, 2] = [self.exp_avg.data_ptr() + 4 * s for s in starts]
        self._table[:, 3] = [self.exp_avg_sq.data_ptr() + 4 * s for s in starts]
        self._numel = np.asarray(numel, dtype=np.int64)
        blocks = np.concatenate([np.stack([np.full((k + chunk - 1) // chunk, i, dtype=np.int32),
                                           np.arange((k + chunk - 1) // chunk, dtype=np.int32)], 1)
                                 for i, k in enumerate(numel)])
        self._blocks = torch.from_numpy(np.ascontiguousarray(blocks)).to(dev)
        self._n_blocks = int(blocks.shape[0])
        self._host = [torch.empty((n, 7), dtype=torch.int64).pin_memory() for _ in range(_RING)]
        self._dev = [torch.empty((n, 7), dtype=torch.int64, device=dev) for _ in range(_RING)]
        self._done = [None] * _RING
        self._slot = 0

    # hyper-parameters live in param_groups[0] (schedulers write there); the attributes are views of it
    lr = property(lambda self: float(self.param_groups[0]["lr"]),
                  lambda self, v: self.param_groups[0].__setitem__("lr", float(v)))
    betas = property(lambda self: tuple(float(b) for b in self.param_groups[0]["betas"]))
    eps = property(lambda self: float(self.param_groups[0]["eps"]))
    weight_decay = property(lambda self: float(self.param_groups[0]["weight_decay"]))

    def set_grad_clamp(self, params, bound):
        """gradients of ``params`` are clamped to [-bound, bound] inside the step's ONE launch (and written back, like the
        reference's ``p.grad.data.clamp_(-1, 1)`` over the ECC parameters, train_scannetv2.py:247-249); bound <= 0: off"""
        ids = {id(p) for p in params}
        pair = np.array([max(float(bound), 0.0), 0.0], dtype=np.float32).view(np.int64)[0]
        n = 0
        for i, p in enumerate(self.params):
            if id(p) in ids:
                self._table[i, 6] = pair
                n += 1
        return n

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if len(self.param_groups) != 1:
            raise ValueError("FlatAdamW takes ONE parameter group")
        tab = self._table
        grads = [p.grad for p in self.params]
        for i, g in enumerate(grads):                 # slow path only for a gradient the kernel cannot read as is
            if g is not None and (g.dtype != torch.float32 or not g.is_contiguous()):
                grads[i] = self.params[i].grad = g.contiguous().float()
        g_ptr = np.fromiter((0 if g is None else g.data_ptr() for g in grads), dtype=np.int64, count=len(grads))
        tab[:, 0] = np.fromiter((p.data_ptr() for p in self.params), dtype=np.int64, count=len(grads))
        tab[:, 1] = g_ptr
        live = g_ptr != 0
        tab[:, 4] = np.where(live, self._numel, 0)
        self.steps += live
        t = np.maximum(self.steps, 1).astype(np.float64)
        corr = np.stack([self.lr / (1.0 - self.betas[0] ** t), 1.0 / np.sqrt(1.0 - self.betas[1] ** t)], 1)
        tab[:, 5] = np.ascontiguousarray(corr.astype(np.float32)).view(np.int64).reshape(-1)
        k = self._slot
        self._slot = (k + 1) % _RING
        if self._done[k] is not None:
            self._done[k].synchronize()               # the copy issued _RING steps ago has long finished
        self._host[k].numpy()[:] = tab
        self._dev[k].copy_(self._host[k], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._done[k] = ev
        _n.check(_n.hip().wsis_adamw_step(_n.ptr(self._dev[k]), _n.ptr(self._blocks), self._n_blocks, self.lr,
                                          self.betas[0], self.betas[1], self.eps, self.weight_decay,
                                          _n.stream_ptr()), "adamw_step")
        return loss

    # state in the layout of torch.optim.AdamW.state_dict()["state"] (checkpoint interchange)
    def state_dict(self):
        state, off = {}, 0
        for i, p in enumerate(self.params):
            n = p.numel()
            if self.steps[i] == 0:
                off += (n + 3) // 4 * 4
                continue                                  # torch creates the state at the first update
            state[i] = {"step": torch.tensor(float(self.steps[i])),
                        "exp_avg": self.exp_avg[off:off + n].view_as(p).clone(),
                        "exp_avg_sq": self.exp_avg_sq[off:off + n].view_as(p).clone()}
            off += (n + 3) // 4 * 4
        group = {k: v for k, v in self.param_groups[0].items() if k != "params"}   # incl. a scheduler's initial_lr
        group["params"] = list(range(len(self.params)))
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        off = 0
        for i, p in enumerate(self.params):
            n = p.numel()
            st = sd["state"].get(i)
            if st is not None:
                self.exp_avg[off:off + n].copy_(st["exp_avg"].reshape(-1))
                self.exp_avg_sq[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
                self.steps[i] = int(st["step"])
            else:       # absent from the checkpoint = never updated there: no stale moments survive a resume
                self.exp_avg[off:off + n].zero_()
                self.exp_avg_sq[off:off + n].zero_()
                self.steps[i] = 0
            off += (n + 3) // 4 * 4
        g = sd["param_groups"][0]
        for k, v in g.items():
            if k != "params":
                self.param_groups[0][k] = tuple(v) if k == "betas" else v
