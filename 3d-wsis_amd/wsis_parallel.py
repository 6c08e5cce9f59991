"""Scene-sharded data parallelism: one process per GPU, independent scenes per rank, one
bucketed gradient all-reduce per step over RCCL/xGMI (backend "nccl" on ROCm; "gloo" in the CPU tests).

The reference only wraps the model in DistributedDataParallel and never initialises a process group
(train_scannetv2.py:734-738); this is the working replacement.  Buckets are flat fp32 buffers in
reverse parameter order (gradients become ready in roughly that order), so a few large collectives replace
361 small ones -- sized for xGMI's per-link bandwidth rather than for launch count."""
import os

# RCCL's helper threads wait on HSA signals by interrupt unless told otherwise, which costs the launch-bound issuing
# thread of this path ~6 % with a process group up.  The HSA runtime reads the variable when it
# starts -- with the first GPU call of the process, torch.cuda.is_available() included -- so it is set here, at import,
# for a process that will join a group (torch.distributed.run exports WORLD_SIZE), unless the caller has chosen a value.
# Single-process runs are left alone.
if int(os.environ.get("WORLD_SIZE", "1")) > 1 or os.environ.get("WSIS_FORCE_DIST", "0") == "1":
    os.environ.setdefault("HSA_ENABLE_INTERRUPT", "0")

import torch
import torch.distributed as dist


def warm_streams(device=None):
    """gives every stream of a training step -- the current one, the rulebook / branch stream, the count-check stream and
    the library's weight-gradient stream -- its first command, i.e. its hardware queue.  HIP maps streams onto
    GPU_MAX_HW_QUEUES (4) hardware queues in the order of their first use; a process group created first puts RCCL's
    streams in between, and two streams of a step then share a queue (the one-rank RCCL line read 9.5 ms per step against
    7.9 without a group: tools/rccl_ab.sh).  ``init_distributed`` calls it before it creates the group."""
    if not torch.cuda.is_available():
        return
    import wsis_native as _n
    from spconv import ops as sp_ops
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    key = (dev.type, dev.index)
    chk = sp_ops._CHECK_STREAMS.get(key)
    if chk is None:
        chk = sp_ops._CHECK_STREAMS[key] = torch.cuda.Stream(device=dev)
    with torch.cuda.device(dev):
        _n.check(_n.hip().wsis_warm_streams(_n.stream_ptr()), "warm_streams")      # main + weight-gradient stream
        for st in (sp_ops._side_stream(dev), chk):
            with torch.cuda.stream(st):
                torch.zeros(1, device=dev).add_(1.0)
        torch.cuda.synchronize(dev)


def init_distributed(backend=None):
    """reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment (torch.distributed.run)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    force = os.environ.get("WSIS_FORCE_DIST", "0") == "1"    # exercise the collective path with one rank (tests)
    if (world > 1 or force) and not dist.is_initialized():
        if os.environ.get("HSA_ENABLE_INTERRUPT") != "0" and os.environ.get("WSIS_ALLOW_INTERRUPT_WAITS", "0") != "1":
            import warnings
            warnings.warn("wsis_parallel.init_distributed: HSA_ENABLE_INTERRUPT is not 0 (set before this module was "
                          "imported): RCCL's interrupt-driven signal waits cost the issuing thread ~6 % per step")
        if backend is None:   # WSIS_DIST_BACKEND=gloo: control-flow tests of N ranks on one GPU (RCCL refuses that)
            backend = os.environ.get("WSIS_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local_rank % max(torch.cuda.device_count(), 1))
            warm_streams()              # this process's streams take their hardware queues before RCCL's do
        import datetime
        # WSIS_DIST_TIMEOUT (seconds): a rank that dies or raises must not leave its peers blocked in a collective for
        # the backend's default of 10-30 minutes
        timeout = datetime.timedelta(seconds=int(os.environ.get("WSIS_DIST_TIMEOUT", "1800")))
        dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=timeout)
    return rank, local_rank, world


def shard_scenes(scene_ids, rank, world):
    """round-robin scene -> rank assignment (independent batch items, no data-path collective)"""
    return [s for i, s in enumerate(scene_ids) if i % world == rank]


def convert_sync_batchnorm(model, group=None):
    """torch.nn.SyncBatchNorm.convert_sync_batchnorm for this build (train_scannetv2.py:734-736 converts when
    num_gpus > 1): every BatchNorm1d keeps its class, parameters and state-dict names and is marked to take its batch
    statistics over all ranks of ``group``.  Heads and GNN layers go through wsis_ops._SyncBatchNormReLU (one all-gather
    forward, one all-reduce backward per layer); the UNet KEEPS the native executor: unet_native._BnSync issues its op
    list in parts (wsis_run_ops_part) that stop in front of every training-mode BatchNorm op and runs the layer's
    exchange between two parts.  Default of this build without the call: per-rank statistics.  Returns the model."""
    n = 0
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m._wsis_sync = True if group is None else group
            n += 1
    model._wsis_sync_bn = n > 0
    return model


def sync_batchnorm_active(model):
    """True when ``model`` -- or any module inside it: the conversion may have been applied to a wrapper or to a
    sub-module -- carries marked BatchNorm layers and a process group of more than one rank is up"""
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return False
    if getattr(model, "_wsis_sync_bn", False):
        return True
    return any(getattr(m, "_wsis_sync", None) is not None for m in model.modules())


class GradSync(object):
    """Averages gradients across ranks with flat buckets (default 16 MiB; 11.1 M fp32 parameters -> 3 buckets)."""

    def __init__(self, model, bucket_bytes=16 << 20, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        params = [p for p in model.parameters() if p.requires_grad]
        params.reverse()
        self.buckets, cur, size = [], [], 0
        for p in params:
            nbytes = p.numel() * p.element_size()
            if cur and size + nbytes > bucket_bytes:
                self.buckets.append(cur)
                cur, size = [], 0
            cur.append(p)
            size += nbytes
        if cur:
            self.buckets.append(cur)
        # path agreement (see _agree): the plan of the first calls is negotiated, then frozen
        self._agreed = None
        self._agree_calls = 0
        # gradient exchange under the backward pass (see enable_overlap)
        self._comm_stream = None
        self._early = None          # (handle, tensor, first element of the remainder) of this step's early all-reduce

    # ---- overlap: the native UNet backward reports when the first ~half of its flat gradient buffer is final
    # (csrc/executor.hip wsis_run_ops_marked); the all-reduce of that part is issued right there, on a communication
    # stream that waits for the milestone only, and runs under the second half of the pass (the reference
    # relies on DDP's bucket hooks for the same effect, train_scannetv2.py:738).  WSIS_SYNC_OVERLAP=0 switches it off.
    def enable_overlap(self, model):
        prog = getattr(model, "_native_prog", None)
        if prog is None or os.environ.get("WSIS_SYNC_OVERLAP", "1") == "0":
            return False
        dev = next(p for b in self.buckets for p in b).device
        if dev.type != "cuda":
            return False
        if self._comm_stream is None:
            self._comm_stream = torch.cuda.Stream(device=dev)
        prog.overlap = self
        return True

    def ready(self):
        """early exchange only once the collective plan is frozen on the flat path (all ranks issue the same calls)"""
        return (self._comm_stream is not None and self._agree_calls >= self.AGREE_CALLS
                and self._agreed is not None and self._agreed[0] > 0)

    def comm_stream_ptr(self):
        return self._comm_stream.cuda_stream

    def early(self, part, rest_first):
        """called from inside the backward pass: ``part`` (a view of the flat buffer) is final once the milestone the
        communication stream waits for has passed"""
        with torch.cuda.stream(self._comm_stream):
            h = dist.all_reduce(part, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._early = (h, part, rest_first)
        self.early_count = getattr(self, "early_count", 0) + 1

    @staticmethod
    def _flat_source(model):
        """(flat gradient buffer, ids of the parameters whose .grad are views of it) -- the native UNet pass
        (model/unet_native.py) writes its 10.96 M parameter gradients densely into one buffer, which is then
        all-reduced in place: no flatten / copy-back for 98 % of the gradient bytes"""
        prog = getattr(model, "_native_prog", None) if model is not None else None
        flat = getattr(prog, "flat_grad", None)
        if flat is None:
            return None, set()
        lo, hi = flat.data_ptr(), flat.data_ptr() + flat.numel() * flat.element_size()
        ok = [p for p in prog.flat_params if p.grad is not None and lo <= p.grad.data_ptr() < hi]
        if len(ok) != len(prog.flat_params):      # some gradient was replaced / accumulated elsewhere: fall back
            return None, set()
        return flat, {id(p) for p in ok}

    def _reserve_tail(self, model):
        """ask the native UNet pass for spare room behind its gradients: the remaining parameters' gradients are
        packed there, so a step needs ONE collective"""
        prog = getattr(model, "_native_prog", None) if model is not None else None
        if prog is not None and prog.tail_floats == 0 and os.environ.get("WSIS_SYNC_TAIL", "1") != "0":
            covered = {id(p) for p in prog.params}
            prog.tail_floats = sum(p.numel() for b in self.buckets for p in b if id(p) not in covered)

    AGREE_CALLS = 3     # the plan changes over the first steps (the tail is reserved in step 1, used from step 2)

    def _agree(self, plan):
        """The choice between the zero-copy flat all-reduce (+ tail packing) and the bucket path is made from
        rank-local state (do the gradients still live in the flat buffer? how large is the tail?).  Ranks that chose
        differently would issue a different number / size of collectives -- a hang or silent corruption -- so the
        choice is COLLECTIVE: during the first ``AGREE_CALLS`` calls every rank contributes its local plan
        (flat elements or 0, tail elements or 0) and the element-wise minimum / maximum over ranks decide: any rank
        without the flat path, or a disagreement on a size, puts every rank on the bucket path for that step.  That
        costs one tiny all-reduce and a host read per negotiated step; afterwards the plan is frozen and a rank whose
        local state no longer supports it fails loudly instead of diverging."""
        if self._agree_calls >= self.AGREE_CALLS:
            ok = self._agreed == (0, 0) or (self._agreed[0] == plan[0] and self._agreed[1] in (0, plan[1]))
            if not ok:
                raise RuntimeError(f"GradSync: this rank's gradient layout {plan} no longer matches the plan agreed "
                                   f"across ranks {self._agreed}; gradients were replaced or accumulated outside the "
                                   f"native UNet pass")
            return self._agreed
        self._agree_calls += 1
        dev = next(p for b in self.buckets for p in b).device
        t = torch.tensor([plan[0], plan[1], -plan[0], -plan[1]], dtype=torch.int64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
        lo_f, lo_t, hi_f, hi_t = (int(v) for v in t.tolist())
        hi_f, hi_t = -hi_f, -hi_t
        if lo_f != hi_f or lo_f == 0:
            self._agreed = (0, 0)                      # some rank cannot use the flat path: buckets everywhere
        elif lo_t != hi_t:
            self._agreed = (lo_f, 0)                   # flat buffer yes, tail sizes differ: no tail packing
        else:
            self._agreed = (lo_f, lo_t)
        return self._agreed

    def __call__(self, model=None):
        if self.world == 1 and os.environ.get("WSIS_FORCE_DIST", "0") != "1":
            return
        work = []
        flat_src, covered = self._flat_source(model)
        flat_handle = None
        self._reserve_tail(model)
        tail_job = None
        # rank-local plan -> collective decision
        local_tail = 0
        if flat_src is not None:
            prog = model._native_prog
            rest = [p for b in self.buckets for p in b if id(p) not in covered]
            n_rest = sum(p.numel() for p in rest)
            if prog.flat_tail is not None and prog.flat_tail.numel() == n_rest and n_rest > 0:
                local_tail = n_rest
        use_flat, use_tail = self._agree((int(flat_src.numel()) if flat_src is not None else 0, local_tail))
        if not use_flat:
            flat_src, covered = None, set()
        self.last_flat_params = len(covered)        # diagnostics / tests: parameters synchronised without a copy
        if flat_src is not None:
            if use_tail:
                grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in rest]
                torch.cat([g.reshape(-1) for g in grads], out=prog.flat_tail)
                tail_job = (rest, grads, prog.flat_tail)
                covered = covered | {id(p) for p in rest}
            early = self._early
            self._early = None
            remainder = flat_src
            if early is not None:       # the first part is already on the wire: exchange the remainder (+ tail)
                remainder = flat_src[early[2]:]
            flat_handle = dist.all_reduce(remainder, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            if early is not None:
                early[0].wait()
                torch.cuda.current_stream().wait_stream(self._comm_stream)
        # flatten every bucket with ONE cat kernel, all-reduce asynchronously, copy back with ONE multi-tensor copy
        for bucket in self.buckets:
            bucket = [p for p in bucket if id(p) not in covered]
            if not bucket:
                continue
            grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in bucket]
            flat = torch.cat([g.reshape(-1) for g in grads])
            handle = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            work.append((bucket, grads, flat, handle))
        for bucket, grads, flat, handle in work:
            handle.wait()
            if self.world > 1:
                flat.div_(self.world)
            views = [v.view_as(g) for v, g in zip(torch.split(flat, [g.numel() for g in grads]), grads)]
            torch._foreach_copy_(grads, views)
            for p, g in zip(bucket, grads):
                if p.grad is None:      # unused parameter on this rank: it still receives the averaged gradient
                    p.grad = g
        if flat_handle is not None:
            flat_handle.wait()
            if self.world > 1:
                flat_src.div_(self.world)
            if tail_job is not None:
                rest, grads, tail = tail_job
                views = [v.view_as(g) for v, g in zip(torch.split(tail, [g.numel() for g in grads]), grads)]
                torch._foreach_copy_(grads, views)
                for p, g in zip(rest, grads):
                    if p.grad is None:
                        p.grad = g
        if self._comm_stream is None and model is not None and self._agree_calls >= self.AGREE_CALLS:
            # plan frozen: from the next backward pass on, the first half of the flat buffer is exchanged from inside
            # the pass (no-op without a native UNet program, on CPU, or with WSIS_SYNC_OVERLAP=0)
            self.enable_overlap(model)
