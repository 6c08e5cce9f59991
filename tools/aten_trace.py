"""Which torch operators a training step still launches outside the native kernels, and where the step's GPU time goes.

1. TorchDispatchMode over one step: every aten op with its argument shapes and the innermost frame of this repository
   that issued it (forward; backward ops run from the autograd engine and carry the name of their node instead).
2. Events at the phase boundaries of an un-profiled step: UNet forward | gather + GNN + heads | loss | backward down to
   the UNet's output gradient | UNet backward | optimizer."""
import collections
import importlib
import os
import sys
import traceback

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
importlib.import_module("3d-wsis_amd")
import torch
from torch.utils._python_dispatch import TorchDispatchMode

import harness

if os.environ.get("AB_DIST", "0") == "1":      # a one-rank RCCL process group beside the step (no gradient exchange)
    os.environ["WSIS_FORCE_DIST"] = "1"
    import wsis_parallel
    wsis_parallel.init_distributed()

cfg = harness.default_cfg()
dev = torch.device("cuda", 0)
batch = harness.to_device(harness.collate([harness.bench_scene(1)]), dev)
model, crit, opt = harness.build_model(cfg, dev)
for _ in range(30):
    harness.build_batch_graphs(batch)
    harness.train_step(model, crit, opt, batch, cfg)
torch.cuda.synchronize()

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SKIP = ("aten.view", "aten.t.", "aten.transpose", "aten.detach", "aten.alias", "aten.expand", "aten.slice", "aten.select",
        "aten.unsqueeze", "aten.squeeze", "aten.reshape", "aten._unsafe_view", "aten.as_strided", "aten.permute",
        "aten.empty", "aten.sym_", "aten.is_", "aten.size", "aten.stride", "aten.unbind", "aten.split", "aten.narrow",
        "aten.lift_fresh", "aten._local_scalar", "aten.result_type", "aten.new_empty", "aten.empty_like",
        "aten.record_stream", "prim.")


class Log(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.rows = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            shapes = tuple(tuple(a.shape) if isinstance(a, torch.Tensor) else None for a in args)
            shapes = tuple(s for s in shapes if s is not None)
            on_gpu = any(isinstance(a, torch.Tensor) and a.is_cuda for a in args) or \
                any(isinstance(a, (list, tuple)) and any(isinstance(x, torch.Tensor) and x.is_cuda for x in a) for a in args)
            where = "?"
            for fr in reversed(traceback.extract_stack(limit=40)):
                if fr.filename.startswith(ROOT) and "aten_trace" not in fr.filename:
                    where = "%s:%d %s" % (os.path.relpath(fr.filename, ROOT), fr.lineno, fr.name)
                    break
            if on_gpu or name.startswith(("aten.zeros", "aten.ones", "aten.full", "aten.arange", "aten.tensor")):
                self.rows[(name, shapes, where)] += 1
        return func(*args, **(kwargs or {}))


log = Log()
with log:
    harness.build_batch_graphs(batch)
    harness.train_step(model, crit, opt, batch, cfg)
torch.cuda.synchronize()
print("== aten ops of one step (count, op, shapes, issuing frame) ==")
tot = 0
for (name, shapes, where), c in sorted(log.rows.items(), key=lambda kv: (kv[0][2], kv[0][0])):
    tot += c
    print("%3d  %-34s %-60s %s" % (c, name[:34], str(shapes)[:60], where))
print("total", tot)

# ---- phase stamps of un-profiled steps
import unet_native  # noqa: E402

import time  # noqa: E402

marks, host_t = {}, {}


def mark(name):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    marks.setdefault(name, []).append(e)
    host_t.setdefault(name, []).append(time.perf_counter())


orig_run = unet_native.run_unet


def run_unet(net, input, sync_group=None):
    mark("unet_fwd_begin")
    out = orig_run(net, input, sync_group=sync_group)
    mark("unet_fwd_end")
    if out.requires_grad:
        out.register_hook(lambda g: (mark("unet_bwd_begin"), g)[1])
    return out


unet_native.run_unet = run_unet
crit_fwd = crit.forward


def crit_forward(*a, **k):
    mark("loss_begin")
    r = crit_fwd(*a, **k)
    mark("loss_end")
    return r


crit.forward = crit_forward
opt_step = opt.step


def step_opt(*a, **k):
    mark("opt_begin")
    r = opt_step(*a, **k)
    mark("opt_end")
    return r


opt.step = step_opt
N = 20
torch.cuda.synchronize()
mark("origin")
for _ in range(N):
    mark("step_begin")
    harness.build_batch_graphs(batch)
    harness.train_step(model, crit, opt, batch, cfg)
mark("step_begin")
torch.cuda.synchronize()
order = ["step_begin", "unet_fwd_begin", "unet_fwd_end", "loss_begin", "loss_end", "unet_bwd_begin", "opt_begin", "opt_end"]
print("== phase durations, mean over %d un-profiled steps (us) ==" % N)
for a, b in zip(order[:-1], order[1:]):
    d = [marks[a][i].elapsed_time(marks[b][i]) * 1e3 for i in range(2, N)]
    print("%-16s -> %-16s %8.1f" % (a, b, sum(d) / len(d)))
d = [marks["opt_end"][i].elapsed_time(marks["step_begin"][i + 1]) * 1e3 for i in range(2, N)]
print("%-16s -> %-16s %8.1f" % ("opt_end", "next step_begin", sum(d) / len(d)))
d = [marks["step_begin"][i].elapsed_time(marks["step_begin"][i + 1]) * 1e3 for i in range(2, N)]
print("step %8.1f" % (sum(d) / len(d)))

# how far the host is ahead of the GPU at every mark (GPU time of the mark - host time of the mark, same origin): ~0 means
# the GPU reached the mark as soon as the host issued it, i.e. it was waiting for the host
print("== host lead at the marks, steps %d..%d (us): GPU reaches the mark this long after the host issued it ==" % (N - 4, N - 1))
e0, h0 = marks["origin"][0], host_t["origin"][0]
for name in order:
    lead = []
    for i in range(N - 4, N):
        g = e0.elapsed_time(marks[name][i]) * 1e3
        hh = (host_t[name][i] - h0) * 1e6
        lead.append(g - hh)
    print("%-16s %s" % (name, " ".join("%8.0f" % v for v in lead)))

print("== host time between the marks, mean over steps 2..%d (us): what issuing each phase costs the host ==" % (N - 1))
for a, b in zip(order[:-1], order[1:]):
    d = [(host_t[b][i] - host_t[a][i]) * 1e6 for i in range(2, N)]
    print("%-16s -> %-16s %8.1f" % (a, b, sum(d) / len(d)))
d = [(host_t["step_begin"][i + 1] - host_t["opt_end"][i]) * 1e6 for i in range(2, N)]
print("%-16s -> %-16s %8.1f" % ("opt_end", "next step_begin", sum(d) / len(d)))
d = [(host_t["step_begin"][i + 1] - host_t["step_begin"][i]) * 1e6 for i in range(2, N)]
print("host step %8.1f" % (sum(d) / len(d)))
