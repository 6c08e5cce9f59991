"""Where the backward pass spends HOST time: wall time inside every custom autograd Function's forward / backward
(summed per class, per step), next to the whole phase.  The device is drained before each step, so the numbers are
issue time.  python tools/host_bwd_profile.py [steps]"""
import importlib, os, sys, time, inspect
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); importlib.import_module("3d-wsis_amd")
import torch, harness
import wsis_ops, unet_native, pointgroup_ops, torch_scatter, losses_3D_WSIS
from spconv import ops as sp_ops
N = int(sys.argv[1]) if len(sys.argv) > 1 else 50
acc = {}
def wrap(cls, which):
    fn = getattr(cls, which)
    def timed(*a, **k):
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            e = acc.setdefault((cls.__name__, which), [0.0, 0])
            e[0] += time.perf_counter() - t; e[1] += 1
    setattr(cls, which, staticmethod(timed))
for mod in (wsis_ops, unet_native, pointgroup_ops, torch_scatter, losses_3D_WSIS, sp_ops):
    for name, cls in inspect.getmembers(mod, inspect.isclass):
        if issubclass(cls, torch.autograd.Function) and cls is not torch.autograd.Function and cls.__module__ == mod.__name__:
            wrap(cls, "forward"); wrap(cls, "backward")
dev = torch.device("cuda:0")
cfg = harness.default_cfg()
model, crit, opt = harness.build_model(cfg, dev)
b = harness.to_device(harness.collate([harness.bench_scene(1)]), dev)
T = {"fwd": 0.0, "bwd": 0.0}
def step(rec):
    harness.build_batch_graphs(b)
    t0 = time.perf_counter()
    loss, _ = harness.forward_loss(model, crit, b, cfg)
    t1 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    loss.backward()
    t2 = time.perf_counter()
    opt.step()
    if rec:
        T["fwd"] += t1 - t0; T["bwd"] += t2 - t1
for _ in range(100): step(False)
acc.clear()
for _ in range(N):
    torch.cuda.synchronize(); step(True)
print("phases ms/step:", {k: round(v / N * 1e3, 2) for k, v in T.items()})
tot = {"forward": 0.0, "backward": 0.0}
for (name, which), (t, n) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    tot[which] += t
    print(f"  {name:28s} {which:8s} {t / N * 1e3:6.3f} ms/step  {n / N:5.1f} calls/step  {t / n * 1e6:7.1f} us/call")
print("inside custom Functions:", {k: round(v / N * 1e3, 2) for k, v in tot.items()})
