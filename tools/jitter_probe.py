"""Where the step-to-step variation comes from: wall time vs GPU-event time per training step, in chunks of 20 steps,
next to a fixed GPU-only probe (device copy of 1 GiB) and a fixed host-only probe (Python loop).
   python tools/jitter_probe.py"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); importlib.import_module("3d-wsis_amd")
import numpy as np, torch, harness
dev = torch.device("cuda:0")
cfg = harness.default_cfg()
b = harness.to_device(harness.collate([harness.make_scene(1)]), dev)
model, crit, opt = harness.build_model(cfg, dev)
for _ in range(6):
    harness.train_step(model, crit, opt, b, cfg)
torch.cuda.synchronize()
src = torch.empty(1 << 28, device=dev); dst = torch.empty_like(src)
def gpu_probe():
    a = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(4): dst.copy_(src)
    e.record(); torch.cuda.synchronize(); return a.elapsed_time(e) / 4
def host_probe():
    t0 = time.perf_counter(); s = 0
    for i in range(200000): s += i * i
    return (time.perf_counter() - t0) * 1e3
for chunk in range(10):
    walls, gpus = [], []
    for _ in range(20):
        a = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); t0 = time.perf_counter(); a.record()
        harness.train_step(model, crit, opt, b, cfg)
        e.record(); torch.cuda.synchronize(); walls.append((time.perf_counter() - t0) * 1e3); gpus.append(a.elapsed_time(e))
    print(f"chunk {chunk}: wall median {np.median(walls):6.2f} gpu-event median {np.median(gpus):6.2f} | copy 1GiB {gpu_probe():6.3f} ms ({2*(1<<30)/gpu_probe()/1e6:6.0f} GB/s) | host loop {host_probe():6.2f} ms")
