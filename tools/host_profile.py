"""Host-side (Python / dispatcher) cost of one training step, by operator: torch.profiler, CPU activity only.
   python tools/host_profile.py [n_rows]"""
import importlib, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); importlib.import_module("3d-wsis_amd")
import torch, harness
from torch.profiler import profile, ProfilerActivity
dev = torch.device('cuda:0')
cfg = harness.default_cfg()
b = harness.to_device(harness.collate([harness.make_scene(1)]), dev)
model, crit, opt = harness.build_model(cfg, dev)
for _ in range(3):
    harness.train_step(model, crit, opt, b, cfg)
torch.cuda.synchronize()
N = 5
t0 = time.perf_counter()
for _ in range(N):
    harness.train_step(model, crit, opt, b, cfg)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host issue {(t1-t0)/N*1e3:.2f} ms/step, wall {(t2-t0)/N*1e3:.2f} ms/step")
with profile(activities=[ProfilerActivity.CPU]) as prof:
    for _ in range(N):
        harness.train_step(model, crit, opt, b, cfg)
    torch.cuda.synchronize()
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 45
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=rows, max_name_column_width=48))
