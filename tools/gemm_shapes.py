"""Which torch GEMMs does a training step still launch, with what shapes and for how long (torch profiler, shapes
recorded): finds Linear layers whose weight gradient goes through a one-workgroup hipBLASLt product.
   python tools/gemm_shapes.py"""
import importlib, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); importlib.import_module("3d-wsis_amd")
import torch, harness
from torch.profiler import profile, ProfilerActivity
dev = torch.device('cuda:0')
cfg = harness.default_cfg()
model, crit, opt = harness.build_model(cfg, dev)
b = harness.to_device(harness.collate([harness.bench_scene(1)]), dev)
for _ in range(5):
    harness.train_step(model, crit, opt, b, cfg)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(3):
        harness.train_step(model, crit, opt, b, cfg)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    if e.key in ("aten::mm", "aten::addmm", "aten::bmm", "aten::matmul", "aten::sum", "aten::copy_", "aten::cat", "aten::mul", "aten::add"):
        dt = getattr(e, "device_time_total", None) or getattr(e, "cuda_time_total", 0)
        rows.append((dt / 3.0, e.count / 3.0, e.key, str(e.input_shapes)))
rows.sort(reverse=True)
print("us/step  calls/step  op  shapes")
for r in rows[:40]:
    print(f"{r[0]:8.1f} {r[1]:6.1f}  {r[2]:12s} {r[3][:150]}")
