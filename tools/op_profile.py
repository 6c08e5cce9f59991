"""Which torch operators launch the small at::native kernels of a training step (torch.profiler, one step)."""
import importlib, os, sys, collections
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
importlib.import_module("3d-wsis_amd")
import torch, harness
from torch.profiler import profile, ProfilerActivity

cfg = harness.default_cfg()
dev = torch.device("cuda", 0)
batch = harness.to_device(harness.collate([harness.bench_scene(1)]), dev)
model, crit, opt = harness.build_model(cfg, dev)
for _ in range(20):
    harness.build_batch_graphs(batch)
    harness.train_step(model, crit, opt, batch, cfg)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    harness.build_batch_graphs(batch)
    harness.train_step(model, crit, opt, batch, cfg)
    torch.cuda.synchronize()
ev = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith("aten::")]
cnt = collections.Counter()
tim = collections.Counter()
for e in ev:
    k = [c for c in e.kernels] if hasattr(e, "kernels") else []
    if k:
        cnt[e.name] += len(k)
        tim[e.name] += sum(x.duration for x in k)
print("aten op -> launched kernels, device us")
for n, c in cnt.most_common(40):
    print(f"{c:5d} {tim[n]:9.1f}  {n}")
# stacks of the most frequent small ops
for name in ("aten::copy_", "aten::fill_", "aten::add_", "aten::mul", "aten::clamp_", "aten::cat", "aten::index", "aten::zero_"):
    st = collections.Counter()
    for e in ev:
        if e.name == name and getattr(e, "kernels", None):
            fr = [s for s in (e.stack or []) if "3d-wsis_amd" in s or "harness" in s]
            st[fr[0] if fr else "?"] += 1
    if st:
        print("--", name)
        for s, c in st.most_common(8):
            print(f"   {c:3d}  {s[-110:]}")
