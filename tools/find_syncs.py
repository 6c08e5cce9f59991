"""Which device -> host reads does one training step make?  Wraps Tensor.item / tolist / cpu / __int__ / __bool__ /
__float__ / numpy / nonzero for ONE step and prints the call sites (how the per-step `edge_u.max()` read of EdgeGraph
was found: it drained the queue every step).  Expected output: only the four rulebook row counts (spconv/ops.py)."""
import importlib, sys, os, traceback, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); importlib.import_module("3d-wsis_amd")
import torch, harness
dev = torch.device('cuda:0')
cfg = harness.default_cfg()
model, crit, opt = harness.build_model(cfg, dev)
b = harness.to_device(harness.collate([harness.bench_scene(1)]), dev)
def step():
    harness.build_batch_graphs(b)
    harness.train_step(model, crit, opt, b, cfg)
for _ in range(5): step()
seen = collections.Counter()
def wrap(name):
    orig = getattr(torch.Tensor, name)
    def f(self, *a, **k):
        if self.is_cuda:
            st = traceback.extract_stack(limit=6)[:-1]
            seen[(name, " <- ".join("%s:%d" % (os.path.basename(s.filename), s.lineno) for s in reversed(st)))] += 1
        return orig(self, *a, **k)
    setattr(torch.Tensor, name, f)
for n in ("item", "tolist", "cpu", "__int__", "__bool__", "__float__", "numpy", "nonzero"):
    wrap(n)
step()
for k, v in seen.most_common(): print(v, k)
