"""Is the training step waiting for the GPU or for the issuing thread?  N steps issued back to back (as bench.py does):
host time to ISSUE them, then the wait in the final synchronize.  A wait near zero = the host is the bottleneck; a wait
of several steps = the GPU is.  Also: the same N steps with a synchronize after every step (host issue time per step when
it cannot run ahead = the host cost of a step).
    python tools/host_probe.py [steps]      (AB_SCENES=4: bench.py's C3 batch)"""
import importlib, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
importlib.import_module("3d-wsis_amd")
import torch
import harness

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
cfg = harness.default_cfg()
dev = torch.device("cuda", 0)
n_scenes = int(os.environ.get("AB_SCENES", "1"))
cfg.batch_size = n_scenes
batch = harness.to_device(harness.collate([harness.bench_scene(1 + i) for i in range(n_scenes)]), dev)
model, crit, opt = harness.build_model(cfg, dev)


def step():
    harness.build_batch_graphs(batch)
    harness.train_step(model, crit, opt, batch, cfg)


for _ in range(300 if n_scenes == 1 else 60):
    step()
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    marks = []
    for _ in range(n):
        step()
        marks.append(time.perf_counter())
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    d = [(marks[i] - (marks[i - 1] if i else t0)) * 1e3 for i in range(n)]
    print("back to back: %d steps issued in %.2f ms (%.3f ms per step; first 5: %s; last 5: %s), final wait %.2f ms, "
          "total %.3f ms per step" % (n, (t1 - t0) * 1e3, (t1 - t0) * 1e3 / n, " ".join("%.2f" % x for x in d[:5]),
                                      " ".join("%.2f" % x for x in d[-5:]), (t2 - t1) * 1e3, (t2 - t0) * 1e3 / n))
iss, tot = [], []
for _ in range(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    iss.append((t1 - t0) * 1e3)
    tot.append((t2 - t0) * 1e3)
iss.sort(); tot.sort()
print("one step at a time: host issue median %.3f ms (p10 %.3f, p90 %.3f); issue + wait median %.3f ms"
      % (iss[n // 2], iss[n // 10], iss[9 * n // 10], tot[n // 2]))
