"""host time of one training step: wall time per step of the issuing thread with the GPU drained between steps (so the
host never waits for the device), and a cProfile of 20 such steps (top entries by own time)"""
import cProfile, importlib, os, pstats, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
importlib.import_module("3d-wsis_amd")
import torch
import harness

cfg = harness.default_cfg()
dev = torch.device("cuda", 0)
batch = harness.to_device(harness.collate([harness.bench_scene(1)]), dev)
model, crit, opt = harness.build_model(cfg, dev)


def step():
    harness.build_batch_graphs(batch)
    harness.train_step(model, crit, opt, batch, cfg)


for _ in range(60):
    step()
torch.cuda.synchronize()
ts = []
for _ in range(30):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    ts.append(time.perf_counter() - t0)
print("host ms per step (issue only): mean %.3f  min %.3f" % (sum(ts) / len(ts) * 1e3, min(ts) * 1e3))
pr = cProfile.Profile()
for _ in range(20):
    torch.cuda.synchronize()
    pr.enable()
    step()
    pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(35)
