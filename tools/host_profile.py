"""cProfile of the issuing thread over N training steps (device drained before each step, so the numbers are pure
issue time): which Python functions the host spends a step in.  python tools/host_profile.py [N]"""
import cProfile, importlib, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); importlib.import_module("3d-wsis_amd")
import torch, harness
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
cfg = harness.default_cfg()
model, crit, opt = harness.build_model(cfg, dev)
b = harness.to_device(harness.collate([harness.bench_scene(1)]), dev)
def step():
    harness.build_batch_graphs(b)
    harness.train_step(model, crit, opt, b, cfg)
for _ in range(100): step()
torch.cuda.synchronize()
pr = cProfile.Profile()
for _ in range(N):
    torch.cuda.synchronize()
    pr.enable(); step(); pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative")
import io
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(70)
for line in s.getvalue().splitlines():
    line = line.replace(ROOT + "/", "")
    print(line[:170])
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(30)
print("---- by own time")
for line in s.getvalue().splitlines()[6:]:
    print(line.replace(ROOT + "/", "")[:170])
