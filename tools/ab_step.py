"""A/B of two environment settings inside ONE process: blocks of steps alternate between the settings (every switch is
read per call or per pass), each block timed by events; removes the box-to-box and clock-ramp spread of separate runs.
    python tools/ab_step.py WSIS_BRANCH=0 WSIS_BRANCH=1 [blocks] [steps per block]
AB_SCENES=4 selects bench.py's C3 batch (scenes 1-4 per step) instead of the C2 scene."""
import importlib, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
importlib.import_module("3d-wsis_amd")
import torch
import harness

if os.environ.get("AB_DIST", "0") == "1":      # a one-rank RCCL process group beside the step (no gradient exchange)
    os.environ["WSIS_FORCE_DIST"] = "1"
    import wsis_parallel
    wsis_parallel.init_distributed()


def setenv(spec):
    for kv in spec.split(","):
        k, v = kv.split("=")
        os.environ[k] = v


A, B = sys.argv[1], sys.argv[2]
blocks = int(sys.argv[3]) if len(sys.argv) > 3 else 6
per = int(sys.argv[4]) if len(sys.argv) > 4 else 40
cfg = harness.default_cfg()
dev = torch.device("cuda", 0)
n_scenes = int(os.environ.get("AB_SCENES", "1"))
cfg.batch_size = n_scenes
batch = harness.to_device(harness.collate([harness.bench_scene(1 + i) for i in range(n_scenes)]), dev)
model, crit, opt = harness.build_model(cfg, dev)


def step():
    harness.build_batch_graphs(batch)
    harness.train_step(model, crit, opt, batch, cfg)


for spec in (A, B):
    setenv(spec)
    for _ in range(20):
        step()
for _ in range(260 if n_scenes == 1 else 60):
    step()
res = {A: [], B: []}
for b in range(blocks):
    for spec in ((A, B) if b % 2 == 0 else (B, A)):
        setenv(spec)
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(per):
            step()
        e1.record()
        torch.cuda.synchronize()
        res[spec].append(e0.elapsed_time(e1) / per)
for spec in (A, B):
    v = res[spec]
    print("%-40s mean %.3f ms  (%s)" % (spec, sum(v) / len(v), " ".join("%.3f" % x for x in v)))
