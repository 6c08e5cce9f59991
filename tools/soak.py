"""soak: training steps over scenes of varying size, twice with the same seeds: identical loss sequences, no NaN
(SOAK_STEPS, SOAK_NOSYNC=1: no device read-back inside the loop)"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); importlib.import_module("3d-wsis_amd")
import torch, harness
cfg = harness.default_cfg()
rooms = [(8.0, 6.5, 2.8), (5.0, 4.0, 2.6), (3.0, 2.5, 2.4), (1.8, 1.4, 1.1), (6.5, 6.0, 2.8), (4.2, 3.1, 2.5), (2.2, 2.0, 1.6), (7.3, 5.1, 2.7)]
scenes = [harness.collate([harness.make_scene(100 + i, room=r, n_box=3 + i % 4)]) for i, r in enumerate(rooms)]
scenes.append(harness.collate([harness.make_scene(200 + i, room=rooms[i % 3 + 1], n_box=3) for i in range(3)]))   # a batch of 3
def run():
    torch.manual_seed(0)
    model, crit, opt = harness.build_model(cfg, "cuda")
    out = []
    for it in range(int(os.environ.get("SOAK_STEPS", "90"))):
        b = harness.to_device(scenes[it % len(scenes)], "cuda")
        loss, _ = harness.train_step(model, crit, opt, b, cfg)
        # SOAK_NOSYNC=1: the losses are read at the end -- the host runs steps ahead of the GPU, as in bench.py
        out.append(loss.detach() if os.environ.get("SOAK_NOSYNC", "0") == "1" else float(loss))
    return [float(x) for x in out]
t0 = time.time(); a = run(); b = run()
import math
print("steps", len(a), "nan", sum(math.isnan(x) for x in a), "identical", a == b, "first/last", a[0], a[-1], "time %.1fs" % (time.time() - t0))
if a != b:
    bad = [i for i, (x, y) in enumerate(zip(a, b)) if x != y]; print("first differing steps", bad[:10], [(a[i], b[i]) for i in bad[:3]])
