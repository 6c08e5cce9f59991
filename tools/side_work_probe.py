"""What the work BESIDE the main stream costs the step (diagnostic, one process, alternating blocks timed by events):
   python tools/side_work_probe.py [blocks] [steps per block]          AB_SCENES=4: the 4-scene step
   full        the timed configuration: rulebooks + segment CSRs rebuilt every step (side stream), dW on its side stream
   no_rb       the rulebooks of the batch built once and attached (spconv.ops.RulebookPrefetcher's set), not rebuilt
   no_rb_csr   ... and the segment CSRs / edge graph built once
   dw_main     everything rebuilt, the weight gradients on the MAIN stream (WSIS_DW_STREAM=0: nothing overlaps the dIn products)
The differences are what the side work costs the step by sharing the GPU (or saves it by overlapping)."""
import importlib, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
importlib.import_module("3d-wsis_amd")
import torch
import harness

blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 4
per = int(sys.argv[2]) if len(sys.argv) > 2 else 30
cfg = harness.default_cfg()
dev = torch.device("cuda", 0)
n_scenes = int(os.environ.get("AB_SCENES", "1"))
cfg.batch_size = n_scenes
batch = harness.to_device(harness.collate([harness.bench_scene(1 + i) for i in range(n_scenes)]), dev)
model, crit, opt = harness.build_model(cfg, dev)
pre = harness.make_prefetcher(model)
harness.build_batch_graphs(batch)
harness.prefetch_rulebooks(pre, batch)
fixed_rb = pre.result()


def step(mode):
    if mode != "no_rb_csr":
        harness.build_batch_graphs(batch)
    if mode in ("no_rb", "no_rb_csr"):
        batch["rulebooks"] = fixed_rb
    else:
        batch.pop("rulebooks", None)
    if mode == "dw_main":
        os.environ["WSIS_DW_STREAM"] = "0"
    else:
        os.environ.pop("WSIS_DW_STREAM", None)
    harness.train_step(model, crit, opt, batch, cfg)


modes = ["full", "no_rb", "no_rb_csr", "dw_main"]
for m in modes:
    for _ in range(15):
        step(m)
for _ in range(200 if n_scenes == 1 else 40):
    step("full")
res = {m: [] for m in modes}
for b in range(blocks):
    for m in (modes if b % 2 == 0 else modes[::-1]):
        for _ in range(5):
            step(m)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(per):
            step(m)
        e1.record()
        torch.cuda.synchronize()
        res[m].append(e0.elapsed_time(e1) / per)
for m in modes:
    v = res[m]
    print(f"{m:10s} mean {sum(v) / len(v):7.3f} ms  ({' '.join('%.3f' % x for x in v)})")
