"""What issuing a step costs the HOST, piece by piece (GPU box):
   python tools/host_issue.py
1. raw launch cost: N trivial launches through hipLaunchKernelGGL / hipExtLaunchKernelGGL (small / 120-byte argument block),
   host microseconds per launch while the queue is never empty;
2. the UNet executor: host time inside wsis_run_ops (forward / backward) with the weight-gradient side stream on a worker
   thread (default), on the calling thread (WSIS_DW_THREAD=0) and off (WSIS_DW_STREAM=0);
3. host time of the phases of a step with the GPU drained in front of every step."""
import ctypes, importlib, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
importlib.import_module("3d-wsis_amd")
import torch
import harness
import wsis_native as _n

dev = torch.device("cuda", 0)
lib = _n.hip()

# ---- 1. raw launches
fn = lib.wsis_debug_gap_probe
fn.restype = ctypes.c_int32
fn.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]
buf = torch.zeros(64, device=dev)
for variant, name in ((0, "hipLaunchKernelGGL, 8-byte args"), (16, "hipExtLaunchKernelGGL, 8-byte args"),
                      (8, "hipLaunchKernelGGL, 160-byte args"), (24, "hipExtLaunchKernelGGL, 160-byte args"),
                      (1, "hipLaunchKernelGGL, 32 KB dynamic LDS"), (17, "hipExtLaunchKernelGGL, 32 KB dynamic LDS")):
    fn(variant, 200, buf.data_ptr(), _n.stream_ptr())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(variant, 3000, buf.data_ptr(), _n.stream_ptr())
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"raw launch  {name:42s}: host {(t1 - t0) / 3000 * 1e6:5.2f} us per launch (device drained {(t2 - t0) / 3000 * 1e6:5.2f})")

# ---- 2. / 3. the step
cfg = harness.default_cfg()
batch = harness.to_device(harness.collate([harness.bench_scene(1)]), dev)
model, crit, opt = harness.build_model(cfg, dev)
import unet_native

acc = {}
orig_run = unet_native._run


def timed_run(lib_, ops, device, mark_op=-1, waiter=None):
    t0 = time.perf_counter()
    r = orig_run(lib_, ops, device, mark_op, waiter)
    key = "run_ops[%d ops]" % len(ops)
    acc.setdefault(key, []).append(time.perf_counter() - t0)
    return r


unet_native._run = timed_run


def step():
    harness.build_batch_graphs(batch)
    harness.train_step(model, crit, opt, batch, cfg)


for _ in range(40):
    step()
for label, env in (("default (dW on a worker thread)", {}), ("WSIS_DW_THREAD=0", {"WSIS_DW_THREAD": "0"}),
                   ("WSIS_DW_STREAM=0", {"WSIS_DW_STREAM": "0"})):
    for k, v in env.items():
        os.environ[k] = v
    for _ in range(5):
        step()
    acc.clear()
    ts = []
    for _ in range(20):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step()
        ts.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    print(f"{label}: host step {sum(ts) / len(ts) * 1e3:.3f} ms (min {min(ts) * 1e3:.3f})")
    for k, v in acc.items():
        print(f"    {k}: {sum(v) / len(v) * 1e3:.3f} ms per call, {len(v) // 20} call(s) per step")
    for k in env:
        del os.environ[k]
