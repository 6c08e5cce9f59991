"""Step time of the C2 training step against the CPU set the process runs on (NUMA placement of the host thread):
   python tools/affinity_probe.py"""
import glob, importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); importlib.import_module("3d-wsis_amd")
import numpy as np, torch, harness

def cpulist(s):
    out = []
    for part in s.strip().split(","):
        if "-" in part:
            a, b = part.split("-"); out += list(range(int(a), int(b) + 1))
        elif part:
            out.append(int(part))
    return out

nodes = {}
for d in sorted(glob.glob("/sys/devices/system/node/node[0-9]*")):
    nodes[int(d.rsplit("node", 1)[1])] = cpulist(open(d + "/cpulist").read())
print("numa nodes:", {k: (v[0], v[-1], len(v)) for k, v in nodes.items()})
print("affinity now:", len(os.sched_getaffinity(0)), "cpus")

dev = torch.device("cuda:0")
try:
    pr = torch.cuda.get_device_properties(0)
    bdf = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
    print("gpu", bdf, "numa_node", open(f"/sys/bus/pci/devices/{bdf}/numa_node").read().strip())
except Exception as e:
    print("gpu numa lookup failed:", e)
cfg = harness.default_cfg()
b = harness.to_device(harness.collate([harness.make_scene(1)]), dev)
model, crit, opt = harness.build_model(cfg, dev)
for _ in range(6):
    harness.train_step(model, crit, opt, b, cfg)
torch.cuda.synchronize()

def run(tag, n=30):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        harness.train_step(model, crit, opt, b, cfg)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    ts = np.array(ts)
    print(f"{tag:28s} median {np.median(ts):6.2f}  mean {ts.mean():6.2f}  min {ts.min():6.2f}  max {ts.max():6.2f}")

import gc
full = os.sched_getaffinity(0)
run("default")
gc.collect(); gc.disable()
run("gc disabled")
run("gc disabled again")
gc.enable()
run("default again")
for k, cpus in nodes.items():
    cp = [c for c in cpus if c in full]
    if not cp:
        continue
    os.sched_setaffinity(0, cp); run(f"node {k} ({len(cp)} cpus)")
    os.sched_setaffinity(0, cp[:1]); run(f"node {k} single cpu {cp[0]}")
os.sched_setaffinity(0, full)
run("default restored")
