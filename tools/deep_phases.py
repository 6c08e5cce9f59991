"""Per-phase times of the resident deep-level launches of one training step on the C2 scene (in-kernel s_memrealtime
stamps: end of the previous phase -> end of this phase, grid barrier included).
  python tools/deep_phases.py [scene_seed]"""
import ctypes
import importlib
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
importlib.import_module("3d-wsis_amd")
import torch  # noqa: E402

import harness  # noqa: E402
import wsis_native as _n  # noqa: E402
from spconv import ops as sp_ops  # noqa: E402

KINDS = {1: "conv", 2: "slabsum", 3: "bn_fwd", 4: "bn_bwd", 5: "cat", 6: "split"}


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    os.environ["WSIS_DW_STREAM"] = "0"
    cfg = harness.default_cfg()
    dev = torch.device("cuda", 0)
    batch = harness.to_device(harness.collate([harness.bench_scene(seed)]), dev)
    model, crit, opt = harness.build_model(cfg, dev)
    for _ in range(30):
        harness.train_step(model, crit, opt, batch, cfg)
    sp_ops.PROFILER = sp_ops.KernelProfiler()
    harness.train_step(model, crit, opt, batch, cfg)
    torch.cuda.synchronize()
    lib = _n.hip()
    lib.wsis_debug_deep_phases.restype = ctypes.c_int32
    cap = 1024
    info = (ctypes.c_int32 * (cap * 8))()
    us = (ctypes.c_double * cap)()
    for which in range(8):
        n = lib.wsis_debug_deep_phases(which, info, us, cap)
        if n < 0:
            break
        tot = sum(us[i] for i in range(n))
        print(f"--- resident launch {which}: {n} phases, {tot:.1f} us")
        agg = {}
        for i in range(n):
            k, nw, zs, rows, cin, cout, K, mg = (info[i * 8 + j] for j in range(8))
            print(f"  {i:3d} {KINDS.get(k, k):8s} rows {rows:6d} gathered {mg:6d} K {K:2d} {cin:3d}->{cout:3d} NW {nw:2d} ZS {zs}  {us[i]:7.2f} us")
            key = (KINDS.get(k, k), rows)
            a = agg.setdefault(key, [0, 0.0])
            a[0] += 1
            a[1] += us[i]
        for key, (c, t) in sorted(agg.items()):
            print(f"  sum {key[0]:8s} rows {key[1]:6d}: {c:3d} phases {t:8.1f} us  ({t / c:.1f} each)")
    sp_ops.PROFILER.summary()
    sp_ops.PROFILER = None


if __name__ == "__main__":
    main()
