"""What does a dependent kernel boundary cost here?  Chains of (almost) empty kernels of different shapes, timed with
events (wsis_debug_gap_probe in csrc/core.hip).  The conv kernels of a step sit behind 5-6 us gaps in the rocprofv3
timeline while other neighbours show 0: is it the dynamic LDS, the workgroup shape, or the argument block?"""
import ctypes
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
importlib.import_module("3d-wsis_amd")
import torch
import wsis_native as _n

lib = _n.hip()
fn = lib.wsis_debug_gap_probe
fn.restype = ctypes.c_int32
fn.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]
buf = torch.zeros(64, device="cuda")
N = 200
names = {0: "256x256 threads, no LDS", 1: "256x256, 32 KB dynamic LDS", 2: "4800x64, no LDS", 3: "4800x64, 32 KB LDS",
         5: "256x256, alternating 32 KB / 0", 8: "256x256, 160-byte argument struct", 9: "256x256, struct + LDS"}
for variant, name in names.items():
    for rep in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        _n.check(fn(variant, 20, buf.data_ptr(), _n.stream_ptr()), "probe")
        torch.cuda.synchronize()
        a.record()
        _n.check(fn(variant, N, buf.data_ptr(), _n.stream_ptr()), "probe")
        b.record()
        torch.cuda.synchronize()
    print(f"variant {variant:2d} {name:40s}: {a.elapsed_time(b) * 1e3 / N:6.2f} us per launch")
