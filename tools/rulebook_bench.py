"""Wall time of building the 9 rulebooks (13 gather tables) of the C2 pyramid, batched vs per-table tile orders:
   python tools/rulebook_bench.py"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); importlib.import_module("3d-wsis_amd")
import numpy as np, torch, harness, spconv
from spconv import ops
b = harness.to_device(harness.collate([harness.make_scene(1)]), "cuda")
idx, shape = b["voxel_coords_int"], b["spatial_shape"]
feat = torch.zeros(idx.shape[0], 1, device="cuda")
def build():
    t = spconv.SparseConvTensor(feat, idx, shape, 1)
    if os.environ.get("RB_COUNTS", "1") != "0" and b.get("level_counts") is not None:
        t._level_counts = b["level_counts"]      # the loader's host-side counts: the whole pyramid from ONE native call
    ops.prebuild_unet_rulebooks(t, 5, side_stream=False)
    return t
for flag in ("1", "0", "1", "0"):
    os.environ["WSIS_TILE_BATCH"] = flag
    for _ in range(5): build()
    torch.cuda.synchronize()
    ts = []
    for _ in range(40):
        torch.cuda.synchronize(); t0 = time.perf_counter(); build(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"WSIS_TILE_BATCH={flag}: median {np.median(ts):.3f} ms  min {min(ts):.3f} ms")
