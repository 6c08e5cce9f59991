"""Load balance of the weight-gradient kernel's offset groups (a workgroup owns up to 8 of the 27 offsets of a 3x3x3
product and makes one 16-MFMA step per ACTIVE (32-row slice, offset) pair): steps per group for the grouping by
k mod 4 (rounds 2-6) and for the class-balanced partition, per level of the C2 pyramid (AB_SCENES=4: the C3 batch).
    python tools/dw_balance.py"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); importlib.import_module("3d-wsis_amd")
import torch, harness
from spconv import ops

OLD = [[k for k in range(27) if k % 4 == g] for g in range(4)]
NEW = [[0, 2, 8, 13, 14, 19, 24], [1, 5, 15, 16, 20, 21, 25], [6, 7, 9, 11, 17, 22, 23], [3, 4, 10, 12, 18, 26]]
dev = "cuda:0"
n_scenes = int(os.environ.get("AB_SCENES", "1"))
b = harness.collate([harness.bench_scene(1 + i) for i in range(n_scenes)])
idx = b["voxel_locs"].int().to(dev).contiguous(); shape = [int(s) for s in b["spatial_shape"]]
for level in range(5):
    rb = ops.build_subm_rulebook(idx, shape, [3] * 3, [1] * 3)
    M = idx.shape[0]
    nbr = rb.nbr.view(27, M)
    order = rb.order.long()
    pad = (-M) % 32
    have = (nbr[:, order] >= 0)
    if pad:
        have = torch.cat([have, torch.zeros(27, pad, dtype=torch.bool, device=dev)], 1)
    act = have.view(27, -1, 32).any(2).float().sum(1).cpu()          # active slices per offset
    line = f"level {level}: {M:7d} rows, {int((M + 31) // 32):6d} slices, active (slice, offset) pairs {int(act.sum()):7d} "
    for name, part in (("k mod 4", OLD), ("balanced", NEW)):
        loads = [float(sum(act[k] for k in g)) for g in part]
        line += f"| {name}: " + " ".join("%6d" % l for l in loads) + "  max/mean %.3f " % (max(loads) / (sum(loads) / 4))
    print(line)
    if os.environ.get("DW_BALANCE_VERBOSE"):
        print("   act", [int(v) for v in act])
    if level < 4:
        rd = ops.build_down_rulebook(idx, shape, [2] * 3, [2] * 3, [0] * 3); idx, shape = rd.out_indices, rd.out_shape
