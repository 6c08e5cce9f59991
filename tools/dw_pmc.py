"""One level of the C2 pyramid, weight gradient only, for PMC passes of spconv_dw2_kernel:
   rocprofv3 --pmc <counters> --kernel-trace --output-format csv -d <out> -- python3 tools/dw_pmc.py <level>
   python tools/dw_pmc.py --parse <out>"""
import importlib, sys, os, glob, csv
if len(sys.argv) > 2 and sys.argv[1] == "--parse":
    acc = {}
    for f in glob.glob(os.path.join(sys.argv[2], "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            for kn in ("spconv_dw3_kernel", "spconv_dw2_kernel", "dw2_reduce_kernel"):
                if kn in r["Kernel_Name"]:
                    acc.setdefault((kn, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(f"{k[0]:20s} {k[1]:32s} launches {len(v):4d}  mean {sum(v)/len(v):16.1f}")
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); importlib.import_module("3d-wsis_amd")
import torch, harness
from spconv import ops
level = int(sys.argv[1]) if len(sys.argv) > 1 else 0
dev = 'cuda:0'
b = harness.collate([harness.make_scene(1)])
idx = b['voxel_locs'].int().to(dev).contiguous(); shape = [int(s) for s in b['spatial_shape']]
for l in range(level):
    rd = ops.build_down_rulebook(idx, shape, [2]*3, [2]*3, [0]*3); idx, shape = rd.out_indices, rd.out_shape
rb = ops.build_subm_rulebook(idx, shape, [3]*3, [1]*3)
C = 32 * (level + 1); M = idx.shape[0]
X = torch.randn(M, C, device=dev); dY = torch.randn(M, C, device=dev)
for _ in range(10):
    ops._dw(X, rb.nbr_p, rb.order, dY, 27, C, C)
torch.cuda.synchronize()
