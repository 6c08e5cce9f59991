import importlib, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); importlib.import_module("3d-wsis_amd")
import torch, harness
from spconv import ops
dev="cuda:0"
b = harness.collate([harness.bench_scene(1)])
idx = b["voxel_locs"].int().to(dev).contiguous(); shape=[int(s) for s in b["spatial_shape"]]
rb = ops.build_subm_rulebook(idx, shape, [3]*3, [1]*3)
M=idx.shape[0]
X=torch.randn(M,6,device=dev); W=torch.randn(27,6,32,device=dev)
def timeit(f,n=50):
    for _ in range(5): f()
    torch.cuda.synchronize(); s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/n*1e3
for v in ("0","1"):
    os.environ["WSIS_IN_CONV"]=v
    print("WSIS_IN_CONV",v, timeit(lambda: ops._conv(X, rb.nbr_p, rb.order, W, None, None, M)), "us")
dY = torch.randn(M, 32, device=dev)
for v in ("0", "1"):
    os.environ["WSIS_IN_CONV"] = v
    print("dW WSIS_IN_CONV", v, timeit(lambda: ops._dw(X, rb.nbr_p, rb.order, dY, 27, 6, 32)), "us (kernel + slab sum + host)")
