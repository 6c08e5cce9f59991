"""tiny dense case through the ring kernel: which rows / columns come out wrong"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); importlib.import_module("3d-wsis_amd")
import torch
from spconv import ops
dev = 'cuda:0'
M, cin, cout = int(sys.argv[1]) if len(sys.argv) > 1 else 96, 32, 32
os.environ["WSIS_RING_MIN_ITEMS"] = "1"
X = torch.arange(M, device=dev, dtype=torch.float32)[:, None].repeat(1, cin) + 1
W = torch.zeros(1, cin, cout, device=dev); W[0] = torch.eye(cin, device=dev)[:, :cout]
WT = ops._weight_t(W, 0)
for nt in (1, 4):
    os.environ["WSIS_RING"] = "1"; os.environ["WSIS_RING_NT"] = str(nt)
    out = ops._conv_t(X, None, None, WT, 0, None, None, M)
    torch.cuda.synchronize()
    want = X[:, :cout]
    bad = (out != want)
    print(f"nt={nt}: wrong elements {int(bad.sum())} of {out.numel()}; rows wrong: {bad.any(1).nonzero().flatten()[:20].tolist()}")
    print(out[:4, :6].tolist(), out[32:34, :6].tolist())
