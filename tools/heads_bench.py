"""kernel times of the fused heads (csrc/heads.hip) at a few shapes: run under rocprofv3 --kernel-trace --stats"""
import importlib, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
importlib.import_module("3d-wsis_amd")
import torch
import wsis_ops
import test_gpu_heads as T

for S, couts, n_lin in ((2289, (20, 3, 1, 1), 3), (2289, (7,), 0), (9156, (20, 3, 1, 1), 3)):
    heads, lins, x = T._setup(S, couts=couts, n_lin=n_lin)
    hg = [h.cuda() for h in heads]
    lg = [l.cuda() for l in lins]
    xg = x.cuda().requires_grad_(True)
    ws = [torch.randn(S, c, device="cuda") for c in couts] + [torch.randn(S, 64, device="cuda") for _ in lins]
    for _ in range(30):
        a, b = wsis_ops.sp_heads(xg, hg, lg)
        loss = sum((o * w).sum() for o, w in zip(a + b, ws))
        loss.backward()
    torch.cuda.synchronize()
    print("done", S, couts, n_lin)
