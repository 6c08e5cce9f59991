"""per-quantity error of the fused heads against the float64 module chains (debug aid for tests/test_gpu_heads.py)"""
import importlib, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
importlib.import_module("3d-wsis_amd")
import torch
import test_gpu_heads as T

S = int(sys.argv[1]) if len(sys.argv) > 1 else 2289
heads, lins, x = T._setup(S)
torch.manual_seed(7)
weights = [torch.randn(S, c) for c in (20, 3, 1, 1)] + [torch.randn(S, 64) for _ in lins]
ro, rdx, rh, rl = T._reference(heads, lins, x, weights)
go, gdx, gh, gl = T._run(heads, lins, x, weights)


def err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-6)


for i, (a, b) in enumerate(zip(go, ro)):
    print("out", i, err(a, b))
print("dx", err(gdx, rdx))
for i, (hg, hr) in enumerate(zip(gh, rh)):
    for (n, pg), (_, pr) in zip(hg.named_parameters(), hr.named_parameters()):
        print("head", i, n, err(pg.grad, pr.grad))
    print("head", i, "rm", err(hg[1].running_mean, hr[1].running_mean), "rv", err(hg[1].running_var, hr[1].running_var))
for i, (lg, lr) in enumerate(zip(gl, rl)):
    print("lin", i, err(lg.weight.grad, lr.weight.grad))
