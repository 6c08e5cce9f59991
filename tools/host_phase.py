"""Host side of one training step: wall time the host spends in each phase (graph builds, forward + loss, backward,
optimizer) in steady state -- where it also waits for the device in the rulebook row-count reads -- and with a device
synchronisation before every step (pure issue time, the GPU never ahead).  `python tools/host_phase.py` on the GPU box;
DESIGN section 5 quotes its numbers (step 10.6 ms, host 9.4 ms + ~1 ms of waits)."""
import importlib, sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); importlib.import_module("3d-wsis_amd")
import torch, harness
dev = torch.device('cuda:0')
cfg = harness.default_cfg()
model, crit, opt = harness.build_model(cfg, dev)
b = harness.to_device(harness.collate([harness.bench_scene(1)]), dev)
T = {"graphs":0.,"fwd":0.,"bwd":0.,"opt":0.}
def step(acc=False):
    t0=time.perf_counter()
    harness.build_batch_graphs(b)
    t1=time.perf_counter()
    loss, _ = harness.forward_loss(model, crit, b, cfg)
    t2=time.perf_counter()
    opt.zero_grad(set_to_none=True)
    loss.backward()
    t3=time.perf_counter()
    grads = [p.grad for p in model.ecc.parameters() if p.grad is not None]
    torch._foreach_clamp_min_(grads, -1.0); torch._foreach_clamp_max_(grads, 1.0)
    opt.step()
    t4=time.perf_counter()
    if acc:
        T["graphs"]+=t1-t0; T["fwd"]+=t2-t1; T["bwd"]+=t3-t2; T["opt"]+=t4-t3
for _ in range(300): step()
torch.cuda.synchronize()
N=100
t=time.perf_counter()
for _ in range(N): step(True)
torch.cuda.synchronize()
tot=(time.perf_counter()-t)/N*1e3
print("step %.2f ms; host phases (ms):"%tot, {k: round(v/N*1e3,2) for k,v in T.items()}, "sum %.2f"%(sum(T.values())/N*1e3))
# same with a device sync before each step: the host never waits for a backlog, the phases are pure issue time
for k in T: T[k]=0.
for _ in range(N):
    torch.cuda.synchronize(); step(True)
print("synced-start host phases (ms):", {k: round(v/N*1e3,2) for k,v in T.items()}, "sum %.2f"%(sum(T.values())/N*1e3))
