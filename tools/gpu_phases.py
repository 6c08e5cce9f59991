"""Where the main stream spends a steady-state training step: events at the phase boundaries of the bench's own step
(graph builds, UNet forward, heads + loss, backward of heads / GNN, UNet backward, clamp + AdamW), GPU time between
them, and the host's lead over the GPU at each boundary (host time stamp of the record vs. the event's completion,
both on one clock through a reference event).  `python tools/gpu_phases.py` on the GPU box."""
import importlib, sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); importlib.import_module("3d-wsis_amd")
import torch, harness
import unet_native
dev = torch.device('cuda:0')
cfg = harness.default_cfg()
model, crit, opt = harness.build_model(cfg, dev)
b = harness.to_device(harness.collate([harness.bench_scene(1)]), dev)
ON = [False]
EV = {}
def mark(name):
    if not ON[0]:
        return
    e = torch.cuda.Event(enable_timing=True); e.record()
    EV.setdefault(name, []).append((e, time.perf_counter()))
orig_fwd, orig_bwd = unet_native.UNetFunction.forward, unet_native.UNetFunction.backward
def fwd(ctx, *a):
    mark('unet_f0'); r = orig_fwd(ctx, *a); mark('unet_f1'); return r
def bwd(ctx, *a):
    mark('unet_b0'); r = orig_bwd(ctx, *a); mark('unet_b1'); return r
unet_native.UNetFunction.forward = staticmethod(fwd)
unet_native.UNetFunction.backward = staticmethod(bwd)
def step():
    mark('start')
    harness.build_batch_graphs(b)
    mark('graphs')
    loss, _ = harness.forward_loss(model, crit, b, cfg)
    mark('fwd_end')
    opt.zero_grad(set_to_none=True)
    loss.backward()
    mark('bwd_end')
    grads = [p.grad for p in model.ecc.parameters() if p.grad is not None]
    torch._foreach_clamp_min_(grads, -1.0); torch._foreach_clamp_max_(grads, 1.0)
    opt.step()
    mark('opt_end')
for _ in range(int(os.environ.get("SETUP", "300"))): step()
torch.cuda.synchronize()
N = 40
t0 = time.perf_counter()
for _ in range(N): step()
torch.cuda.synchronize()
plain = (time.perf_counter() - t0) / N * 1e3
ON[0] = True
ref = torch.cuda.Event(enable_timing=True); ref.record(); torch.cuda.synchronize(); ref_host = time.perf_counter()
t0 = time.perf_counter()
for _ in range(N): step()
torch.cuda.synchronize()
marked = (time.perf_counter() - t0) / N * 1e3
print(f"step {plain:.2f} ms plain, {marked:.2f} ms with the events")
names = ['start', 'graphs', 'unet_f0', 'unet_f1', 'fwd_end', 'unet_b0', 'unet_b1', 'bwd_end', 'opt_end']
skip = 5
def gpu_t(name, i):      # completion time of the event on the host clock (ms since ref)
    return ref.elapsed_time(EV[name][i][0])
def host_t(name, i):
    return (EV[name][i][1] - ref_host) * 1e3
print("phase                      gpu ms   host ms   host lead at the end of the phase (ms; <= 0: the GPU waited for the host)")
for a, c in zip(names[:-1], names[1:]):
    g = sum(gpu_t(c, i) - gpu_t(a, i) for i in range(skip, N)) / (N - skip)
    h = sum(host_t(c, i) - host_t(a, i) for i in range(skip, N)) / (N - skip)
    lead = sum(gpu_t(c, i) - host_t(c, i) for i in range(skip, N)) / (N - skip)
    print(f"{a:8s} -> {c:8s}   {g:8.2f}  {h:8.2f}   {lead:8.2f}")
g = sum(gpu_t('start', i + 1) - gpu_t('opt_end', i) for i in range(skip, N - 1)) / (N - skip - 1)
print(f"opt_end  -> next start {g:8.2f}")
