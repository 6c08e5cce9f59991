"""GPU time vs host issue time of the phases of one training step (events + perf_counter):
   python tools/step_phases.py"""
import importlib, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); importlib.import_module("3d-wsis_amd")
import numpy as np, torch, harness
dev=torch.device('cuda:0')
cfg=harness.default_cfg()
sc=harness.make_scene(1); bh=harness.collate([sc]); b=harness.to_device(bh,dev)
model,crit,opt=harness.build_model(cfg,dev)
import backbone_3D_WSIS
# monkeypatch forward phases with events
ev={}
def mark(name):
    e=torch.cuda.Event(enable_timing=True); e.record(); ev.setdefault(name,[]).append((e,time.perf_counter()))
orig_unet=model.unet.forward
def unet_fwd(x):
    mark('unet_start'); y=orig_unet(x); mark('unet_end'); return y
model.unet.forward=unet_fwd
import unet_native                      # default path: the whole UNet is one native call per pass
orig_run=unet_native.run_unet
def run_unet(net, t):
    mark('unet_start'); y=orig_run(net, t); mark('unet_end'); return y
unet_native.run_unet=run_unet
orig_ecc=model.ecc.forward
def ecc_fwd(x):
    mark('ecc_start'); y=orig_ecc(x); mark('ecc_end'); return y
model.ecc.forward=ecc_fwd
def step():
    mark('step_start')
    loss,ret=harness.forward_loss(model,crit,b,cfg)
    mark('fwd_end')
    opt.zero_grad(set_to_none=True)
    loss.backward()
    mark('bwd_end')
    for p in model.ecc.parameters():
        if p.grad is not None: p.grad.data.clamp_(-1,1)
    mark('clamp_end')
    opt.step()
    mark('opt_end')
for _ in range(3): step()
torch.cuda.synchronize(); ev.clear()
N=5
for _ in range(N): step()
torch.cuda.synchronize()
names=['step_start','unet_start','unet_end','ecc_start','ecc_end','fwd_end','bwd_end','clamp_end','opt_end']
for a,bn in zip(names[:-1],names[1:]):
    g=sum(ev[a][i][0].elapsed_time(ev[bn][i][0]) for i in range(N))/N
    c=sum(ev[bn][i][1]-ev[a][i][1] for i in range(N))/N*1e3
    print(f"{a:12s}->{bn:12s} gpu {g:7.2f} ms   cpu-issue {c:7.2f} ms")
tot=sum(ev['step_start'][i][0].elapsed_time(ev['opt_end'][i][0]) for i in range(N))/N
print("total gpu", tot)
