"""Per-layer microbenchmark of the sparse-conv kernels on the C2 scene's real rulebooks (GPU box):
   python tools/conv_bench.py   -> us per launch, algorithmic GB/s and TFLOP/s per UNet layer."""
import importlib, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); importlib.import_module("3d-wsis_amd")
import numpy as np, torch, spconv, harness
from spconv import ops
dev='cuda:0'
sc=harness.make_scene(1); b=harness.collate([sc])
idx=b['voxel_locs'].int().to(dev).contiguous(); shape=[int(s) for s in b['spatial_shape']]
levels=[]
cur_idx, cur_shape = idx, shape
for l in range(5):
    rb=ops.build_subm_rulebook(cur_idx, cur_shape, [3]*3,[1]*3)
    P=int((rb.nbr>=0).sum()); M=cur_idx.shape[0]
    ent={'M':M,'P':P,'subm':rb}
    if l<4:
        rd=ops.build_down_rulebook(cur_idx, cur_shape,[2]*3,[2]*3,[0]*3)
        ent['down']=rd; cur_idx, cur_shape = rd.out_indices, rd.out_shape
    levels.append(ent)
    print(f"level {l}: M={M} P={P} nbrs/row={P/M:.2f}")
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/n*1e3
tot={'fwd':0,'dw':0}
planes=[32,64,96,128,160]
for l,ent in enumerate(levels):
    C=planes[l]; rb=ent['subm']; M,P=ent['M'],ent['P']
    for (cin,cout,cnt) in ([(C,C,8 if l<4 else 4)] + ([(2*C,C,1)] if l<4 else []) + ([(6,32,1)] if l==0 else [])):
        X=torch.randn(M,cin,device=dev); W=torch.randn(27,cin,cout,device=dev)*0.05; dY=torch.randn(M,cout,device=dev)
        t1=timeit(lambda: ops._conv(X,rb.nbr_p,rb.order,W,None,None,M))
        t0=timeit(lambda: ops._conv(X,rb.nbr,None,W,None,None,M))
        t2=timeit(lambda: ops._dw(X,rb.nbr_p,rb.order,dY,27,cin,cout))
        by=P*(cin+cout)*4+P*8; fl=2*P*cin*cout
        print(f"L{l} subm {cin:3d}->{cout:3d} x{cnt}: fwd {t1:7.1f}us (noorder {t0:7.1f}) {by/t1/1e3:7.1f} GB/s {fl/t1/1e6:6.2f} TF | dW {t2:7.1f}us {by/t2/1e3:7.1f} GB/s {fl/t2/1e6:6.2f} TF")
        tot['fwd']+=cnt*t1*2; tot['dw']+=cnt*t2
    if 'down' in ent:
        rd=ent['down']; Mo=rd.out_indices.shape[0]; cin,cout=C,planes[l+1]
        X=torch.randn(M,cin,device=dev); W=torch.randn(8,cin,cout,device=dev)*0.05; dY=torch.randn(Mo,cout,device=dev)
        t1=timeit(lambda: ops._conv(X,rd.nbr_p,rd.order,W,None,None,Mo))
        WT=torch.randn(8,cout,cin,device=dev)
        t3=timeit(lambda: ops._conv(dY,rd.nbr_up_p,rd.order_up,WT,None,None,M))
        t2=timeit(lambda: ops._dw(X,rd.nbr_p,rd.order,dY,8,cin,cout))
        by=M*(cin+cout)*4+M*8
        print(f"L{l} down {cin:3d}->{cout:3d}: fwd {t1:7.1f}us {by/t1/1e3:7.1f} GB/s | up/dIn {t3:7.1f}us {by/t3/1e3:7.1f} GB/s | dW {t2:7.1f}us")
        tot['fwd']+=2*(t1+t3); tot['dw']+=2*t2
print("estimated per-step conv: fwd+dIn %.2f ms, dW %.2f ms"%(tot['fwd']/1e3, tot['dw']/1e3))
