"""Microbenchmark of the BatchNorm kernels at the C2 pyramid's (rows, channels):  python tools/bn_bench.py"""
import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); importlib.import_module("3d-wsis_amd")
import torch, wsis_native as _n
lib=_n.hip(); dev='cuda:0'
def timeit(f,n=50):
    for _ in range(5): f()
    torch.cuda.synchronize(); s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/n*1e3
for M,C in [(153685,32),(153685,64),(26819,64),(26819,128),(6500,96),(1600,128),(400,160),(1190,64)]:
    x=torch.randn(M,C,device=dev); dy=torch.randn(M,C,device=dev); y=torch.empty_like(x); dx=torch.empty_like(x)
    g=torch.rand(C,device=dev)+0.5; b=torch.randn(C,device=dev); mean=torch.empty(C,device=dev); var=torch.empty(C,device=dev)
    dg=torch.empty(C,device=dev); db=torch.empty(C,device=dev)
    wsb=lib.wsis_bn_workspace_bytes(M,C); ws=torch.empty(wsb,dtype=torch.uint8,device=dev); st=_n.stream_ptr()
    t_stats=timeit(lambda: lib.wsis_bn_stats(x.data_ptr(),M,C,mean.data_ptr(),var.data_ptr(),None,None,0.1,ws.data_ptr(),wsb,st))
    t_apply=timeit(lambda: lib.wsis_bn_apply(x.data_ptr(),mean.data_ptr(),var.data_ptr(),g.data_ptr(),b.data_ptr(),1e-4,1,y.data_ptr(),M,C,st))
    t_bwd_red=timeit(lambda: lib.wsis_bn_bwd(x.data_ptr(),dy.data_ptr(),mean.data_ptr(),var.data_ptr(),g.data_ptr(),b.data_ptr(),1e-4,1,1,None,dg.data_ptr(),db.data_ptr(),None,M,C,ws.data_ptr(),wsb,st))
    t_bwd=timeit(lambda: lib.wsis_bn_bwd(x.data_ptr(),dy.data_ptr(),mean.data_ptr(),var.data_ptr(),g.data_ptr(),b.data_ptr(),1e-4,1,1,dx.data_ptr(),dg.data_ptr(),db.data_ptr(),None,M,C,ws.data_ptr(),wsb,st))
    mb=M*C*4/1e6
    if os.environ.get("BN_COMPACT"):
        print(f"M={M:7d} C={C:4d}: apply {t_apply:5.1f}us  bwd apply {t_bwd-t_bwd_red:5.1f}us"); continue
    print(f"M={M:7d} C={C:4d} ({mb:6.1f} MB/tensor): stats {t_stats:6.1f}us ({mb/t_stats*1e-3*1e3:6.0f} GB/s)  apply {t_apply:6.1f}us ({2*mb/t_apply:6.0f} GB/s)  bwd reduce {t_bwd_red:6.1f}us ({2*mb/t_bwd_red:6.0f} GB/s)  bwd total {t_bwd:6.1f}us (apply part {t_bwd-t_bwd_red:6.1f}us, {3*mb/max(t_bwd-t_bwd_red,1e-3):6.0f} GB/s)")
