import importlib, sys
sys.path.insert(0, "."); importlib.import_module("3d-wsis_amd")
import torch, wsis_native as _n
lib=_n.hip(); dev="cuda"
g=torch.Generator(device=dev).manual_seed(0)
for (M,C,mu,sd) in [(153685,32,0.3,1.5),(26819,64,-2.0,0.7),(2289,64,0.1,1.0),(20054,128,5.0,0.2),(700,24,50.0,0.05),(6572,96,0.0,1.0)]:
    x=torch.randn(M,C,device=dev,generator=g)*sd+mu
    outs=[]
    for rep in range(3):
        mean=torch.empty(C,device=dev); var=torch.empty(C,device=dev)
        wsb=lib.wsis_bn_workspace_bytes(M,C); ws=torch.empty(wsb,dtype=torch.uint8,device=dev)
        _n.check(lib.wsis_bn_stats(_n.ptr(x),M,C,_n.ptr(mean),_n.ptr(var),None,None,0.1,_n.ptr(ws),wsb,_n.stream_ptr()),"bn")
        outs.append((mean.clone(),var.clone()))
    xd=x.double(); m64=xd.mean(0); v64=xd.var(0,unbiased=False)
    em=float(((outs[0][0].double()-m64).abs()/ (v64.sqrt())).max()); ev=float(((outs[0][1].double()-v64).abs()/v64).max())
    det=all(torch.equal(outs[0][0],o[0]) and torch.equal(outs[0][1],o[1]) for o in outs)
    print(M,C,"mean err/sigma %.2e var rel err %.2e deterministic %s"%(em,ev,det))
