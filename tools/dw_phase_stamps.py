"""In-kernel s_memtime phase stamps of spconv_dw_kernel per UNet level (diagnostic build, wsis_debug_dw_diag):
   python tools/dw_phase_stamps.py"""
import importlib, sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); importlib.import_module("3d-wsis_amd")
import numpy as np, torch, spconv, harness, wsis_native as _n
from spconv import ops
dev='cuda:0'
sc=harness.make_scene(1); b=harness.collate([sc])
idx=b['voxel_locs'].int().to(dev).contiguous(); shape=[int(s) for s in b['spatial_shape']]
lib=_n.hip()
f=lib.wsis_debug_dw_diag; f.restype=ctypes.c_int32
f.argtypes=[ctypes.c_void_p]*5+[ctypes.c_int64,ctypes.c_int64,ctypes.c_int32,ctypes.c_int32,ctypes.c_int32,ctypes.c_void_p,ctypes.c_void_p,ctypes.c_void_p]
names=["prologue","scan wait","compaction","issue loads","wait+mfma","epilogue","blocks","iterations"]
for lvl,C in enumerate([32,64]):
    rb=ops.build_subm_rulebook(idx, shape, [3]*3,[1]*3)
    M=idx.shape[0]
    X=torch.randn(M,C,device=dev); dY=torch.randn(M,C,device=dev)
    n=ctypes.c_int32(0)
    assert f(None,None,None,None,None,M,M,27,C,C,None,ctypes.addressof(n),None)==0
    items=n.value; ncib=(C+31)//32
    part=torch.empty(items*ncib*32*C,device=dev)
    dbg=torch.zeros(items*ncib*4*8,dtype=torch.int64,device=dev)
    for _ in range(3):
        rc=f(X.data_ptr(),rb.nbr_p.data_ptr(),rb.order.data_ptr(),dY.data_ptr(),part.data_ptr(),M,M,27,C,C,dbg.data_ptr(),ctypes.addressof(n),None)
    torch.cuda.synchronize(); assert rc==0
    d=dbg.cpu().numpy().reshape(items*ncib*4,8).astype(np.float64)
    tot=d[:,:6].sum(1)
    print(f"L{lvl} M={M} C={C} items={items} waves={len(d)}: wave time mean {tot.mean():.0f} max {tot.max():.0f} p90 {np.percentile(tot,90):.0f} ticks; blocks/wave {d[:,6].mean():.1f} iterations/wave {d[:,7].mean():.1f}")
    print("    mean per wave: "+"  ".join(f"{nm}:{d[:,i].mean():.0f}" for i,nm in enumerate(names[:6])))
    heavy=d[tot>=np.percentile(tot,90)]
    print("    slowest 10%:   "+"  ".join(f"{nm}:{heavy[:,i].mean():.0f}" for i,nm in enumerate(names)))
    rd=ops.build_down_rulebook(idx, shape,[2]*3,[2]*3,[0]*3); idx, shape = rd.out_indices, rd.out_shape
