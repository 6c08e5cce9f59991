"""In-kernel s_memtime phase stamps of spconv_fwd_kernel per UNet level (diagnostic build of the kernel,
wsis_debug_spconv_diag; cdna_hip_programming.md section 7 'In-kernel stamps'):  python tools/conv_phase_stamps.py"""
import importlib, sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); importlib.import_module("3d-wsis_amd")
import numpy as np, torch, spconv, harness, wsis_native as _n
from spconv import ops
dev='cuda:0'
sc=harness.make_scene(1); b=harness.collate([sc])
idx=b['voxel_locs'].int().to(dev).contiguous(); shape=[int(s) for s in b['spatial_shape']]
lib=_n.hip()
f=lib.wsis_debug_spconv_diag; f.restype=ctypes.c_int32
f.argtypes=[ctypes.c_void_p]*6+[ctypes.c_int64,ctypes.c_int32,ctypes.c_int32,ctypes.c_int32,ctypes.c_int32,ctypes.c_void_p,ctypes.c_void_p]
planes=[32,64,96,128,160]
names=["prologue","wait+ldswrite","barrier1","issue prefetch","frag+mfma","barrier2","tail","steps"]
for lvl in range(5):
    rb=ops.build_subm_rulebook(idx, shape, [3]*3,[1]*3)
    M=idx.shape[0]; C=planes[lvl]
    X=torch.randn(M,C,device=dev); W=torch.randn(27,C,C,device=dev)*0.05
    out=torch.empty(M,C,device=dev)
    nt=(M+127)//128
    kz=int(lib.wsis_spconv_fwd_workspace_bytes(M,27,C,C)//(M*C*4)) if lib.wsis_spconv_fwd_workspace_bytes(M,27,C,C)>256 else 1
    part=torch.empty(max(kz,1)*M*C,device=dev)
    dbg=torch.zeros(nt*kz*8,dtype=torch.int64,device=dev)
    for _ in range(3):
        rc=f(X.data_ptr(),rb.nbr_p.data_ptr(),rb.order.data_ptr(),W.data_ptr(),out.data_ptr(),part.data_ptr(),M,27,C,C,kz,dbg.data_ptr(),None)
    torch.cuda.synchronize(); assert rc==0
    d=dbg.cpu().numpy().reshape(nt*kz,8).astype(np.float64)
    tot=d[:,:7].sum(1)
    print(f"L{lvl} M={M} C={C} tiles={nt} kz={kz} WGs={nt*kz}: mean cycles/WG {tot.mean():.0f} ({tot.mean()/2400:.1f} us @2.4GHz) steps/WG {d[:,7].mean():.1f}")
    print("    "+"  ".join(f"{n}:{d[:,i].mean():.0f}" for i,n in enumerate(names[:7])))
    print(f"    WG time p50 {np.percentile(tot,50):.0f} p90 {np.percentile(tot,90):.0f} max {tot.max():.0f} cycles; steps p50 {np.percentile(d[:,7],50):.0f} p90 {np.percentile(d[:,7],90):.0f} max {d[:,7].max():.0f}; cycles/step of the slowest 10%: {(tot[tot>=np.percentile(tot,90)]/np.maximum(d[tot>=np.percentile(tot,90),7],1)).mean():.0f}")
    if lvl<4:
        rd=ops.build_down_rulebook(idx, shape,[2]*3,[2]*3,[0]*3); idx, shape = rd.out_indices, rd.out_shape
