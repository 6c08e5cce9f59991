"""the dense products / contraction of one GRU step of the superpoint GNN: library kernels against hipBLASLt (torch)"""
import importlib, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
importlib.import_module("3d-wsis_amd")
import torch
import wsis_native as _n
import harness
from torch_scatter import SegmentCSR

dev = torch.device("cuda", 0)
lib = _n.hip()
S, E = 2289, 20054
torch.manual_seed(0)
hx = torch.randn(S, 32, device=dev)
W = torch.randn(32, 2080, device=dev) * 0.1
h = torch.randn(E, 64, device=dev)
src = torch.randint(0, S, (E,), device=dev)
dst = torch.randint(0, S, (E,), device=dev)
csr_src, csr_dst = SegmentCSR(src, S), SegmentCSR(dst, S)
st = _n.stream_ptr()


def t(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


U = torch.empty(S, 2080, device=dev)
print("U = hx @ W     torch %.1f us   own %.1f us" % (
    t(lambda: torch.mm(hx, W, out=U)), t(lambda: lib.wsis_ecc_u_fwd(_n.ptr(hx), _n.ptr(W), _n.ptr(U), S, st))))
U2 = torch.empty_like(U)
lib.wsis_ecc_u_fwd(_n.ptr(hx), _n.ptr(W), _n.ptr(U2), S, st)
print("   max diff", float((U2 - hx @ W).abs().max()))
d_inp = torch.randn(S, 32, device=dev)
d_m = torch.empty(E, 32, device=dev)
dUo, dh = torch.empty(S, 2080, device=dev), torch.empty(E, 64, device=dev)
dUo2, dh2 = torch.empty(S, 2080, device=dev), torch.empty(E, 64, device=dev)


def two():
    lib.wsis_segment_reduce_bwd(_n.ptr(d_inp), _n.ptr(csr_src.index), _n.ptr(csr_src.offsets), None, _n.ptr(d_m), E, S, 32, 1, st)
    lib.wsis_ecc_contract_bwd_acc(_n.ptr(h), _n.ptr(U), _n.ptr(d_m), _n.ptr(csr_dst.perm), _n.ptr(csr_dst.offsets),
                                  _n.ptr(dUo), _n.ptr(dh), S, E, 0, st)


def one():
    lib.wsis_ecc_contract_bwd_mean(_n.ptr(h), _n.ptr(U), _n.ptr(d_inp), _n.ptr(csr_src.index), _n.ptr(csr_src.offsets),
                                   _n.ptr(csr_dst.perm), _n.ptr(csr_dst.offsets), _n.ptr(dUo2), _n.ptr(dh2), S, E, 0, st)


print("contract bwd    two launches %.1f us   folded %.1f us" % (t(two), t(one)))
two(); one()
print("   equal", bool(torch.equal(dUo, dUo2)), bool(torch.equal(dh, dh2)))
m = torch.empty(E, 32, device=dev)
print("contract fwd %.1f us" % t(lambda: lib.wsis_ecc_contract_fwd(_n.ptr(h), _n.ptr(U), _n.ptr(csr_dst.perm), _n.ptr(csr_dst.offsets), _n.ptr(m), S, E, st)))
