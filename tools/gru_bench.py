"""GRUCellEx kernels alone (run under rocprofv3 --kernel-trace): forward, backward in a 7-slot sequence + its reduce"""
import importlib, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
importlib.import_module("3d-wsis_amd")
import torch
import wsis_native as _n

dev = torch.device("cuda", 0)
lib = _n.hip()
S = int(sys.argv[1]) if len(sys.argv) > 1 else 2289
torch.manual_seed(0)
x, h = torch.randn(S, 32, device=dev), torch.randn(S, 32, device=dev)
Wig, big = torch.randn(32, 32, device=dev) * 0.2, torch.randn(32, device=dev) * 0.1
Wih, Whh = torch.randn(96, 32, device=dev) * 0.2, torch.randn(96, 32, device=dev) * 0.2
bih, bhh = torch.randn(96, device=dev) * 0.1, torch.randn(96, device=dev) * 0.1
dhy = torch.randn(S, 32, device=dev)
gp = [Wig, big, Wih, Whh, bih, bhh]
st = _n.stream_ptr()
hy, dx, dh = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
dgp = [torch.empty_like(t) for t in gp]
R = 7
for cfg in os.environ.get("GRU_CFGS", "3").split(","):
    os.environ["WSIS_GRU_ROWS"] = cfg
    ws_bytes = (lib.wsis_gru_cell_workspace_bytes(S) - 256) * R + 256
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    for it in range(20):
        for i in range(R):
            _n.check(lib.wsis_gru_cell_fwd(_n.ptr(x), _n.ptr(h), *[_n.ptr(t) for t in gp], _n.ptr(hy), S, 32, st), "fwd")
        for i in range(R):
            _n.check(lib.wsis_gru_cell_bwd_seq(_n.ptr(x), _n.ptr(h), *[_n.ptr(t) for t in gp], _n.ptr(dhy), None, 32, _n.ptr(dx),
                                               _n.ptr(dh), *[_n.ptr(t) for t in dgp], S, 32, i, R, 1 if i == R - 1 else 0,
                                               _n.ptr(ws), ws_bytes, st), "bwd")
    torch.cuda.synchronize()
    print("cfg", cfg, "done")
