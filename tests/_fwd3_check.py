"""Helper of tests/test_gpu_fwd3.py (run as a child process: the launch plan knobs are read once per process).
Writes <out>.npz with the outputs / partials of a fixed set of convolution launches."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); importlib.import_module("3d-wsis_amd")
import numpy as np, torch, harness, wsis_native as _n
from spconv import ops

dev = "cuda:0"
lib = _n.hip()
b = harness.collate([harness.make_scene(21)])
idx = b["voxel_locs"].int().to(dev).contiguous(); shape = [int(s) for s in b["spatial_shape"]]
rd0 = ops.build_down_rulebook(idx, shape, [2] * 3, [2] * 3, [0] * 3)
idx1, shape1 = rd0.out_indices, rd0.out_shape
rb1 = ops.build_subm_rulebook(idx1, shape1, [3] * 3, [1] * 3)
rd1 = ops.build_down_rulebook(idx1, shape1, [2] * 3, [2] * 3, [0] * 3)
M1, M2 = idx1.shape[0], rd1.out_indices.shape[0]
g = torch.Generator(device=dev).manual_seed(0)
res = {}


def bn_case(name, Xg, nbr, order, W, flip, rows, K, Cin, Cout):
    x = torch.randn(rows, Cout, device=dev, generator=g); gamma = torch.randn(Cout, device=dev, generator=g)
    beta = torch.randn(Cout, device=dev, generator=g) * 0.3
    mean, var = x.mean(0).contiguous(), x.var(0, unbiased=False).contiguous()
    n_part = (rows + 31) // 32
    part = torch.full((n_part, 2, Cout), float("nan"), device=dev); out = torch.full((rows, Cout), float("nan"), device=dev)
    ws = torch.empty(max(lib.wsis_spconv_fwd_t_workspace_bytes(rows, K, Cin, Cout), 256), dtype=torch.uint8, device=dev)
    _n.check(lib.wsis_spconv_fwd_t_bn(_n.ptr(Xg), _n.ptr(nbr), _n.ptr(order), _n.ptr(W), flip, _n.ptr(out), _n.ptr(part), _n.ptr(x),
                                      _n.ptr(mean), _n.ptr(var), _n.ptr(gamma), _n.ptr(beta), 1e-4, 1, Xg.shape[0], rows, K, Cin, Cout,
                                      _n.ptr(ws), ws.numel(), _n.ptr(_n.sync_block()), _n.stream_ptr()), name)
    xh = (x - mean) * torch.rsqrt(var + 1e-4)
    dz = torch.where(xh * gamma + beta <= 0, torch.zeros_like(out), out)
    res[name + "_out"] = out.cpu().numpy(); res[name + "_part"] = part.cpu().numpy()
    res[name + "_want"] = torch.stack([dz.double().sum(0), (dz * xh).double().sum(0)]).cpu().numpy()


# SubM 64 -> 64 at level 1 (no slabs, 2.2 slices per resident workgroup): forward with residual + statistics
X = torch.randn(M1, 64, device=dev, generator=g); W = torch.randn(27, 64, 64, device=dev, generator=g) * 0.05
r = torch.randn(M1, 64, device=dev, generator=g)
st = torch.full(((M1 + 31) // 32, 2, 64), float("nan"), device=dev)
out = ops._conv_t(X, rb1.nbr_p, rb1.order, ops._weight_t(W, 0), 0, None, r, M1, stats=st)
res["subm_out"] = out.cpu().numpy(); res["subm_stats"] = st.cpu().numpy(); res["subm_order"] = rb1.order.cpu().numpy()
# dIn passes with the BatchNorm-backward epilogue: SubM (K = 27) and the strided conv's dIn (K = 8: waves without any
# pair in their first slice)
bn_case("din_subm", torch.randn(M1, 64, device=dev, generator=g), rb1.nbr_p, rb1.order, W, 1, M1, 27, 64, 64)
bn_case("din_strided", torch.randn(M2, 96, device=dev, generator=g), rd1.nbr_up_p, rd1.order_up,
        torch.randn(8, 64, 96, device=dev, generator=g) * 0.1, 0, M1, 8, 96, 64)
torch.cuda.synchronize()
np.savez(sys.argv[1], **res)
print("OK")
