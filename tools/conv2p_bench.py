"""One-shot against persistent form of the wave-autonomous conv kernel per UNet level (C2 scene, real tables):
   python tools/conv2p_bench.py          us per launch with WSIS_FWD2P=0 / 1 (and WSIS_FWD2P_MIN sweeps in argv)"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); importlib.import_module("3d-wsis_amd")
import torch, harness
from spconv import ops

dev = "cuda:0"


def timeit(f, n=40):
    for _ in range(5): f()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3


def main():
    ns = int(os.environ.get("CONV2_SCENES", "1"))
    b = harness.collate([harness.bench_scene(1 + i) for i in range(ns)])
    idx = b["voxel_locs"].int().to(dev).contiguous(); shape = [int(s) for s in b["spatial_shape"]]
    planes = [32, 64, 96, 128, 160]
    g = torch.Generator(device=dev).manual_seed(0)
    for l in range(3):
        rb = ops.build_subm_rulebook(idx, shape, [3] * 3, [1] * 3)
        rd = ops.build_down_rulebook(idx, shape, [2] * 3, [2] * 3, [0] * 3)
        M, C = idx.shape[0], planes[l]
        for (cin, cout, tab, Mo, K, name) in ((C, C, rb, M, 27, "subm"), (2 * C, C, rb, M, 27, "subm2"),
                                              (C, planes[l + 1], rd, rd.out_indices.shape[0], 8, "down")):
            X = torch.randn(M, cin, device=dev, generator=g)
            W = torch.randn(K, cin, cout, device=dev, generator=g) * 0.05
            WT = ops._weight_t(W, 0)
            st = torch.empty(((Mo + 31) // 32, 2, cout), device=dev)
            row = f"L{l} {name:5s} rows {Mo:6d} {cin:3d}->{cout:3d} items {(Mo + 31) // 32 * (cout // 32):5d}:"
            for v in ("0", "2"):
                os.environ["WSIS_FWD2P"] = v
                t = timeit(lambda: ops._conv_t(X, tab.nbr_p, tab.order, WT, 0, None, None, Mo, stats=st))
                row += f"  FWD2P={v} {t:6.1f} us"
            print(row, flush=True)
        idx, shape = rd.out_indices, rd.out_shape


if __name__ == "__main__":
    main()
