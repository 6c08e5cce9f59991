"""The 6-channel input convolution on the matrix cores (csrc/spconv_in.hip, backbone_3D_WSIS.py:43) against the generic
kernel it replaces (WSIS_IN_CONV=0) and against an fp64 gather-GEMM over the unpacked table."""
import pytest
import torch

import harness
from spconv import ops

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("room,bias,res", [((1.3, 1.1, 0.9), False, False), ((2.3, 1.9, 1.5), True, True)])
def test_input_conv_mfma_matches_generic_kernel_and_fp64(monkeypatch, room, bias, res):
    b = harness.collate([harness.make_scene(9, room=room, n_box=2)])
    idx = b["voxel_locs"].int().to(DEV).contiguous()
    shape = [int(s) for s in b["spatial_shape"]]
    rb = ops.build_subm_rulebook(idx, shape, [3] * 3, [1] * 3)
    M = idx.shape[0]
    assert M % 32 != 0
    g = torch.Generator(device=DEV).manual_seed(M)
    X = torch.randn(M, 6, device=DEV, generator=g)
    W = torch.randn(27, 6, 32, device=DEV, generator=g) * 0.2
    bv = torch.randn(32, device=DEV, generator=g) if bias else None
    rs = torch.randn(M, 32, device=DEV, generator=g) if res else None
    outs = []
    for v in ("0", "1"):
        monkeypatch.setenv("WSIS_IN_CONV", v)
        outs.append(ops._conv(X, rb.nbr_p, rb.order, W, bv, rs, M))
    torch.cuda.synchronize()
    want = torch.zeros(M, 32, dtype=torch.float64, device=DEV)
    for k in range(27):
        sel = rb.nbr[k] >= 0
        want[sel] += X.double()[rb.nbr[k][sel].long()] @ W[k].double()
    if bias:
        want += bv.double()
    if res:
        want += rs.double()
    scale = float(want.abs().max())
    for o in outs:
        assert float((o.double() - want).abs().max()) <= 2e-6 * scale
    assert float((outs[0] - outs[1]).abs().max()) <= 2e-6 * scale


@pytest.mark.parametrize("room", [(1.3, 1.1, 0.9), (2.3, 1.9, 1.5), (4.6, 3.6, 2.2)])
def test_input_conv_weight_gradient_mfma_matches_generic_kernel_and_fp64(monkeypatch, room):
    """dW [27, 6, 32] of the input convolution on the im2col kernel (spconv_in_dw_kernel) against the generic kernel
    (WSIS_IN_CONV=0) and an fp64 gather-GEMM; identical run to run"""
    b = harness.collate([harness.make_scene(11, room=room, n_box=2)])
    idx = b["voxel_locs"].int().to(DEV).contiguous()
    shape = [int(s) for s in b["spatial_shape"]]
    rb = ops.build_subm_rulebook(idx, shape, [3] * 3, [1] * 3)
    M = idx.shape[0]
    g = torch.Generator(device=DEV).manual_seed(M)
    X = torch.randn(M, 6, device=DEV, generator=g)
    dY = torch.randn(M, 32, device=DEV, generator=g)
    outs = []
    for v in ("0", "1", "1"):
        monkeypatch.setenv("WSIS_IN_CONV", v)
        outs.append(ops._dw(X, rb.nbr_p, rb.order, dY, 27, 6, 32))
    torch.cuda.synchronize()
    want = torch.zeros(27, 6, 32, dtype=torch.float64, device=DEV)
    for k in range(27):
        sel = rb.nbr[k] >= 0
        want[k] = X.double()[rb.nbr[k][sel].long()].t() @ dY.double()[sel]
    scale = float(want.abs().max())
    for o in outs:
        assert float((o.double() - want).abs().max()) <= 1e-5 * scale
    assert torch.equal(outs[1], outs[2])
