"""CPU: the workgroup -> work item map of the forward / dIn convolution launch (csrc/spconv2.hip: item_of) is a
bijection for every launch shape, keeps the blocks / slabs of a slice together in the weight order, and -- dealt to
CU i % band -- balances a convex weight curve better than the launch order it replaces."""
import ctypes
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib():
    lib = ctypes.CDLL(os.path.join(ROOT, "3d-wsis_amd", "libwsis_hip.so"))
    fn = lib.wsis_debug_item_of
    fn.restype = ctypes.c_int32
    fn.argtypes = [ctypes.c_int32] * 5 + [ctypes.c_void_p]
    return fn


def _items(fn, gx, gy, gz, band):
    out = np.zeros(3, dtype=np.int32)
    res = np.zeros((gx * gy * gz, 3), dtype=np.int64)
    for i in range(gx * gy * gz):
        assert fn(i, gx, gy, gz, band, out.ctypes.data) == 0
        res[i] = out
    return res


@pytest.mark.parametrize("gx,gy,gz", [(1, 1, 1), (7, 1, 1), (206, 3, 1), (48, 4, 1), (11, 5, 8), (839, 2, 1), (300, 1, 1),
                                      (256, 1, 1), (257, 1, 1), (512, 2, 1), (341, 3, 1), (1025, 1, 1), (4803, 1, 1)])
@pytest.mark.parametrize("band", [256, 8, 0x7fffffff, -64, -5])      # < 0: runs of -band items per XCD
def test_item_of_is_a_bijection(gx, gy, gz, band):
    it = _items(_lib(), gx, gy, gz, band)
    assert (it[:, 0] >= 0).all() and (it[:, 0] < gx).all()
    assert (it[:, 1] >= 0).all() and (it[:, 1] < gy).all()
    assert (it[:, 2] >= 0).all() and (it[:, 2] < gz).all()
    lin = (it[:, 0] * gy + it[:, 1]) * gz + it[:, 2]
    assert sorted(lin.tolist()) == list(range(gx * gy * gz))
    if band == 0x7fffffff:        # no deal: the items in weight order
        assert lin.tolist() == list(range(gx * gy * gz))


def test_xcd_runs_keep_neighbouring_items_on_one_xcd():
    """band = -C: workgroup i runs on XCD i % 8; the items an XCD computes form runs of C consecutive items of the weight
    order (full super-blocks of 8 C; the remainder is dealt plainly), and every XCD gets the same number of full runs"""
    fn = _lib()
    gx, C = 4803, 64
    it = _items(fn, gx, 1, 1, -C)[:, 0]
    nb = gx // (8 * C)
    for x in range(8):
        mine = it[x::8]
        full = np.sort(mine[mine < nb * 8 * C])
        assert len(full) == nb * C
        runs = full.reshape(nb, C)
        assert (np.diff(runs, axis=1) == 1).all()                  # each run: C consecutive items
        assert (runs[:, 0] // C % 8 == x).all()                    # runs x, x + 8, x + 16, ...
    assert (it[nb * 8 * C:] == np.arange(nb * 8 * C, gx)).all()    # remainder in launch order


def test_item_of_rejects_bad_arguments():
    fn = _lib()
    out = np.zeros(3, dtype=np.int32)
    assert fn(5, 1, 1, 1, 256, out.ctypes.data) == -1
    assert fn(0, 0, 1, 1, 256, out.ctypes.data) == -1
    assert fn(0, 1, 1, 1, 256, None) == -1
    assert fn(0, 1, 1, 1, 0, out.ctypes.data) == -1


@pytest.mark.parametrize("gx,gy", [(206, 3), (48, 4), (300, 2), (120, 5)])
def test_deal_balances_a_convex_weight_curve(gx, gy):
    """slices come heaviest first (weights 27 .. 9 steps, steep at the start like a scene's): with workgroup i on CU
    i % 256 the busiest CU of the deal must carry less than with the plain (slice, block) launch order, and stay within
    40 % of the mean"""
    fn = _lib()
    w = np.concatenate([np.linspace(27, 12, gx // 4), np.linspace(12, 9, gx - gx // 4)]).round()
    it = _items(fn, gx, gy, 1, 256)
    n = gx * gy
    cu = np.arange(n) % 256
    dealt = np.bincount(cu, weights=w[it[:, 0]], minlength=256)
    plain = np.bincount(cu, weights=w[np.arange(n) % gx], minlength=256)      # round 4: workgroup (bx, by), x fastest
    assert dealt.max() <= plain.max()
    assert dealt.max() <= 1.4 * dealt.sum() / 256 + 27
