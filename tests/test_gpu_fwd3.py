"""The persistent conv kernel (spconv_fwd3_kernel) against the one-shot kernel on launches WITHOUT offset slabs, where it
also writes the epilogue reductions (by default only slab-split launches take it, so these paths need WSIS_FWD3=2; the
plan knobs are read once per process, hence the child processes)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_persistent_kernel_matches_the_one_shot_kernel_and_its_reductions(tmp_path):
    outs = {}
    for tag, v in (("fwd2", "0"), ("fwd3", "2")):
        f = str(tmp_path / (tag + ".npz"))
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_fwd3_check.py"), f],
                           env=dict(os.environ, WSIS_FWD3=v), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "OK" in r.stdout, r.stderr[-2000:]
        outs[tag] = np.load(f)
    a, b = outs["fwd2"], outs["fwd3"]
    # outputs: bit-identical (same offsets per wave, same order of additions)
    for k in ("subm_out", "din_subm_out", "din_strided_out"):
        assert not np.isnan(b[k]).any(), k
        assert np.array_equal(a[k], b[k]), k
    # forward statistics: every 32-row slice of the tile order, (sum, centred sum of squares)
    o = b["subm_out"][b["subm_order"]]
    n = (o.shape[0] // 32) * 32
    sl = o[:n].reshape(-1, 32, o.shape[1]).astype(np.float64)
    want_s, want_q = sl.sum(1), ((sl - sl.mean(1, keepdims=True)) ** 2).sum(1)
    for tag in ("fwd2", "fwd3"):
        st = outs[tag]["subm_stats"].astype(np.float64)
        assert not np.isnan(st).any(), tag
        assert np.abs(st[: n // 32, 0] - want_s).max() < 1e-4 and np.abs(st[: n // 32, 1] - want_q).max() < 1e-3, tag
    # BatchNorm-backward partials: totals (which rows form a slice is the kernel's business)
    for k in ("din_subm", "din_strided"):
        for tag in ("fwd2", "fwd3"):
            p = outs[tag][k + "_part"].astype(np.float64)
            assert not np.isnan(p).any(), (k, tag)          # every slice wrote its partials
            want = outs[tag][k + "_want"]
            sc = max(np.abs(want).max(), 1.0)
            assert np.abs(p[:, 0].sum(0) - want[0]).max() < 1e-5 * sc and np.abs(p[:, 1].sum(0) - want[1]).max() < 1e-5 * sc, (k, tag)
