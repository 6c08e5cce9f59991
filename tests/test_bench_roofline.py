"""CPU: the roofline bookkeeping of bench.py -- the per-launch median over the instrumented steps (one stalled launch of
one step must not move a level's figure) and the per-level grouping with its two ceilings (HBM time on the algorithmic
bytes of SURVEY 8d, fp32-MFMA time on the pair flops)."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("wsis_bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def _entry(steps, stall=None):
    # two products per step: (rows, Cin, Cout, pairs, bytes, flops, ms main kernel, ms with the finishing launch)
    base = [(1000, 32, 32, 9000, 9000 * 64 * 4 + 9000 * 8, 2 * 9000 * 32 * 32, 0.050, 0.050),
            (100, 64, 64, 2000, 2000 * 128 * 4 + 2000 * 8, 2 * 2000 * 64 * 64, 0.020, 0.024)]
    per = []
    for s in range(steps):
        for i, r in enumerate(base):
            extra = stall[1] if stall and stall[0] == (s, i) else 0.0
            per.append(r[:6] + (r[6] + extra, r[7] + extra))
    return {"launches": len(per), "bytes": sum(r[4] for r in per), "flops": sum(r[5] for r in per),
            "ms_main": sum(r[6] for r in per), "ms": sum(r[7] for r in per), "per_launch": per}


def test_median_over_steps_removes_one_stalled_launch():
    clean = _entry(3)
    stalled = _entry(3, stall=((1, 1), 0.440))
    assert stalled["ms"] > clean["ms"] + 0.4
    got = bench.median_over_steps(stalled, 3)
    assert abs(got["ms"] - clean["ms"]) < 1e-12 and abs(got["ms_main"] - clean["ms_main"]) < 1e-12
    assert got["launches"] == clean["launches"] and got["bytes"] == clean["bytes"]
    assert len(got["per_launch"]) == len(clean["per_launch"])
    # fewer than three steps, or steps that do not issue the same products: left alone
    two = _entry(2, stall=((0, 0), 0.1))
    assert bench.median_over_steps(two, 2) is two
    odd = _entry(3)
    odd["per_launch"][3] = (7,) + odd["per_launch"][3][1:]
    assert bench.median_over_steps(odd, 3) is odd


def test_per_level_groups_by_output_rows_and_prices_both_ceilings():
    e = _entry(3)
    lv = bench.per_level(e["per_launch"], 3)
    assert [l["rows"] for l in lv] == [1000, 100] and [l["launches"] for l in lv] == [1, 1]
    assert abs(lv[0]["us"] - 50.0) < 1e-6 and abs(lv[1]["us"] - 24.0) < 1e-6 and abs(lv[1]["us_main_kernel_only"] - 20.0) < 1e-6
    b0, f0 = e["per_launch"][0][4], e["per_launch"][0][5]
    want = max(b0 / (bench.HBM_PEAK_GBS * 1e9), f0 / (bench.MFMA_FP32_PEAK_TFLOPS * 1e12)) * 1e6
    assert abs(lv[0]["ceiling_us"] - round(want, 1)) < 0.11
    assert abs(lv[0]["frac"] - round(b0 / 50e-6 / 1e9 / bench.HBM_PEAK_GBS, 4)) < 1e-4
