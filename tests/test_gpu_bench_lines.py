"""bench.py's documented one-GPU command lines end in ONE parseable JSON line with the contract's keys -- the default
form and the N = 1 point of the 1 -> 8 series (`--gpus 1 --scenes-per-gpu K`: several scenes per step, no process group).
Small rooms, a handful of steps: this checks the plumbing of the line, not its numbers."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ["--small", "--steps", "2", "--warmup", "1", "--setup-steps", "2", "--no-cpu-baseline", "--no-stages"]


def _line(extra):
    env = dict(os.environ)
    env.pop("WSIS_FORCE_DIST", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + COMMON + extra, env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("extra,spg", [([], 1), (["--gpus", "1", "--scenes-per-gpu", "2"], 2)], ids=["default", "spg2"])
def test_one_gpu_lines(extra, spg):
    d = _line(extra)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "scaling_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["unit"] == "scenes/s" and d["value"] > 0 and d["dtype"] == "f32"
    assert d["config"]["scenes_per_gpu"] == spg and "workload" in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and 0 < r["frac"] < 1
