"""Two real ranks through the product path (BASELINE configs[4] "C5"): two child processes, both on cuda:0,
process group on gloo, each stepping the real ``Network`` on its own scene through ``harness.train_step`` with
``wsis_parallel.GradSync`` -- plan agreement, flat in-place all-reduce of the native UNet's gradient buffer, tail
packing of the other parameters, and (after the plan is frozen) the early exchange of the first half of the buffer
from inside the backward pass.  Replaces the reference's never-initialised DDP wrapper, train_scannetv2.py:734-738.
The parent only launches the children and compares what they saved."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "_two_rank_worker.py")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run_two(tmp_path, scenario, timeout=600):
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ)
        env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), WSIS_DIST_BACKEND="gloo", WSIS_DIST_TIMEOUT="240",
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, WORKER, str(tmp_path), scenario], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    try:
        for p in procs:
            out, _ = p.communicate(timeout=timeout)
            outs.append(out)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()            # exactly the children started above
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {rank} failed (a hang shows up here as the {timeout} s limit):\n{out[-4000:]}"
    return [torch.load(os.path.join(tmp_path, f"{scenario}{r}.pt"), weights_only=False) for r in range(2)]


@pytest.mark.timeout(900)
def test_two_ranks_step_the_real_network_with_flat_and_early_exchange(tmp_path):
    r0, r1 = _run_two(tmp_path, "steady")
    i0, i1 = r0["info"], r1["info"]
    assert i0["voxels"] != i1["voxels"], "the ranks must see different scenes"
    # (c) the early exchange ran from inside the backward pass once the plan was frozen, on both ranks, last step included
    assert i0["early_count"] > 0 and i0["early_count"] == i1["early_count"]
    assert i0["early_last_step"] == 1 and i1["early_last_step"] == 1
    # the flat zero-copy path carried every UNet parameter and the tail the others
    assert i0["flat_params"] >= i0["n_unet_params"] > 100 and i0["agreed"] == i1["agreed"]
    assert i0["agreed"][0] > 0 and i0["agreed"][1] > 0
    ecc = set(i0["ecc_names"])
    assert set(r0["synced"]) == set(r1["synced"]) == set(r0["local"])
    worst = 0.0
    for n in r0["synced"]:
        a, b = r0["synced"][n], r1["synced"][n]
        assert torch.equal(a, b), f"(a) ranks differ after the exchange: {n}"
        want = (r0["local"][n].double() + r1["local"][n].double()) / 2       # (b) mean of the single-process gradients
        if n in ecc:
            want = want.clamp(-1.0, 1.0)                                     # train_scannetv2.py:247-249
        scale = float(want.abs().max()) + 1e-30
        err = float((a.double() - want).abs().max()) / scale
        worst = max(worst, err)
        assert err <= 2e-6, f"(b) {n}: {err}"
    # identical averaged gradients + identical update rule -> identical weights on both ranks
    for n in r0["weights"]:
        assert torch.equal(r0["weights"][n], r1["weights"][n]), n
    assert all(abs(x - y) > 0 for x, y in zip(i0["losses"][:1], i1["losses"][:1])), "different scenes, different losses"


@pytest.mark.timeout(900)
def test_a_rank_without_the_flat_layout_puts_every_rank_on_the_bucket_path(tmp_path):
    """(d) rank 1 breaks its flat gradient layout and drops one gradient during plan agreement: nobody hangs, every
    rank takes the bucket path for that step and the averaged gradients are still right"""
    r0, r1 = _run_two(tmp_path, "broken")
    assert r0["info"]["agreed"] == r1["info"]["agreed"] == [0, 0]
    assert r0["info"]["flat_params"] == r1["info"]["flat_params"] == 0
    missing = [n for n in r0["local"] if n.startswith("linear.3")][0]
    for n in r0["synced"]:
        assert torch.equal(r0["synced"][n], r1["synced"][n]), n
        l1 = torch.zeros_like(r0["local"][n]) if n == missing else r1["local"][n]
        want = (r0["local"][n].double() + l1.double()) / 2
        scale = float(want.abs().max()) + 1e-30
        assert float((r0["synced"][n].double() - want).abs().max()) / scale <= 2e-6, n


@pytest.mark.timeout(900)
def test_batchnorm_statistics_shared_across_two_ranks(tmp_path):
    """wsis_parallel.convert_sync_batchnorm (the reference converts to SyncBatchNorm when num_gpus > 1,
    train_scannetv2.py:734-736): a layer on a tensor split 700 / 1300 over the ranks equals fp64 BatchNorm1d + ReLU over
    the whole tensor -- outputs, input gradients, running statistics; the ranks' dgamma / dbeta add up to the
    full-batch ones --, and the converted Network keeps identical weights and running statistics on both ranks."""
    r = _run_two(tmp_path, "syncbn")
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(2000, 24, generator=g) * 2.0 + 0.5).double().requires_grad_(True)
    w = torch.randn(2000, 24, generator=g).double()
    gamma, beta = (torch.rand(24, generator=g) + 0.5).double(), (torch.randn(24, generator=g) * 0.1).double()
    bn = torch.nn.BatchNorm1d(24, eps=1e-4, momentum=0.1).double()
    with torch.no_grad():
        bn.weight.copy_(gamma)
        bn.bias.copy_(beta)
    y = torch.relu(bn(x))
    (y * w).sum().backward()
    for rr in r:
        a, b = rr["layer"]["rows"]
        assert torch.allclose(rr["layer"]["y"].double(), y[a:b].detach(), rtol=1e-5, atol=1e-5)
        assert torch.allclose(rr["layer"]["dx"].double(), x.grad[a:b], rtol=1e-4, atol=1e-5)
        assert torch.allclose(rr["layer"]["running_mean"].double(), bn.running_mean, rtol=1e-5, atol=1e-6)
        assert torch.allclose(rr["layer"]["running_var"].double(), bn.running_var, rtol=1e-5, atol=1e-6)
    # |mean| = 1000 sigma: E[x^2] - mean^2 over fp32-rounded local means would be off by ~10 % here
    xb = (torch.randn(2000, 24, generator=g) * 0.05 + 50.0)
    bnb = torch.nn.BatchNorm1d(24, eps=1e-6, momentum=1.0).double()
    yb = bnb(xb.double())
    for rr in r:
        a, b = rr["layer"]["rows"]
        assert torch.allclose(rr["layer"]["big_running_var"].double(), bnb.running_var, rtol=2e-3, atol=0), "cancellation"
        assert torch.allclose(rr["layer"]["big_y"].double(), yb[a:b].detach(), rtol=2e-3, atol=2e-3)
    for k, want in (("dgamma", bn.weight.grad), ("dbeta", bn.bias.grad)):
        got = r[0]["layer"][k].double() + r[1]["layer"][k].double()
        assert torch.allclose(got, want, rtol=1e-4, atol=1e-4), k
    # the UNet keeps the native executor: its op list is issued in parts around every layer's statistics exchange
    assert r[0]["info"]["native_prog"] and r[1]["info"]["native_prog"], "the synced pass must run through the executor"
    for rr in r:
        a, b = rr["info"]["loss_native_vs_walk"]
        assert abs(a - b) <= 2e-4 * max(1.0, abs(a)), (a, b)
    assert r[0]["info"]["voxels"] != r[1]["info"]["voxels"]
    for n in r[0]["weights"]:
        assert torch.equal(r[0]["weights"][n], r[1]["weights"][n]), n
    stats = [n for n in r[0]["buffers"] if n.endswith("running_mean") or n.endswith("running_var")]
    assert len(stats) > 80
    for n in stats:
        assert torch.equal(r[0]["buffers"][n], r[1]["buffers"][n]), n
    assert all(l == l for l in r[0]["info"]["losses"] + r[1]["info"]["losses"])
