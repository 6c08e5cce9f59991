"""The resident deep-level kernel (csrc/deep.hip: a run of the executor's op list on tensors of at most WSIS_DEEP_ROWS rows
as ONE launch with grid barriers between its phases) against the same op list issued launch by launch (WSIS_DEEP=0).

Every phase runs the code of the launch it replaces (fwd2_body with the one-shot kernel's plan; the BatchNorm finish /
apply arithmetic of bn.hip in the same order), so the comparison is for EQUALITY: loss, every output row, every
parameter gradient and every running statistic.  The one-shot side runs with WSIS_SLAB_BN_PARTIALS=1: where a dIn
product is split into offset slabs (levels of <= 96 work items) the resident kernel takes the BatchNorm-backward sums
from the slab sum's epilogue, the default one-shot path from a reduction of its own (same values to rounding, a
different order) -- with the switch both sides use the first form.  The parity of either path against the fp64 oracle
is the business of tests/test_gpu_network.py (which run with the resident kernel on: it is the default)."""
import os

import pytest
import torch

import harness
import wsis_native

pytestmark = [pytest.mark.gpu, pytest.mark.experimental]


def _one_pass(monkeypatch, deep, batch_host, cfg, train=True, rows=None, fence=None):
    monkeypatch.setenv("WSIS_DEEP", "1" if deep else "0")
    monkeypatch.setenv("WSIS_SLAB_BN_PARTIALS", "1")
    torch.manual_seed(0)
    model, crit, opt = harness.build_model(cfg, "cuda")
    if not train:
        model.eval()
    batch = harness.to_device(batch_host, "cuda")
    if train:
        loss, ret = harness.forward_loss(model, crit, batch, cfg)
        loss.backward()
    else:
        with torch.no_grad():
            loss, ret = harness.forward_loss(model, crit, batch, cfg)
    torch.cuda.synchronize()
    grads = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    stats = {n: v.clone() for n, v in model.state_dict().items() if "running_" in n}
    return float(loss), grads, stats, ret["semantic_scores"].detach().clone()


def _launch_counts(monkeypatch, deep, batch_host, cfg):
    """number of convolution records the profiler sees per pass does not change with the resident kernel"""
    from spconv import ops as sp_ops
    monkeypatch.setenv("WSIS_DEEP", "1" if deep else "0")
    monkeypatch.setenv("WSIS_DW_STREAM", "0")
    torch.manual_seed(0)
    model, crit, opt = harness.build_model(cfg, "cuda")
    batch = harness.to_device(batch_host, "cuda")
    harness.train_step(model, crit, opt, batch, cfg)
    sp_ops.PROFILER = sp_ops.KernelProfiler()
    try:
        harness.train_step(model, crit, opt, batch, cfg)
        s = sp_ops.PROFILER.summary()
    finally:
        sp_ops.PROFILER = None
    return s


@pytest.mark.parametrize("room,n_box", [((1.8, 1.4, 1.1), 2), ((4.2, 3.1, 2.5), 5)])
def test_resident_kernel_equals_launch_by_launch_training(monkeypatch, room, n_box):
    """small room: every level below 0 fits the row limit (levels 1-4 are one launch per pass); 4 m room: levels 2-4"""
    cfg = harness.default_cfg()
    batch_host = harness.collate([harness.make_scene(31, room=room, n_box=n_box)])
    lib = wsis_native.hip()
    l0, g0, s0, y0 = _one_pass(monkeypatch, False, batch_host, cfg)
    n0, p0 = lib.wsis_deep_launches(), lib.wsis_deep_phases()
    l1, g1, s1, y1 = _one_pass(monkeypatch, True, batch_host, cfg)
    assert lib.wsis_deep_launches() - n0 == 2                 # one resident launch per pass ...
    assert lib.wsis_deep_phases() - p0 > 100                  # ... of > 100 phases together
    assert torch.equal(y0, y1)
    assert l0 == l1
    assert set(g0) == set(g1) and len(g0) > 150
    assert [n for n in g0 if not torch.equal(g0[n], g1[n])] == []
    assert [n for n in s0 if not torch.equal(s0[n], s1[n])] == []
    assert wsis_native.sync_errors() == []


def test_resident_kernel_equals_launch_by_launch_eval(monkeypatch):
    cfg = harness.default_cfg()
    batch_host = harness.collate([harness.make_scene(32, room=(2.4, 2.0, 1.6), n_box=3)])
    l0, _, _, y0 = _one_pass(monkeypatch, False, batch_host, cfg, train=False)
    l1, _, _, y1 = _one_pass(monkeypatch, True, batch_host, cfg, train=False)
    assert torch.equal(y0, y1) and l0 == l1


def test_resident_kernel_with_a_batch_of_scenes_and_whole_network_below_the_limit(monkeypatch):
    """two tiny scenes per batch (batch items never mix in a rulebook) with WSIS_DEEP_ROWS above the level-0 row count:
    the whole UNet but its 6-channel input conv is ONE launch per pass"""
    cfg = harness.default_cfg()
    cfg.batch_size = 2
    batch_host = harness.collate([harness.make_scene(41, room=(1.0, 0.9, 0.7), n_box=1),
                                  harness.make_scene(42, room=(0.9, 1.0, 0.8), n_box=2)])
    assert batch_host["voxel_locs"].shape[0] < 34000          # (BatchNorm phases: at most 16 chunks of 64 slices)
    monkeypatch.setenv("WSIS_DEEP_ROWS", "65536")
    lib = wsis_native.hip()
    l0, g0, s0, y0 = _one_pass(monkeypatch, False, batch_host, cfg)
    p0 = lib.wsis_deep_phases()
    l1, g1, s1, y1 = _one_pass(monkeypatch, True, batch_host, cfg)
    assert lib.wsis_deep_phases() - p0 > 150                  # everything but the level-0 convolutions (1 or 2 waves per item)
    assert torch.equal(y0, y1) and l0 == l1
    assert [n for n in g0 if not torch.equal(g0[n], g1[n])] == []
    assert [n for n in s0 if not torch.equal(s0[n], s1[n])] == []


def test_resident_kernel_repeated_steps_are_reproducible(monkeypatch):
    """12 optimizer steps over scenes of three sizes, twice: identical loss sequences (the barrier words, the sub-group
    counters and the phase table ring are reused launch after launch)"""
    cfg = harness.default_cfg()
    monkeypatch.setenv("WSIS_DEEP", "1")
    rooms = [(3.0, 2.5, 2.4), (1.8, 1.4, 1.1), (4.2, 3.1, 2.5)]
    scenes = [harness.collate([harness.make_scene(300 + i, room=r, n_box=3)]) for i, r in enumerate(rooms)]

    def run():
        torch.manual_seed(0)
        model, crit, opt = harness.build_model(cfg, "cuda")
        out = []
        for it in range(12):
            b = harness.to_device(scenes[it % len(scenes)], "cuda")
            loss, _ = harness.train_step(model, crit, opt, b, cfg)
            out.append(float(loss))
        return out
    a, b = run(), run()
    assert a == b and all(x == x for x in a)
    assert wsis_native.sync_errors() == []


def test_profiler_sees_the_same_products_with_and_without_the_resident_kernel(monkeypatch):
    """bench.py's roofline takes a duration per convolution product: inside the resident launch it comes from in-kernel
    stamps (phase end to phase end, grid barrier included) -- same number of products, same bytes, plausible times"""
    cfg = harness.default_cfg()
    batch_host = harness.collate([harness.make_scene(31, room=(4.2, 3.1, 2.5), n_box=5)])
    a = _launch_counts(monkeypatch, False, batch_host, cfg)
    b = _launch_counts(monkeypatch, True, batch_host, cfg)
    for name in ("spconv_fwd_kernel", "spconv_dw_kernel"):
        assert a[name]["launches"] == b[name]["launches"] and a[name]["bytes"] == b[name]["bytes"]
    ka, kb = a["spconv_fwd_kernel"], b["spconv_fwd_kernel"]
    assert [r[:6] for r in ka["per_launch"]] == [r[:6] for r in kb["per_launch"]]
    assert all(0.0 < r[7] < 5.0 for r in kb["per_launch"])          # ms per product
