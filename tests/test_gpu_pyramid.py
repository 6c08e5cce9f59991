"""The rulebook pyramid from ONE native call (wsis_rulebook_pyramid: every level's row count known on the host, one
arena, no read-back) against the per-table build it replaces in the issuing thread: identical tables, orders, coarse
coordinates -- and the same training step."""
import numpy as np
import pytest
import torch

import harness
import spconv
from spconv import ops

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _tensor(batch):
    t = spconv.SparseConvTensor(torch.zeros(batch["voxel_coords_int"].shape[0], 1, device=DEV), batch["voxel_coords_int"],
                                batch["spatial_shape"], len(batch["scene_list"]))
    t._level_counts = batch["level_counts"]
    return t


@pytest.mark.parametrize("seeds,room", [((2,), (2.4, 2.0, 1.6)), ((5, 6, 7), (1.5, 1.3, 1.1))])
def test_native_pyramid_equals_the_per_table_build(seeds, room, monkeypatch):
    batch = harness.to_device(harness.collate([harness.bench_scene(s, room=room, n_box=2) for s in seeds]), DEV)
    a = _tensor(batch)
    ops.prebuild_unet_rulebooks(a, 5)
    assert isinstance(a.indice_dict, ops.PyramidDict), "the batch carries level counts: the native build must be taken"
    monkeypatch.setenv("WSIS_PYRAMID_NATIVE", "0")
    b = _tensor(batch)
    ops.prebuild_unet_rulebooks(b, 5)
    assert not isinstance(b.indice_dict, ops.PyramidDict)
    torch.cuda.synchronize()
    ops.verify_pending_counts()
    assert set(a.indice_dict) == set(b.indice_dict) and len(a.indice_dict) == 9
    for k in b.indice_dict:
        ra, rb = a.indice_dict[k], b.indice_dict[k]
        for name in ("nbr", "nbr_up", "order", "order_up", "nbr_p", "nbr_up_p", "out_indices", "in_indices"):
            x, y = getattr(ra, name, None), getattr(rb, name, None)
            assert (x is None) == (y is None), (k, name)
            if x is not None:
                assert x.shape == y.shape and torch.equal(x, y), (k, name)
        assert list(ra.out_shape) == list(rb.out_shape) and list(ra.in_shape) == list(rb.in_shape)
        # the coordinate hash answers the same queries
        ka, va, ca = ra.out_hash
        kb, vb, cb = rb.out_hash
        assert ca == cb
        sa, sb = torch.sort(ka).values, torch.sort(kb).values
        assert torch.equal(sa, sb)
        assert torch.equal(va[torch.argsort(ka)], vb[torch.argsort(kb)])


def test_wrong_level_counts_are_caught_inside_the_pass(monkeypatch):
    batch = harness.to_device(harness.collate([harness.bench_scene(9, room=(1.6, 1.4, 1.2), n_box=1)]), DEV)
    t = _tensor(batch)
    t._level_counts = [c + (1 if i == 1 else 0) for i, c in enumerate(batch["level_counts"])]
    ops.prebuild_unet_rulebooks(t, 5)
    torch.cuda.synchronize()
    with pytest.raises(Exception):
        ops.verify_pending_counts()


def test_training_step_is_the_same_with_either_build(monkeypatch):
    cfg = harness.default_cfg()
    scene = harness.bench_scene(12, room=(2.0, 1.7, 1.3), n_box=2)
    losses = []
    for native in ("1", "0"):
        monkeypatch.setenv("WSIS_PYRAMID_NATIVE", native)
        batch = harness.to_device(harness.collate([scene]), DEV)
        model, crit, opt = harness.build_model(cfg, DEV)
        out = [float(harness.train_step(model, crit, opt, batch, cfg)[0]) for _ in range(2)]
        losses.append(out)
    assert losses[0] == losses[1]


@pytest.mark.parametrize("native", ["1", "0"])
@pytest.mark.parametrize("with_counts", [True, False])
def test_batch_size_below_the_number_of_scenes_fails_loudly_on_every_build(monkeypatch, native, with_counts):
    """A ``SparseConvTensor`` whose ``batch_size`` is below the number of scenes in ``indices`` is a caller error
    (upstream indexes past its dense grid).  Both builds refuse it instead of building different tables: the native
    pyramid leaves the out-of-range rows out (bitmap AND sort form) and its count check raises; the per-table walk
    counts the offending rows -- with host counts in the same deferred read, without them right away."""
    import wsis_native as _n
    monkeypatch.setenv("WSIS_PYRAMID_NATIVE", native)
    scenes = [harness.bench_scene(s, room=(1.4, 1.2, 1.0), n_box=1) for s in (21, 22)]
    batch = harness.to_device(harness.collate(scenes), DEV)
    t = spconv.SparseConvTensor(torch.zeros(batch["voxel_coords_int"].shape[0], 1, device=DEV), batch["voxel_coords_int"],
                                batch["spatial_shape"], 1)                      # two scenes, batch_size 1
    if with_counts:
        t._level_counts = batch["level_counts"]
    with pytest.raises(_n.WsisError):
        ops.prebuild_unet_rulebooks(t, 5)
        torch.cuda.synchronize()
        ops.verify_pending_counts()
    # a valid tensor right behind it builds normally (nothing of the refused build is left pending)
    ok = _tensor(batch)
    ops.prebuild_unet_rulebooks(ok, 5)
    torch.cuda.synchronize()
    ops.verify_pending_counts()
