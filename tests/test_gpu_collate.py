"""Device-side batch assembly: ``harness.collate_device`` hashes the points on the GPU
(``voxelization_idx`` with a CUDA LongTensor) and counts the pyramid's levels there; it must hand the step exactly the
batch ``to_device(collate(...))`` builds with the reference's single host thread per worker
(modules/datasets/scannetv2_dataset.py:445-449, train_scannetv2.py:149-194)."""
import numpy as np
import pytest
import torch

import harness
from spconv import ops

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("seeds,room", [((3,), (2.2, 1.8, 1.4)), ((4, 5, 6), (1.4, 1.2, 1.0))])
def test_device_collate_equals_host_collate(seeds, room):
    scenes = [harness.bench_scene(s, room=room, n_box=2) for s in seeds]
    host = harness.to_device(harness.collate(scenes), DEV)
    dev = harness.collate_device(scenes, DEV)
    torch.cuda.synchronize()
    for k in ("voxel_locs", "p2v_map", "v2p_map", "voxel_coords_int", "locs_float", "feats", "superpoint",
              "edge_u_list", "edge_v_list", "semantic_labels", "superpoint_instance_labels"):
        assert dev[k].is_cuda and torch.equal(dev[k], host[k]), k
    assert list(dev["level_counts"]) == list(host["level_counts"])
    assert np.array_equal(dev["spatial_shape"], host["spatial_shape"])
    assert dev["sp_instance_slots"] == host["sp_instance_slots"] and dev["edge_src_rows"] == host["edge_src_rows"]
    # the device counts against the oracle's pyramid as well
    want = ops.level_voxel_counts(host["voxel_locs"].cpu().numpy(), host["spatial_shape"], 5)
    got = ops.level_voxel_counts_device(dev["voxel_locs"], dev["spatial_shape"], 5).tolist()
    assert [int(v) for v in got] == want


@pytest.mark.parametrize("seeds,room", [((3,), (2.2, 1.8, 1.4)), ((4, 5, 6), (1.4, 1.2, 1.0))])
def test_packed_collate_equals_host_collate(seeds, room):
    """``pack_scene`` (one pinned buffer per sample) + ``collate_packed`` (one H2D per scene, everything else on the
    device): every tensor of the batch, the ECC graph and the host-side numbers equal ``to_device(collate(scenes))``"""
    scenes = [harness.bench_scene(s, room=room, n_box=2) for s in seeds]
    host = harness.to_device(harness.collate(scenes), DEV)
    dev = harness.collate_packed([harness.pack_scene(sc) for sc in scenes], DEV)
    torch.cuda.synchronize()
    for k in ("voxel_locs", "p2v_map", "v2p_map", "voxel_coords_int", "locs_float", "feats", "superpoint",
              "edge_u_list", "edge_v_list", "semantic_labels", "instance_labels", "superpoint_semantic_labels",
              "superpoint_instance_labels", "superpoint_offset_vector", "superpoint_instance_voxel_num",
              "superpoint_instance_size"):
        assert dev[k].is_cuda and dev[k].dtype == host[k].dtype, k
        assert torch.equal(dev[k], host[k]), k
    for k in ("offsets", "sp_batch_offsets"):
        assert torch.equal(dev[k].cpu(), host[k].cpu()), k
    assert list(dev["level_counts"]) == list(host["level_counts"])
    assert np.array_equal(dev["spatial_shape"], host["spatial_shape"])
    assert dev["sp_instance_slots"] == host["sp_instance_slots"] and dev["edge_src_rows"] == host["edge_src_rows"]
    gd, gh = dev["GIs"][0], host["GIs"][0]
    assert gd.num_nodes == gh.num_nodes and torch.equal(gd.get_pyg_buffers(), gh.get_pyg_buffers())
    assert torch.equal(gd.get_buffers(), gh.get_buffers())
    for k in ("superpoint_csr", "p2v_csr"):
        assert torch.equal(dev[k].perm, host[k].perm) and torch.equal(dev[k].offsets, host[k].offsets), k


def test_a_step_on_the_packed_batch_gives_the_same_loss():
    cfg = harness.default_cfg()
    scenes = [harness.bench_scene(8, room=(2.0, 1.6, 1.2), n_box=2)]
    losses = []
    for make in (lambda: harness.to_device(harness.collate(scenes), DEV),
                 lambda: harness.collate_packed([harness.pack_scene(sc) for sc in scenes], DEV)):
        model, crit, opt = harness.build_model(cfg, DEV)
        loss, _ = harness.train_step(model, crit, opt, make(), cfg)
        losses.append(float(loss))
    assert losses[0] == losses[1]


def test_a_step_on_the_device_collated_batch_gives_the_same_loss():
    cfg = harness.default_cfg()
    scenes = [harness.bench_scene(8, room=(2.0, 1.6, 1.2), n_box=2)]
    losses = []
    for make in (lambda: harness.to_device(harness.collate(scenes), DEV), lambda: harness.collate_device(scenes, DEV)):
        model, crit, opt = harness.build_model(cfg, DEV)
        loss, _ = harness.train_step(model, crit, opt, make(), cfg)
        losses.append(float(loss))
    assert losses[0] == losses[1]


def test_batched_csr_build_equals_the_single_builds():
    """torch_scatter.segment_csr_batch: the CSRs of several index vectors from one sort, identical to SegmentCSR each
    (stable sort, local row numbers), empty vector and unsorted ids included"""
    from torch_scatter import SegmentCSR, segment_csr_batch
    g = torch.Generator(device=DEV).manual_seed(3)
    pairs = [(torch.randint(0, 700, (20011,), device=DEV, generator=g), 700),
             (torch.randint(0, 15000, (19000,), device=DEV, generator=g).int(), 15001),
             (torch.randint(0, 50, (3,), device=DEV, generator=g), 64),
             (torch.zeros(0, dtype=torch.int64, device=DEV), 5),
             (torch.randint(0, 2289, (20054,), device=DEV, generator=g), 2289),
             (torch.arange(999, -1, -1, device=DEV), 1000)]
    got = segment_csr_batch(pairs)
    torch.cuda.synchronize()
    for (index, S), c in zip(pairs, got):
        ref = SegmentCSR(index, S)
        assert c.N == ref.N and c.S == ref.S
        assert torch.equal(c.offsets, ref.offsets)
        if c.N:
            assert torch.equal(c.perm[:c.N], ref.perm[:ref.N])
        assert torch.equal(c.index, ref.index)


@pytest.mark.parametrize("counting", ["1", "0"])
def test_counting_csr_build_handles_every_segment_size(monkeypatch, counting):
    """round 6: the batched CSRs by counting (count per segment, scan, place at an atomic cursor, order every segment's
    slice by value) against the single-table sort: segments of one row, of 65 .. 2,048 rows (bitonic network in LDS), one
    of 9,000 rows (the network on the slice in memory), empty segments, ids outside [0, S) (one bucket behind the last
    segment: never visible through the offsets) -- and the sort path (WSIS_CSR_COUNTING=0, the default: the counting form
    measured slower in the step) stays what it was"""
    from torch_scatter import SegmentCSR, segment_csr_batch
    monkeypatch.setenv("WSIS_CSR_COUNTING", counting)
    g = torch.Generator(device=DEV).manual_seed(11)
    big = torch.cat([torch.full((9000,), 3, device=DEV), torch.randint(0, 40, (6000,), device=DEV, generator=g)])
    big = big[torch.randperm(big.numel(), device=DEV, generator=g)]
    mid = torch.randint(0, 12, (9000,), device=DEV, generator=g)                  # ~750 rows per segment
    sparse = torch.randint(0, 200000, (150000,), device=DEV, generator=g)         # mostly empty / one-row segments
    bad = torch.randint(0, 50, (4000,), device=DEV, generator=g)                  # ids below S only are segments
    pairs = [(big, 40), (mid, 12), (sparse, 200000), (bad, 30), (torch.zeros(1, dtype=torch.int64, device=DEV), 1)]
    got = segment_csr_batch(pairs)
    torch.cuda.synchronize()
    for (index, S), c in zip(pairs, got):
        assert sorted(c.perm[:c.N].tolist()) == list(range(c.N))      # a permutation of all rows
        if index is bad:
            if counting == "1":       # (ids >= S are a caller error: the counting form keeps them behind offsets[S])
                ok = index < S
                want = torch.argsort(index[ok], stable=True)
                rows = torch.nonzero(ok).flatten()[want]
                n_in = int(ok.sum())
                assert int(c.offsets[-1]) == n_in and torch.equal(c.perm[:n_in].long(), rows)
            continue
        ref = SegmentCSR(index, S)
        assert torch.equal(c.offsets, ref.offsets)
        assert torch.equal(c.perm[:c.N], ref.perm[:ref.N])
