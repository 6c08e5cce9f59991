"""CPU, world_size 2, gloo: the scene-sharded data-parallel plumbing (parallel.GradSync / shard_scenes) that
bench.py uses over RCCL at N > 1.  The averaged gradients must equal those of one process that saw all scenes."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

import wsis_parallel as parallel


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _model():
    torch.manual_seed(7)
    return nn.Sequential(nn.Linear(6, 33), nn.ReLU(), nn.Linear(33, 5), nn.ReLU(), nn.Linear(5, 1))


def _data(scene):
    g = torch.Generator().manual_seed(100 + scene)
    return torch.randn(16, 6, generator=g), torch.randn(16, 1, generator=g)


def _worker(rank, world, port, out_dir):
    import importlib, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    importlib.import_module("3d-wsis_amd")
    import wsis_parallel as par
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, lr, w = par.init_distributed("gloo")
    assert (r, w) == (rank, world)
    model = _model()
    sync = par.GradSync(model, bucket_bytes=400)     # tiny buckets -> several collectives
    assert len(sync.buckets) > 2
    scenes = par.shard_scenes(list(range(4)), rank, world)
    loss = 0.0
    for s in scenes:
        x, y = _data(s)
        loss = loss + ((model(x) - y) ** 2).mean()
    (loss / len(scenes)).backward()
    # one parameter without a gradient on rank 1 only (unused branch) must still take part
    if rank == 1:
        model[4].bias.grad = None
    sync(model)
    torch.save([p.grad.clone() for p in model.parameters()], os.path.join(out_dir, f"g{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_gradsync_matches_single_process(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    g0 = torch.load(os.path.join(tmp_path, "g0.pt"))
    g1 = torch.load(os.path.join(tmp_path, "g1.pt"))
    for a, b in zip(g0, g1):
        assert torch.equal(a, b), "ranks must hold identical averaged gradients"
    # single-process reference: mean over the per-rank losses
    model = _model()
    total = 0.0
    per_rank = []
    for rank in range(world):
        loss = 0.0
        scenes = parallel.shard_scenes(list(range(4)), rank, world)
        for s in scenes:
            x, y = _data(s)
            loss = loss + ((model(x) - y) ** 2).mean()
        per_rank.append(loss / len(scenes))
    (sum(per_rank) / world).backward()
    params = list(model.parameters())
    for i, (a, p) in enumerate(zip(g0, params)):
        ref = p.grad.clone()
        if i == len(params) - 1:
            # last bias: rank 1 contributed zeros (its grad was None)
            continue
        assert torch.allclose(a, ref, rtol=1e-5, atol=1e-7)


def test_shard_scenes_partition():
    ids = list(range(32))
    parts = [parallel.shard_scenes(ids, r, 8) for r in range(8)]
    assert sorted(sum(parts, [])) == ids and all(len(p) == 4 for p in parts)
    assert parallel.shard_scenes(ids, 0, 1) == ids
