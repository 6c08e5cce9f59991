import importlib, sys, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))); importlib.import_module("3d-wsis_amd")
import torch, harness, pointgroup_ops
device = torch.device("cuda")
scenes = [harness.make_scene(s, room=(3.2, 2.6, 2.2), n_box=4) for s in (1, 2, 3, 4)]
b = harness.collate(scenes)
sem = b["superpoint"] % 20
keep = torch.nonzero(sem > 1).flatten()
coords = b["locs_float"][keep].contiguous().to(device)
batch_idx = b["locs"][keep, 0].int().contiguous()
offs = torch.zeros(len(scenes) + 1, dtype=torch.int32)
offs[1:] = torch.cumsum(torch.bincount(batch_idx.long(), minlength=len(scenes)), 0).int()
bi_d, off_d = batch_idx.to(device), offs.to(device)
for rep in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    idx, sl = pointgroup_ops.ballquery_batch_p(coords, bi_d, off_d, 0.03, 50)
    torch.cuda.synchronize(); print("ballquery wall ms", (time.perf_counter() - t0) * 1e3, idx.numel())
