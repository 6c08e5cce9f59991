"""Child process of tests/test_gpu_two_ranks.py: ONE rank of a world of two, both on cuda:0, process group on gloo
(RCCL refuses two ranks on one device; the collectives' control flow -- plan agreement, flat in-place all-reduce, tail
packing, early half exchange from inside the backward pass -- is the same code on either backend).

Steps the real ``Network`` on this rank's own scene through ``harness.train_step(..., grad_sync=GradSync)`` -- the
replacement of the reference's never-initialised DDP wrapper (train_scannetv2.py:734-738) -- and writes what the
parent test compares: local (unsynchronised) gradients of the last step, the synchronised gradients, counters.

    python tests/_two_rank_worker.py <out_dir> <scenario>      (RANK / WORLD_SIZE / MASTER_* from the environment)
scenario "steady": 6 steps, overlap on;  "broken": rank 1 breaks its flat gradient layout during plan agreement;
"syncbn": BatchNorm statistics shared across the ranks (wsis_parallel.convert_sync_batchnorm): one layer on a split
tensor, then two training steps of the converted Network."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
importlib.import_module("3d-wsis_amd")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import harness  # noqa: E402
import wsis_parallel as parallel  # noqa: E402


def local_gradients(model, crit, batch, cfg):
    """this rank's own gradients for the current weights: one forward + backward without exchange or update"""
    prog = getattr(model, "_native_prog", None)
    hook = prog.overlap if prog is not None else None
    if prog is not None:
        prog.overlap = None                      # no early exchange from inside THIS backward pass
    for p in model.parameters():
        p.grad = None
    loss, _ = harness.forward_loss(model, crit, batch, cfg)
    loss.backward()
    out = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
    if prog is not None:
        prog.overlap = hook
    return out


def main():
    out_dir, scenario = sys.argv[1], sys.argv[2]
    rank, _, world = parallel.init_distributed()
    assert world == 2 and dist.get_backend() == "gloo"
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    cfg = harness.default_cfg()
    scene = harness.bench_scene(11 + rank, room=(1.8 + 0.2 * rank, 1.5, 1.2), n_box=2)     # different scenes per rank
    batch = harness.to_device(harness.collate([scene]), dev)
    model, crit, opt = harness.build_model(cfg, dev)
    for t in list(model.parameters()) + list(model.buffers()):
        dist.broadcast(t.data, 0)
    sync = parallel.GradSync(model)
    info = {"rank": rank, "voxels": int(batch["voxel_locs"].shape[0])}

    if scenario == "syncbn":
        import wsis_ops
        # (1) one BatchNorm(+ReLU) layer: the ranks hold 700 / 1300 rows of one seeded [2000, 24] tensor
        g = torch.Generator().manual_seed(5)
        x_full = torch.randn(2000, 24, generator=g) * 2.0 + 0.5
        w_full = torch.randn(2000, 24, generator=g)
        gamma, beta = torch.rand(24, generator=g) + 0.5, torch.randn(24, generator=g) * 0.1
        rows = slice(0, 700) if rank == 0 else slice(700, 2000)
        bn = torch.nn.BatchNorm1d(24, eps=1e-4, momentum=0.1).to(dev)
        with torch.no_grad():
            bn.weight.copy_(gamma)
            bn.bias.copy_(beta)
        parallel.convert_sync_batchnorm(bn)
        xr = x_full[rows].to(dev).requires_grad_(True)
        y = wsis_ops.batch_norm_relu(xr, bn, relu=True)
        (y * w_full[rows].to(dev)).sum().backward()
        layer = {"y": y.detach().cpu(), "dx": xr.grad.cpu(), "dgamma": bn.weight.grad.cpu(), "dbeta": bn.bias.grad.cpu(),
                 "running_mean": bn.running_mean.cpu(), "running_var": bn.running_var.cpu(), "rows": (rows.start, rows.stop)}
        # (1b) |mean| = 1000 sigma: the cross-rank merge must not cancel (Chan's combination, not E[x^2] - mean^2)
        x_big = torch.randn(2000, 24, generator=g) * 0.05 + 50.0
        bn_b = torch.nn.BatchNorm1d(24, eps=1e-6, momentum=1.0).to(dev)      # running_var = this batch
        parallel.convert_sync_batchnorm(bn_b)
        yb = wsis_ops.batch_norm_relu(x_big[rows].to(dev), bn_b, relu=False)
        layer["big_y"], layer["big_running_var"] = yb.detach().cpu(), bn_b.running_var.cpu()
        # (2) the converted Network: two steps, statistics and weights must stay identical across the ranks
        parallel.convert_sync_batchnorm(model)
        assert parallel.sync_batchnorm_active(model)
        losses = []
        for _ in range(2):
            harness.build_batch_graphs(batch)
            loss, _ = harness.train_step(model, crit, opt, batch, cfg, grad_sync=sync)
            losses.append(float(loss))
        torch.cuda.synchronize()
        prog = getattr(model, "_native_prog", None)
        info.update(losses=losses, native_prog=prog is not None and prog.bn_sync is not None)
        # the same forward pass as the per-module walk with _SyncBatchNormReLU (statistics from a pass over x instead of
        # the convolution epilogues' partials): the losses agree to rounding
        both = []
        for native in ("1", "0"):
            os.environ["WSIS_SYNC_BN_NATIVE"] = native
            with torch.no_grad():
                l2, _ = harness.forward_loss(model, crit, batch, cfg)
            both.append(float(l2))
        os.environ["WSIS_SYNC_BN_NATIVE"] = "1"
        info["loss_native_vs_walk"] = both
        torch.save({"layer": layer, "info": info,
                    "weights": {n: p.detach().cpu() for n, p in model.named_parameters()},
                    "buffers": {n: b.detach().cpu() for n, b in model.named_buffers()}},
                   os.path.join(out_dir, f"syncbn{rank}.pt"))
        dist.barrier()
        dist.destroy_process_group()
        return

    if scenario == "broken":
        # call 1 of the plan agreement: rank 1's UNet gradients no longer live in the flat buffer (one was replaced)
        loss, _ = harness.forward_loss(model, crit, batch, cfg)
        loss.backward()
        local = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
        if rank == 1:
            p = model._native_prog.params[3]
            p.grad = p.grad.clone()
            head = next(p for n, p in model.named_parameters() if n.startswith("linear.3"))
            head.grad = None                      # and one parameter without any gradient on this rank
        sync(model)
        torch.cuda.synchronize()
        info["flat_params"] = sync.last_flat_params
        info["agreed"] = list(sync._agreed)
        torch.save({"local": {k: v.cpu() for k, v in local.items()},
                    "synced": {n: p.grad.detach().cpu() for n, p in model.named_parameters() if p.grad is not None},
                    "info": info}, os.path.join(out_dir, f"broken{rank}.pt"))
        dist.barrier()
        dist.destroy_process_group()
        return

    steps = 6                                     # past GradSync.AGREE_CALLS: the plan is frozen, the overlap is live
    losses = []
    for i in range(steps - 1):
        harness.build_batch_graphs(batch)
        loss, _ = harness.train_step(model, crit, opt, batch, cfg, grad_sync=sync)
        losses.append(float(loss))
    # last step: first this rank's own gradients for the current weights, then the real step
    local = local_gradients(model, crit, batch, cfg)
    early_before = getattr(sync, "early_count", 0)
    harness.build_batch_graphs(batch)
    loss, _ = harness.train_step(model, crit, opt, batch, cfg, grad_sync=sync)
    torch.cuda.synchronize()
    losses.append(float(loss))
    ecc = {id(p) for p in model.ecc.parameters()}
    info.update(early_count=getattr(sync, "early_count", 0), early_last_step=getattr(sync, "early_count", 0) - early_before,
                flat_params=sync.last_flat_params, agreed=list(sync._agreed), losses=losses,
                ecc_names=[n for n, p in model.named_parameters() if id(p) in ecc],
                n_unet_params=len(model._native_prog.params))
    torch.save({"local": {k: v.cpu() for k, v in local.items()},
                "synced": {n: p.grad.detach().cpu() for n, p in model.named_parameters() if p.grad is not None},
                "weights": {n: p.detach().cpu() for n, p in model.named_parameters()},
                "info": info}, os.path.join(out_dir, f"steady{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
