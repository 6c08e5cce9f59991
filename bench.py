#!/usr/bin/env python
"""bench.py -- scenes/s, forward+backward(+AdamW step) of the 3D-WSIS per-scene hot path on MI355X.

Workload at N = 1 (BASELINE.json configs[1], SURVEY.md 8d "C2"): one synthetic ScanNet-shaped scene
(~150 k active voxels at 2 cm, ~2.3 k superpoints, ~20 k graph edges), inputs resident in HBM before the timed
region; one step = per-batch segment / graph structures -> superpoint centres -> voxelization -> rulebooks ->
SparseConvTensor -> Network (SubMConv3d UNet + ECC GNN + edge affinity) -> MultiTaskLoss (stage-3 switches) ->
backward -> [RCCL gradient all-reduce when N > 1] -> ECC grad clamp -> AdamW step (train_scannetv2.py:143-252).
N > 1 (BASELINE.json configs[4], "C5"): --scenes-per-gpu defaults to 4 (batch 32 over 8 GPUs), scenes sharded by
rank, no data-path collective, weak scaling.  `python bench.py --gpus N` without a launcher starts the N ranks itself
(a torch.distributed.run child, before anything touches the GPU) and fails non-zero if the RCCL world is not N.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline      -- dominant kernel family (the forward / dIn launches of the 49 sparse convs: spconv_fwd2_kernel,
                   spconv_in_kernel for the 6-channel input conv): algorithmic gather/scatter bytes (P*(Cin+Cout)*4 + P*8 per launch,
                   SURVEY 8d) / HIP-event time of those launches, against the 8 TB/s HBM peak.
  cpu_baseline  -- the oracle (torch-CPU restatement mirroring upstream's gather -> mm -> scatter-add) timed on
                   this box's host cores on the same scene (rank 0, N = 1 only).
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
importlib.import_module("3d-wsis_amd")

# Multi-process runs: with an RCCL process group up, its helper threads wait on HSA signals all the time and the
# interrupt-driven wait path slows the (launch-bound) issuing thread: 63.0 scenes/s against 66.9 without a group on
# one GPU; polling waits bring it back to 65.9.  Must be set before the HSA runtime starts (= before torch touches
# the GPU); single-process runs are left alone.
if int(os.environ.get("WORLD_SIZE", "1")) > 1 or os.environ.get("WSIS_FORCE_DIST", "0") == "1":
    os.environ.setdefault("HSA_ENABLE_INTERRUPT", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)
# fp64 matrix peak: AMD's MI355X datasheet figure (78.6 TFLOP/s, v_mfma_f64_16x16x4_f64: 256 FLOP/clk/CU x 256 CUs x
# 2.4 GHz = 157 -- the datasheet halves it for fp64); MI355X_MICROARCH.md has no fp64 row, so the datasheet number is used
FP64_MATRIX_PEAK_TF = 78.6
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "conv_traffic.json")   # written by profiles/collect_traffic.py


def measured_traffic():
    """HBM bytes per spconv_fwd_kernel launch from rocprofv3 PMC passes of this same command (FETCH_SIZE doubled
    per the gfx950 note in MI355X_MICROARCH.md, WRITE_SIZE as is), or None when no counter run is on file."""
    try:
        with open(TRAFFIC_FILE) as f:
            t = json.load(f)
        return float(t["hbm_bytes_per_launch"])
    except (OSError, KeyError, ValueError):
        return None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)      # 40 x ~12 ms: one hiccup moves the mean by < 1 %
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--pipeline", action="store_true",
                    help="build the next step's rulebooks in slices between the phases of the current step "
                         "(spconv.ops.RulebookPipeline; measured neutral: 69.0 / 69.3 vs 68.9 / 69.1 scenes/s)")
    ap.add_argument("--prefetch", action="store_true",
                    help="build the next step's rulebooks from a helper thread (measured slower: GIL contention)")
    ap.add_argument("--profile-steps", type=int, default=3,
                    help="extra event-instrumented steps for the roofline (>= 3: per launch the median of its occurrences)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=5)
    ap.add_argument("--cpu-warmup", type=int, default=3)
    ap.add_argument("--cpu-threads", type=int, default=8)
    ap.add_argument("--cpu-timeout", type=int, default=240)
    ap.add_argument("--cpu-worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--setup-steps", type=int, default=300,
                    help="untimed initialisation passes before the W warm-up steps: lazy code-object loads, allocator "
                         "growth and -- the long part -- the engine clock: the step time of this MFMA-heavy job keeps "
                         "falling for the first few hundred steps of a process (13.1 ms after 3 passes, 12.0 ms after "
                         "300 or 600, DESIGN section 5); a training run sits in that steady state")
    ap.add_argument("--no-stages", action="store_true",
                    help="skip the side measurements (copy/triad bandwidth, optimizer step, clustering stage)")
    ap.add_argument("--scene-seed", type=int, default=1)
    ap.add_argument("--scenes-per-gpu", type=int, default=0,
                    help="scenes per rank and step; default 1 at --gpus 1 (C2), 4 at --gpus > 1 (C5: batch 32 / 8 GPUs)")
    ap.add_argument("--hoist-graphs", action="store_true",
                    help="build the per-batch segment CSRs / edge graph once outside the step (round-1 behaviour) "
                         "instead of inside every timed step")
    ap.add_argument("--small", action="store_true", help="debug: a small room instead of the C2 scene")
    ap.add_argument("--sync-bn", action="store_true",
                    help="N > 1: BatchNorm statistics over all ranks (the reference's SyncBatchNorm conversion): one small "
                         "collective per BatchNorm layer and pass, issued between two parts of the executor's op list")
    return ap.parse_args()


MFMA_FP32_PEAK_TFLOPS = 157.3      # dense fp32 matrix peak (v_mfma_f32_32x32x2_f32: 256 FLOP/clk/CU x 256 CUs x 2.4 GHz)
C1_ROOM, C1_POINTS = (3.0, 3.0, 2.4), 10000      # BASELINE configs[0] / SURVEY 8d C1: single 10 k-pt room (seed 0)


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) without torchrun's environment: start the N ranks as ONE child job
    (python -m torch.distributed.run, one process per GPU) BEFORE this process touches the GPU, relay its output and
    exit with its code.  Returns only when no launch is needed."""
    if args.gpus <= 1 or args.cpu_worker:
        return
    ws = os.environ.get("WORLD_SIZE")
    if ws is not None:
        if int(ws) != args.gpus:
            sys.stderr.write(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={ws}\n")
            sys.exit(2)
        return
    import subprocess
    # --standalone: the launcher binds its own rendezvous port (a port probed here and closed again could be taken by
    # another job before the launcher binds it)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--standalone", "--local-addr", "127.0.0.1", os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.exit(subprocess.run(cmd, env=env).returncode)


def cpu_worker(args):
    """child process: oracle fwd+bwd on the host cores (never touches the GPU); prints one JSON line."""
    import harness
    from oracle import network_ref
    import losses_3D_WSIS
    threads = args.cpu_threads
    torch.set_num_threads(threads)
    cfg = harness.default_cfg()
    c1 = c1_cpu_leg(harness)
    print(json.dumps({"c1": c1}), flush=True)          # survives a time-out of the C2 passes below
    scene = harness.bench_scene(args.scene_seed)        # the SAME scene the GPU leg steps on
    batch_host = harness.collate([scene])
    torch.manual_seed(123)
    ref = network_ref.RefNetwork()
    crit = losses_3D_WSIS.MultiTaskLoss(None, cfg.loss, cfg.model)

    def one():
        ref.zero_grad(set_to_none=True)
        loss, _ = network_ref.forward_loss_cpu(ref, crit, batch_host)
        loss.backward()

    t0 = time.time()
    one()                       # first warm-up pass (first-touch page faults dominate it)
    warm = time.time() - t0
    out = {"c1": c1, "warm": warm, "warm_passes": 1, "iters": 0, "dt": warm,
           "voxels": int(batch_host["voxel_locs"].shape[0]), "threads": threads}
    print(json.dumps(out), flush=True)
    for w in range(1, max(args.cpu_warmup, 1)):
        t1 = time.time()
        one()
        out.update(warm_passes=w + 1, dt=time.time() - t1)
        print(json.dumps(out), flush=True)
    t0 = time.time()
    for it in range(args.cpu_iters):
        one()
        out.update(iters=it + 1, dt=(time.time() - t0) / (it + 1))
        print(json.dumps(out), flush=True)              # the parent keeps the last complete line


def c1_cpu_leg(harness):
    """BASELINE configs[0]: single 10 k-pt room, voxelize + superpoint scatter-mean + affinity on the
    host cores -- the oracle's functions (oracle/pg_ops, scatter_ref, affinity_ref), the reference's CPU plumbing."""
    import numpy as np
    from oracle import affinity_ref, pg_ops, scatter_ref
    sc = harness.make_scene(0, room=C1_ROOM, n_box=2, max_points=C1_POINTS)
    b = harness.collate([sc])
    S = int(b["sp_batch_offsets"][-1])
    rng = np.random.default_rng(0)
    q, k, v = (torch.from_numpy(rng.standard_normal((S, 64)).astype("float32")) for _ in range(3))
    pos = torch.from_numpy(rng.standard_normal(int(b["edge_u_list"].shape[0])).astype("float32"))
    feats = torch.cat([b["feats"], b["locs_float"]], 1)

    def one():
        vl, p2v, v2p = pg_ops.voxelization_idx(b["locs"].numpy(), 1, 4)
        vf = torch.from_numpy(pg_ops.voxelization(feats.numpy(), v2p, 4))
        pooled = scatter_ref.scatter(vf[torch.from_numpy(p2v).long()], b["superpoint"], dim=0, reduce="mean")
        aff, res = affinity_ref.edge_affinity(q, k, v, pos, b["edge_u_list"], b["edge_v_list"])
        A = affinity_ref.affinity_matrix(b["edge_u_list"].numpy(), b["edge_v_list"].numpy(), aff.numpy(), S)
        return vl.shape[0], pooled, A

    one()
    t0 = time.time()
    n = 0
    while n < 3 or (time.time() - t0 < 2.0 and n < 50):
        M, _, _ = one()
        n += 1
    dt = (time.time() - t0) / n
    return {"workload": "C1: single 10 k-pt room, voxelization_idx + voxelization + superpoint scatter-mean + edge "
                        "affinity + dense affinity matrix on the host (oracle)", "points": int(b["locs"].shape[0]),
            "voxels": int(M), "superpoints": S, "edges": int(b["edge_u_list"].shape[0]), "passes": n,
            "ms_per_pass": round(dt * 1e3, 2), "rooms_per_s": round(1.0 / dt, 2)}


def _gpu_ms(fn, iters):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):          # lazy code-object loads and allocator growth happen in the first calls
        fn()
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def side_measurements(harness, optimizer, device, args):
    """SURVEY 8(d) companions of the headline number, each outside the timed region:
    measured device copy / triad bandwidth (next to the nominal 8 TB/s), the optimizer step on its own, and the
    PointGroup-style clustering stage (ballquery_batch_p on the GPU + bfs_cluster on the host) on a C3 batch."""
    import pointgroup_ops
    out = {}
    n = 1 << 28                                     # 1 GiB fp32 per array
    x = torch.empty(n, dtype=torch.float32, device=device).fill_(1.0)
    y = torch.empty_like(x).fill_(2.0)
    z = torch.empty_like(x)
    ms = _gpu_ms(lambda: z.copy_(x), 10)
    out["measured_copy_GBs"] = round(2 * n * 4 / (ms * 1e-3) / 1e9, 1)
    ms = _gpu_ms(lambda: torch.add(x, y, alpha=3.0, out=z), 10)
    out["measured_triad_GBs"] = round(3 * n * 4 / (ms * 1e-3) / 1e9, 1)
    del x, y, z
    out["optimizer_step_ms"] = round(_gpu_ms(optimizer.step, 5), 3)      # fused AdamW over the 11.1 M parameters

    # C3: 4 scenes, non floor/wall points, r = 0.03 m, threshold 50
    scenes = [harness.make_scene(s, room=(3.2, 2.6, 2.2), n_box=4) for s in (1, 2, 3, 4)]
    b = harness.collate(scenes)
    sem = b["superpoint"] % 20       # stand-in for the predicted class: constant per superpoint, 20 classes
    keep = torch.nonzero(sem > 1).flatten()
    coords = b["locs_float"][keep].contiguous().to(device)
    batch_idx = b["locs"][keep, 0].int().contiguous()
    offs = torch.zeros(len(scenes) + 1, dtype=torch.int32)
    offs[1:] = torch.cumsum(torch.bincount(batch_idx.long(), minlength=len(scenes)), 0).int()
    bi_d, off_d = batch_idx.to(device), offs.to(device)
    sem_keep = sem[keep].int().contiguous()
    res = {}

    def bq():
        res["idx"], res["start_len"] = pointgroup_ops.ballquery_batch_p(coords, bi_d, off_d, 0.03, 50)

    ms_bq = _gpu_ms(bq, 5)
    idx_c, sl_c = res["idx"].cpu(), res["start_len"].cpu()
    t0 = time.perf_counter()
    cl_idx, cl_off = pointgroup_ops.bfs_cluster(sem_keep, idx_c, sl_c, 50)
    ms_bfs = (time.perf_counter() - t0) * 1e3
    sem_d = sem_keep.to(device)

    def bfs_dev():
        res["cl"] = pointgroup_ops.bfs_cluster(sem_d, res["idx"], res["start_len"], 50)

    ms_bfs_dev = _gpu_ms(bfs_dev, 5)
    same = bool(torch.equal(res["cl"][0].cpu(), cl_idx) and torch.equal(res["cl"][1].cpu(), cl_off))
    # BASELINE configs C3 / C4 as side numbers (same process, after the timed region)
    out.update(other_configs(harness, device, args))

    # test-time grouping on the superpoint graph (test_scannetv2.py:281-455) on the C2 scene, synthetic predictions
    import inference
    sc = harness.bench_scene(args.scene_seed)
    sem, off, occ, size = harness.synthetic_predictions(sc, args.scene_seed)
    graph = (sc["edges"][:, 0], sc["edges"][:, 1])
    xyz = sc["xyz"].astype("float32")
    inference.clustering_in_graph("c2", xyz, sc["superpoint"], graph, sem, off, occ, size)      # warm-up
    t0 = time.perf_counter()
    conf, _, _ = inference.clustering_in_graph("c2", xyz, sc["superpoint"], graph, sem, off, occ, size)
    out["graph_cluster_stage"] = {"workload": "clustering_in_graph on the C2 scene (host arrays in, host masks out)",
                                  "points": int(len(xyz)), "superpoints": int(sc["S"]), "instances": int(len(conf)),
                                  "ms": round((time.perf_counter() - t0) * 1e3, 2)}
    # batch assembly: what ONE host thread needs per scene for the reference's collate (voxel hash + level
    # counts on the host, then H2D) against the device-side collate (raw arrays shipped, hash + counts on the GPU);
    # wall time of the calling thread, device drained on both sides
    def _wall(fn, n=5):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
            torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    nt = torch.get_num_threads()
    torch.set_num_threads(1)                   # a DataLoader worker runs torch with ONE thread (with the parent's pool a
    ms_host_collate = _wall(lambda: harness.collate([sc]))      # small CPU op costs milliseconds of wake-up)
    ms_host_total = _wall(lambda: harness.to_device(harness.collate([sc]), device))
    torch.set_num_threads(nt)
    ms_dev_total = _wall(lambda: harness.collate_device([sc], device))
    pk = harness.pack_scene(sc)
    t0 = time.perf_counter()
    for _ in range(5):
        harness.pack_scene(sc, buf=pk["buf"])          # (a loader reuses its pinned buffers)
    ms_pack = (time.perf_counter() - t0) / 5 * 1e3
    ms_packed = _wall(lambda: harness.collate_packed([pk], device))

    def _host_only(fn, n=5):          # time the calling thread spends before it could start on the next batch
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        dt = (time.perf_counter() - t0) / n * 1e3
        torch.cuda.synchronize()
        return dt
    ms_packed_host = _host_only(lambda: harness.collate_packed([pk], device))
    out["batch_assembly"] = {"workload": "one C2 scene: collate (+ H2D, per-batch CSRs) by one host thread (torch.set_num_threads(1), "
                                         "as in a DataLoader worker) vs on the device",
                             "host_collate_ms_per_scene": round(ms_host_collate, 2),
                             "host_collate_plus_to_device_ms_per_scene": round(ms_host_total, 2),
                             "device_collate_ms_per_scene": round(ms_packed, 2),
                             "device_collate_unpacked_ms_per_scene": round(ms_dev_total, 2),
                             "pack_scene_ms": round(ms_pack, 2),
                             "packed_collate_host_thread_ms_per_scene": round(ms_packed_host, 2),
                             "packed": "pack_scene: the per-sample arrays in one pinned buffer (a loader worker's output); "
                                       "collate_packed: one H2D per scene, concatenation / offsets / edge sort / voxel hash / "
                                       "level counts / CSRs on the device, device drained; host_thread: without the drain",
                             "scenes_per_s_one_host_thread": round(1e3 / ms_host_total, 1),
                             "scenes_per_s_device_collate": round(1e3 / ms_packed, 1)}
    out["propagation_stage"] = propagation_stage(harness, sc, device, args)
    out["cluster_stage"] = {"workload": "C3: 4 synthetic scenes, non floor/wall points, r=0.03 m, threshold 50",
                            "points": int(coords.shape[0]), "neighbour_pairs": int(idx_c.numel()),
                            "clusters": int(cl_off.numel() - 1), "ballquery_ms": round(ms_bq, 3),
                            "bfs_cluster_ms": round(ms_bfs_dev, 3), "bfs_cluster_host_ms": round(ms_bfs, 3),
                            "device_equals_host": same}
    return out


def propagation_stage(harness, sc, device, args):
    """a17 on the C2 scene (train_scannetv2.py:562-575 -> scannetv2_dataset.py:664-736): dense S x S fp64 affinity matrix
    from the edge list, then the per-class propagation for iterations_num = 0, 1, 2 -- the sparse chain (default) and the
    dense f64-MFMA products, ms per scene each; the dense form's TFLOP/s against the fp64 matrix peak."""
    import wsis_ops
    S = int(sc["S"])
    rng = np.random.default_rng(args.scene_seed)
    eu = torch.from_numpy(sc["edges"][:, 0].copy()).to(device)
    ev = torch.from_numpy(sc["edges"][:, 1].copy()).to(device)
    aff = torch.from_numpy(rng.random(len(sc["edges"])).astype(np.float32)).to(device)
    adjacency = np.zeros((S, S), dtype=np.int64)
    adjacency[sc["edges"][:, 0], sc["edges"][:, 1]] = 1
    adjacency = torch.from_numpy(adjacency).to(device)
    label = sc["sp_sem"]
    pred, _, _, _ = harness.synthetic_predictions(sc, args.scene_seed)
    pred = np.where(label >= 0, label, pred)
    conf = (0.5 + 0.5 * rng.random(S)).astype(np.float32)
    classes = 20
    n_present = int(len(np.unique(label[label >= 0])))
    A = wsis_ops.affinity_matrix(eu, ev, aff, S)
    res = {"workload": "C2 scene: weak_label_propagation over all present classes (fp64), affinity matrix resident",
           "superpoints": S, "edges": int(len(sc["edges"])), "classes_present": n_present,
           "labelled_superpoints": int((label >= 0).sum()), "fp64_matrix_peak_TFLOPs": FP64_MATRIX_PEAK_TF, "iterations": {}}
    ms_build = _gpu_ms(lambda: wsis_ops.affinity_matrix(eu, ev, aff, S), 5)
    res["affinity_matrix_build_ms"] = round(ms_build, 3)
    for it in (0, 1, 2):
        def sparse():
            return wsis_ops.weak_label_propagation(A, adjacency, conf, pred, label, it, classes, dense=False)

        def dense():
            return wsis_ops.weak_label_propagation(A, adjacency, conf, pred, label, it, classes, dense=True)
        f_s, s_s = sparse()
        f_d, s_d = dense()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            sparse()
        ms_s = (time.perf_counter() - t0) / 3 * 1e3
        t0 = time.perf_counter()
        dense()
        ms_d = (time.perf_counter() - t0) * 1e3
        # device time of the dense products alone (the part the f64 matrix cores run)
        T0 = torch.rand((S, S), dtype=torch.float64, device=device)
        ms_gemm = _gpu_ms(lambda: wsis_ops.dgemm(T0, T0), 3)
        flops = 2.0 * S ** 3
        row = {"sparse_ms_per_scene": round(ms_s, 3), "dense_ms_per_scene": round(ms_d, 3),
               "speedup": round(ms_d / ms_s, 1), "labels_equal": bool(np.array_equal(f_s, f_d)),
               "max_abs_score_diff": float(np.abs(s_s - s_d).max()), "propagated": int((f_s != -100).sum())}
        if it == 1:
            row["dense_product_ms"] = round(ms_gemm, 3)
            row["dense_product_TFLOPs"] = round(flops / (ms_gemm * 1e-3) / 1e12, 2)
            row["dense_product_frac_of_fp64_matrix_peak"] = round(flops / (ms_gemm * 1e-3) / 1e12 / FP64_MATRIX_PEAK_TF, 4)
        res["iterations"][str(it)] = row
    return res


def other_configs(harness, device, args):
    """C3: training step on a batch of 4 scenes (full loss); C4: eval-mode forward of a ~1 M-point room."""
    out = {}
    cfg = harness.default_cfg()
    cfg.batch_size = 4
    scenes = [harness.bench_scene(s) for s in (1, 2, 3, 4)]         # seeds 1-4, <= 250 k points each
    b = harness.to_device(harness.collate(scenes), device)
    model, crit, opt = harness.build_model(cfg, device)
    for _ in range(3):
        harness.train_step(model, crit, opt, b, cfg)
    ms = _gpu_ms(lambda: harness.train_step(model, crit, opt, b, cfg), 5)
    out["c3_batch4_train"] = {"workload": "C3: 4 C2-sized scenes (seeds 1-4) per step, fwd+bwd+AdamW, full loss",
                              "active_voxels": int(b["voxel_locs"].shape[0]), "ms_per_step": round(ms, 3),
                              "scenes_per_s": round(4e3 / ms, 2)}
    # the same roofline accounting as the headline, for the conv family at this batch size (side stream off, 2 steps)
    from spconv import ops as sp_ops
    prev = os.environ.get("WSIS_DW_STREAM")
    os.environ["WSIS_DW_STREAM"] = "0"
    sp_ops.PROFILER = sp_ops.KernelProfiler()
    for _ in range(2):
        harness.train_step(model, crit, opt, b, cfg)
    summ3 = sp_ops.PROFILER.summary()
    k = summ3.get("spconv_fwd_kernel")
    kdw, kbn = summ3.get("spconv_dw_kernel"), summ3.get("bn_op")
    sp_ops.PROFILER = None
    if prev is None:
        del os.environ["WSIS_DW_STREAM"]
    else:
        os.environ["WSIS_DW_STREAM"] = prev
    if k and k["ms"] > 0:
        gbs = k["bytes"] / (k["ms"] * 1e-3) / 1e9
        out["c3_batch4_train"]["conv_roofline"] = {"achieved_GBs": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4),
                                                   "avg_launch_us": round(k["ms"] * 1e3 / k["launches"], 2),
                                                   "launches_per_step": k["launches"] // 2}
    if kdw and kdw["ms"] > 0:      # the weight-gradient products of the same steps (main kernel + fixed-order slab sum)
        gbs = kdw["bytes"] / (kdw["ms"] * 1e-3) / 1e9
        out["c3_batch4_train"]["dw_roofline"] = {"achieved_GBs": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4),
                                                 "avg_launch_us": round(kdw["ms"] * 1e3 / kdw["launches"], 2),
                                                 "ms_per_step": round(kdw["ms"] / 2, 3),
                                                 "ms_per_step_main_kernel_only": round(kdw["ms_main"] / 2, 3),
                                                 "launches_per_step": kdw["launches"] // 2,
                                                 "tflops": round(kdw["flops"] / (kdw["ms"] * 1e-3) / 1e12, 2),
                                                 "mfma_frac_fp32": round(kdw["flops"] / (kdw["ms"] * 1e-3) / 1e12 / MFMA_FP32_PEAK_TFLOPS, 4)}
    if kbn and kbn["ms"] > 0:      # BatchNorm(+ReLU) ops of the UNet: x read + y written (backward: x, dy read, dx written, + addend)
        gbs = kbn["bytes"] / (kbn["ms"] * 1e-3) / 1e9
        out["c3_batch4_train"]["bn_roofline"] = {"achieved_GBs": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4),
                                                 "ms_per_step": round(kbn["ms"] / 2, 3), "ops_per_step": kbn["launches"] // 2,
                                                 "alg_bytes_per_step": kbn["bytes"] // 2,
                                                 "how": "event pair around all launches of each BatchNorm op of the native "
                                                        "executor (statistics finish + apply; backward finish + apply); "
                                                        "bytes = (tensors of [rows, C] read + written) x 4"}
    del b
    cfg.batch_size = 1
    big = harness.to_device(harness.collate([harness.bench_scene(5, room=(13.0, 10.0, 3.0), n_box=36)]), device)
    model.eval()

    def infer():
        with torch.no_grad():
            harness.forward_loss(model, crit, big, cfg)

    infer()
    ms = _gpu_ms(infer, 3)
    out["c4_inference"] = {"workload": "C4: eval-mode forward (+loss) of one ~1 M-point room",
                           "points": int(big["locs"].shape[0]), "active_voxels": int(big["voxel_locs"].shape[0]),
                           "ms": round(ms, 3)}
    return out


def median_over_steps(entry, steps):
    """the profiler's summary with every launch's two durations replaced by the MEDIAN of its `steps` occurrences (the
    instrumented steps issue the same sequence of products): one stall in one launch of one step -- a clock change, a
    page migration -- no longer moves a level's figure (round 5: 440 us once at level 2 of an otherwise normal run)"""
    if not entry or steps < 3 or entry["launches"] % steps:
        return entry
    per = entry["per_launch"]
    n = entry["launches"] // steps
    import statistics
    out = []
    for i in range(n):
        occ = [per[j * n + i] for j in range(steps)]
        if any(o[:6] != occ[0][:6] for o in occ):      # not the same product at the same place: leave everything as it is
            return entry
        out.append(occ[0][:6] + (statistics.median(o[6] for o in occ), statistics.median(o[7] for o in occ)))
    e = dict(entry)
    e["per_launch"] = out * steps
    e["ms_main"] = float(sum(o[6] for o in out)) * steps
    e["ms"] = float(sum(o[7] for o in out)) * steps
    return e


def per_level(records, steps):
    """roofline.per_level: the forward / dIn products of a step grouped by the pyramid level of their OUTPUT rows
    (level = rank of the row count, 0 = finest): launches, time with the finishing slab sums, algorithmic GB/s"""
    rows = sorted({r[0] for r in records}, reverse=True)
    out = []
    for lvl, m in enumerate(rows):
        sel = [r for r in records if r[0] == m]
        nbytes, flops = sum(r[4] for r in sel), sum(r[5] for r in sel)
        ms_main, ms = sum(r[6] for r in sel), sum(r[7] for r in sel)
        gbs = nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        # ceiling of every product of the level: the larger of its HBM time on the algorithmic bytes and its fp32-MFMA
        # time on the pair flops (the layers of 96 channels and more sit above the 19.6 FLOP/B ridge)
        ceil_ms = sum(max(r[4] / (HBM_PEAK_GBS * 1e9), r[5] / (MFMA_FP32_PEAK_TFLOPS * 1e12)) for r in sel) * 1e3
        out.append({"level": lvl, "rows": int(m), "launches": len(sel) // steps, "us": round(ms * 1e3 / steps, 1),
                    "ceiling_us": round(ceil_ms * 1e3 / steps, 1),
                    "frac_of_binding_ceiling": round(ceil_ms / ms, 4) if ms > 0 else 0.0,
                    "us_main_kernel_only": round(ms_main * 1e3 / steps, 1),
                    "alg_MB": round(nbytes / steps / 1e6, 1), "alg_GBs": round(gbs, 1),
                    "frac": round(gbs / HBM_PEAK_GBS, 4),
                    "mfma_frac_fp32": round(flops / (ms * 1e-3) / 1e12 / MFMA_FP32_PEAK_TFLOPS, 4) if ms > 0 else 0.0})
    return out


def cpu_baseline(full_voxels, args):
    """The oracle (torch-CPU port of the upstream gather -> mm -> scatter-add algorithm, oracle/network_ref.py)
    timed on this box's host cores in a child process with a hard time limit, on the SAME C2 scene the GPU leg
    steps on: one warm-up pass + up to --cpu-iters timed passes (the child reports after every pass, so a time-out
    keeps the passes that finished); the C1 leg (10 k-pt room, host plumbing) runs first in the same child."""
    import subprocess
    cores = os.cpu_count() or 1
    # SURVEY 8d asks for torch.set_num_threads(os.cpu_count()); on the 256-logical-core GPU box that setting is the SLOWER
    # baseline by two orders of magnitude (round 3: the 10 k-pt C1 leg 2,770 ms per pass at 256 threads against 19 ms at
    # 32 -- thousands of small index_select / mm / index_add_ calls, each fanned out over every core -- and no C2 pass
    # inside 240 s), so the baseline runs on min(cores, 32) threads and says so
    threads = max(1, min(cores, int(os.environ.get("WSIS_CPU_BASELINE_THREADS", "32"))))
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-worker", "--cpu-threads", str(threads),
           "--cpu-iters", str(args.cpu_iters), "--cpu-warmup", str(args.cpu_warmup), "--scene-seed",
           str(args.scene_seed)]
    env = dict(os.environ)
    env["HIP_VISIBLE_DEVICES"] = ""      # the child must not open the GPU
    env["OMP_NUM_THREADS"] = str(threads)
    model_name = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model_name = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    note = ""
    try:
        proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=env)
        try:
            stdout, _ = proc.communicate(timeout=args.cpu_timeout)
        except subprocess.TimeoutExpired:
            proc.kill()                      # the exact child started above
            stdout, _ = proc.communicate()
            note = f" (child stopped at the {args.cpu_timeout} s limit)"
        lines = [l for l in stdout.splitlines() if l.startswith("{")]
        r = json.loads(lines[-1])
    except Exception as e:   # noqa: BLE001 -- a failed baseline must not take the GPU result down
        return {"value": None, "unit": "scenes/s", "cores": threads, "kind": "port",
                "sample": f"cpu baseline child failed: {type(e).__name__}"}
    if "dt" not in r:
        return {"value": None, "unit": "scenes/s", "cores": threads, "kind": "port", "c1": r.get("c1"),
                "sample": f"no C2 pass finished within {args.cpu_timeout} s"}
    timed = r["iters"] > 0
    return {"value": round(1.0 / r["dt"], 5), "unit": "scenes/s", "cores": r["threads"], "kind": "port",
            "c1": r.get("c1"),
            "sample": (f"{r['iters']} timed fwd+bwd pass(es)" if timed else "warm-up passes only") +
                      f" of the {r['voxels']}-voxel C2 scene itself (the scene of the GPU leg; {full_voxels} voxels) "
                      f"after {r.get('warm_passes', 1)} warm-up pass(es) (first one {r['warm']:.1f} s), "
                      f"{r['dt']:.2f} s per pass{note}; torch-CPU oracle, "
                      f"{r['threads']} threads of {cores} logical cores, {model_name}"}


def main():
    args = parse()
    if args.cpu_worker:
        cpu_worker(args)
        return
    self_launch(args)           # --gpus N > 1 without a launcher: the ranks run in a child job, this process exits
    import harness
    import wsis_parallel as parallel
    from spconv import ops as sp_ops

    if os.environ.get("NCCL_DEBUG", "").upper() in ("", "VERSION", "WARN"):
        os.environ["NCCL_DEBUG"] = "NONE"          # nothing from RCCL on stdout next to the JSON line
    rank, local_rank, world = parallel.init_distributed()
    real_world = dist.get_world_size() if dist.is_initialized() else 1
    if real_world != max(args.gpus, 1) and os.environ.get("WSIS_FORCE_DIST", "0") != "1":
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but the process group has {real_world} rank(s)\n")
        sys.exit(2)
    assert torch.cuda.is_available(), "bench.py needs the MI355X (no CPU fallback for the product path)"
    dev_index = local_rank % torch.cuda.device_count()     # > 1 rank per GPU only in gloo control-flow tests
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)

    cfg = harness.default_cfg()
    spg = args.scenes_per_gpu if args.scenes_per_gpu > 0 else (1 if world == 1 else 4)
    cfg.batch_size = spg
    seeds = [args.scene_seed + rank * spg + i for i in range(spg)]      # C5: seeds 1..32, 4 per GPU
    if args.small:
        scenes = [harness.bench_scene(sd, room=(2.0, 1.6, 1.2), n_box=2) for sd in seeds]
    else:
        scenes = [harness.bench_scene(sd) for sd in seeds]
    batch_host = harness.collate(scenes)
    batch = harness.to_device(batch_host, device)
    model, criterion, optimizer = harness.build_model(cfg, device)
    use_dist = dist.is_initialized()
    grad_sync = parallel.GradSync(model) if use_dist else None
    sync_bn = bool(args.sync_bn) and use_dist
    if sync_bn:     # the reference's behaviour for num_gpus > 1 (train_scannetv2.py:734-736); default here: per-rank
        parallel.convert_sync_batchnorm(model)
    if os.environ.get("WSIS_BENCH_NOSYNC", "0") == "1":   # diagnostics only: process group up, no gradient exchange
        grad_sync = None
    if use_dist:   # identical initial weights on every rank
        for p in model.parameters():
            dist.broadcast(p.data, 0)
        for b in model.buffers():
            dist.broadcast(b.data, 0)

    # default: the rulebooks are built inline at the top of every forward pass (side stream).  --prefetch moves the
    # build of the NEXT step's rulebooks (same synthetic scene, rebuilt every step, inside the timed region) to a
    # helper thread; on one MI355X that was slower (17.1 -> 18.0 ms/step: the helper contends for the GIL).
    pre = harness.make_prefetcher(model) if args.prefetch else None
    if pre is not None:
        harness.prefetch_rulebooks(pre, batch)
        batch["rulebooks"] = pre.result()
    # --pipeline: the rulebooks of the NEXT step's scene (the same synthetic scene, rebuilt every step, inside the
    # timed region) are built in slices between the phases of the current step (spconv.ops.RulebookPipeline)
    pipe = harness.make_pipeline(model) if (pre is None and args.pipeline) else None
    if pipe is not None:
        harness.start_rulebooks(pipe, batch)
        batch["rulebooks"] = pipe.finish()

    def step(exchange=True):
        if not args.hoist_graphs:          # a new batch needs its segment CSRs / edge graph: part of the step
            harness.build_batch_graphs(batch)
        if pre is not None:
            harness.prefetch_rulebooks(pre, batch)
        if pipe is not None:
            harness.start_rulebooks(pipe, batch)
        out = harness.train_step(model, criterion, optimizer, batch, cfg, grad_sync=grad_sync if exchange else None,
                                 pipeline=pipe)
        if pre is not None:
            batch["rulebooks"] = pre.result()
        if pipe is not None:
            batch["rulebooks"] = pipe.finish()
        return out

    # initialisation (not part of the W warm-up steps): the first passes load code objects lazily (hipBLASLt /
    # rocPRIM / this library), grow the caching allocator to the step's footprint and ramp the clocks -- the step
    # time keeps falling for the first few hundred steps (a fixed COUNT, not a duration: every rank must run the
    # same number of steps, each holds a gradient collective)
    for _ in range(args.setup_steps):
        step()
    for _ in range(args.warmup):
        step()

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # The N = 1 point of the 1 -> 8 series is NOT the default line (one scene per step): every --gpus N > 1 line steps
    # `spg` scenes per GPU, and a step of 4 scenes runs ~1.5x the scenes/s of a step of one on the same GPU.  So every
    # line carries `scaling_baseline`: what ONE GPU does with this line's per-GPU batch and no gradient exchange --
    # measured here, in this process, on this rank's own batch (N > 1), or taken from the 4-scene side measurement of
    # the default run (N = 1).  Efficiency of a line = value / (n_gpus x scaling_baseline.scenes_per_s).
    local_base = None
    if use_dist and grad_sync is not None:
        nb = max(5, min(20, args.steps))
        # Exchange REALLY off for these steps: besides grad_sync=None, the hook the native UNet backward calls for the
        # early half-exchange is unhooked (left armed it would start an all-reduce of half the flat buffer on the
        # communication stream every step, with nobody waiting for it), and a pending early handle is drained first.
        # The ranks step on their own batches without averaging, so parameters, buffers and optimizer state are
        # snapshotted before and put back after: the timed region trains the same replicas as without this block.
        prog = getattr(model, "_native_prog", None)
        saved_hook = getattr(prog, "overlap", None) if prog is not None else None
        if prog is not None:
            prog.overlap = None
        if grad_sync._early is not None:
            grad_sync._early[0].wait()
            grad_sync._early = None
        torch.cuda.synchronize()
        snap_p = [p.detach().clone() for p in model.parameters()]
        snap_b = [b.detach().clone() for b in model.buffers()]
        snap_o = optimizer.state_dict() if hasattr(optimizer, "exp_avg") else __import__("copy").deepcopy(optimizer.state_dict())
        for _ in range(3):
            step(exchange=False)
        fence()
        tb0 = time.perf_counter()
        for _ in range(nb):
            step(exchange=False)
        torch.cuda.synchronize()
        tb = time.perf_counter() - tb0
        if use_dist:
            tt = torch.tensor([tb], dtype=torch.float64, device=device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            tb = float(tt.item())
        local_base = {"scenes_per_gpu": spg, "scenes_per_s": round(spg * nb / tb, 3), "ms_per_step": round(tb / nb * 1e3, 3),
                      "how": f"{nb} steps of this line's per-GPU batch in this process with the gradient exchange off "
                             "(no collective at all: the early half-exchange hook unhooked; slowest rank); BatchNorm mode "
                             "as in the line; parameters / optimizer state restored afterwards"}
        torch.cuda.synchronize()
        with torch.no_grad():
            for p, q in zip(model.parameters(), snap_p):
                p.copy_(q)
            for b, q in zip(model.buffers(), snap_b):
                b.copy_(q)
        optimizer.load_state_dict(snap_o)
        del snap_p, snap_b, snap_o
        if prog is not None:
            prog.overlap = saved_hook
        assert grad_sync._early is None, "an early exchange was issued during the exchange-off baseline"
        for _ in range(2):      # back to the timed configuration
            step()

    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, _ = step()
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    scenes_per_s = world * spg * args.steps / elapsed

    # ---- roofline of the dominant kernel: event-instrumented extra steps (same inputs, same process) ----
    roof = None
    extra = {}
    summ, summ_ov = {}, {}
    if args.profile_steps > 0:
        # per-kernel durations are only a property of the kernel when it runs alone: the instrumented steps keep the
        # weight-gradient launches on the main stream (in the timed region they overlap the dIn launches of
        # spconv_fwd_kernel on a side stream, which stretches both -- good for the step, meaningless per kernel).
        # EVERY rank runs these steps (they contain the gradient collective); only rank 0 instruments them.
        prev = os.environ.get("WSIS_DW_STREAM")
        os.environ["WSIS_DW_STREAM"] = "0"
        if rank == 0:
            sp_ops.PROFILER = sp_ops.KernelProfiler()
        for _ in range(args.profile_steps):
            step()
        if rank == 0:
            summ = sp_ops.PROFILER.summary()
            sp_ops.PROFILER = None
        if prev is None:
            del os.environ["WSIS_DW_STREAM"]
        else:
            os.environ["WSIS_DW_STREAM"] = prev
        # the same launches in the TIMED configuration (weight gradients on their side stream, overlapping the dIn
        # products): what a product takes while it shares the GPU -- reported next to the alone-on-the-GPU figure
        if rank == 0:
            sp_ops.PROFILER = sp_ops.KernelProfiler()
        for _ in range(args.profile_steps):
            step()
        if rank == 0:
            summ_ov = sp_ops.PROFILER.summary()
            sp_ops.PROFILER = None
    if rank == 0 and summ:
        summ = {n: median_over_steps(e, args.profile_steps) for n, e in summ.items()}
        summ_ov = {n: median_over_steps(e, args.profile_steps) for n, e in summ_ov.items()}
        k = summ.get("spconv_fwd_kernel")
        if k and k["ms"] > 0:
            gbs = k["bytes"] / (k["ms"] * 1e-3) / 1e9
            roof = {"kernel": "spconv_fwd2_kernel (spconv_in_kernel for the 6-channel input conv) INCLUDING the fixed-order slab sums that "
                              "finish a product (spconv2_reduce_kernel / spconv2_reduce_stats_kernel)",
                    "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": measured_traffic(),
                    "traffic_source": "profiles/conv_traffic.json: the builder's rocprofv3 PMC passes of this command "
                                      "(profiles/collect_traffic.py), committed -- NOT measured in this run",
                    "alg_bytes_per_launch": k["bytes"] // k["launches"],
                    "launches_per_step": k["launches"] // args.profile_steps,
                    "avg_launch_us": round(k["ms"] * 1e3 / k["launches"], 2),
                    "avg_launch_us_main_kernel_only": round(k["ms_main"] * 1e3 / k["launches"], 2),
                    "frac_main_kernel_only": round(k["bytes"] / (k["ms_main"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    "alg_bytes_per_step": k["bytes"] // args.profile_steps,
                    "tflops": round(k["flops"] / (k["ms"] * 1e-3) / 1e12, 2),
                    "mfma_frac_fp32": round(k["flops"] / (k["ms"] * 1e-3) / 1e12 / MFMA_FP32_PEAK_TFLOPS, 4),
                    "per_level": per_level(k["per_launch"], args.profile_steps),
                    "frac_overlapped": None, "avg_launch_us_overlapped": None,
                    "frac_of_binding_ceiling": round(sum(max(r[4] / (HBM_PEAK_GBS * 1e9), r[5] / (MFMA_FP32_PEAK_TFLOPS * 1e12))
                                                         for r in k["per_launch"]) * 1e3 / k["ms"], 4),
                    "measured": "start / stop HIP events of every product's launch on its stream (hipExtLaunchKernelGGL; "
                                "WSIS_PROF_EXACT=0: events recorded around it; a slab sum, where there is one, is "
                                "counted with its product) in %d extra steps (3 or more: per launch the median of its occurrences) with the dW side stream off (kernel alone "
                                "on the GPU); rocprofv3 of WSIS_DW_STREAM=0 agrees, see profiles/" % args.profile_steps}
            ko = summ_ov.get("spconv_fwd_kernel")
            if ko and ko["ms"] > 0:      # same products, timed configuration (dW on the side stream beside them)
                roof["frac_overlapped"] = round(ko["bytes"] / (ko["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                roof["avg_launch_us_overlapped"] = round(ko["ms"] * 1e3 / ko["launches"], 2)
        d = summ.get("spconv_dw_kernel")
        if d and d["ms"] > 0:
            extra["dw_kernel"] = {"achieved_GBs": round(d["bytes"] / (d["ms"] * 1e-3) / 1e9, 1),
                                  "avg_launch_us": round(d["ms"] * 1e3 / d["launches"], 2),
                                  "launches_per_step": d["launches"] // args.profile_steps,
                                  "tflops": round(d["flops"] / (d["ms"] * 1e-3) / 1e12, 2)}
        if k and d:
            extra["conv_ms_per_step"] = round((k["ms"] + d["ms"]) / args.profile_steps, 3)
        bno = summ.get("bn_op")
        if bno and bno["ms"] > 0:      # the UNet's BatchNorm(+ReLU) ops: tensors read + written x 4 bytes / event time
            extra["bn_ops"] = {"achieved_GBs": round(bno["bytes"] / (bno["ms"] * 1e-3) / 1e9, 1),
                               "frac": round(bno["bytes"] / (bno["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                               "ms_per_step": round(bno["ms"] / args.profile_steps, 3),
                               "ops_per_step": bno["launches"] // args.profile_steps}

    if rank == 0 and world == 1 and not args.no_stages:
        extra.update(side_measurements(harness, optimizer, device, args))
        if roof and extra.get("measured_copy_GBs"):       # the same bytes against what a device copy reaches here
            roof["frac_vs_measured_copy"] = round(roof["achieved"] / extra["measured_copy_GBs"], 4)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(int(batch_host["voxel_locs"].shape[0]), args) if spg == 1 else None

    # the JSON line must be the LAST thing on stdout: anything a library printed through C stdio (fully buffered on
    # a pipe) is flushed now, by every rank, before rank 0 prints
    import ctypes
    sys.stdout.flush()
    ctypes.CDLL(None).fflush(None)
    if use_dist:
        dist.barrier()
    if rank == 0:
        M = int(batch_host["voxel_locs"].shape[0])
        out = {
            "metric": "scenes/sec fwd+bwd ScanNet 2cm (~150k voxels) @1/8 GPU; HBM GB/s vs roofline",
            "value": round(scenes_per_s, 3), "unit": "scenes/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "setup_steps": args.setup_steps, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("C2: 1 synthetic ScanNet-shaped scene on 1 GPU" if (world == 1 and spg == 1) else
                                    f"C5-shaped: {spg} synthetic ScanNet-shaped scene(s) per GPU x {world} GPU(s) = "
                                    f"batch {spg * world}, scenes sharded by rank, " +
                                    ("no process group (one rank: nothing to exchange)" if not use_dist else
                                     "gradient all-reduce per step over " +
                                     ("RCCL" if dist.get_backend() == "nccl" else
                                      dist.get_backend() + " (control-flow run, ranks may share a GPU)"))) + ", 2 cm voxels, fwd+bwd+AdamW step (SubMConv3d UNet 32..160 + ECC GNN "
                                    "+ edge affinity + MultiTaskLoss); per-batch segment CSRs / edge graph and all "
                                    "rulebooks are rebuilt inside every step" +
                                    (" EXCEPT the segment CSRs / edge graph (--hoist-graphs)" if args.hoist_graphs
                                     else ""),
                       "active_voxels": M, "points": int(batch_host["locs"].shape[0]),
                       "superpoints": int(batch_host["sp_batch_offsets"][-1]),
                       "edges": int(batch_host["edge_u_list"].shape[0]), "scenes_per_gpu": spg,
                       "global_batch": spg * world, "scene_seeds_rank0": seeds,
                       "untimed_setup_steps": args.setup_steps,
                       "parallelism": f"scene-sharded dp{world}",
                       "batchnorm": ("statistics shared across the ranks (--sync-bn: wsis_parallel.convert_sync_batchnorm; the UNet "
                                     "keeps the native executor, its op list issued in parts around every layer's "
                                     "statistics exchange)" if sync_bn else
                                     "per-rank batch statistics (the reference converts to SyncBatchNorm when num_gpus > 1: "
                                     "--sync-bn)"),
                       "loss": float(loss)},
            "roofline": roof, "cpu_baseline": cpu,
        }
        out.update(extra)
        if local_base is None and world == 1 and spg == 1 and "c3_batch4_train" in extra:
            c3 = extra["c3_batch4_train"]
            local_base = {"scenes_per_gpu": 4, "scenes_per_s": c3["scenes_per_s"], "ms_per_step": c3["ms_per_step"],
                          "how": "c3_batch4_train of this run: 4 scenes per step on this GPU, no process group -- the "
                                 "N = 1 point of the 1 -> 8 series (python bench.py --gpus 1 --scenes-per-gpu 4 prints it "
                                 "as a line of its own); NOT this line's one-scene value"}
        elif local_base is None and world == 1:
            local_base = {"scenes_per_gpu": spg, "scenes_per_s": round(scenes_per_s, 3), "ms_per_step": round(ms_per_step, 3),
                          "how": "this line (one GPU, no process group)"}
        out["scaling_baseline"] = local_base
        out["per_gpu_scenes_per_s"] = round(scenes_per_s / world, 3)
        if local_base and world > 1:
            out["efficiency_vs_scaling_baseline"] = round(scenes_per_s / (world * local_base["scenes_per_s"]), 4)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
