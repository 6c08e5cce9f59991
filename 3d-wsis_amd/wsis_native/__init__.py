"""ctypes binding of the C ABI declared in include/wsis_hip.h.

Two libraries (built in-tree by ``3d-wsis_amd/csrc/Makefile`` / ``__graft_entry__.build()``):

* ``libwsis_host.so`` -- host-only operators (voxelization_idx, bfs_cluster); safe in forked
  DataLoader workers, never initialises HIP.
* ``libwsis_hip.so``  -- every device operator (hipcc, gfx950).

There is NO fallback: if a library is missing the accessor raises, and device operators refuse CPU
tensors.  Only ``tests/``, ``bench.py``'s cpu_baseline leg and ``__graft_entry__.smoke()`` may
import ``oracle/``.
"""
import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int32, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
_PKG = os.path.dirname(_HERE)

_host = None
_hip = None

I32 = c_int32
I64 = c_int64
F32 = c_float
F64 = c_double
P = c_void_p


class WsisError(RuntimeError):
    pass


def _load(name):
    path = os.path.join(_PKG, name)
    if not os.path.exists(path):
        raise WsisError(
            f"{name} not found at {path}: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `make -C 3d-wsis_amd/csrc` (no CPU fallback exists for this path)")
    return ctypes.CDLL(path)


_HOST_SIGS = {
    "wsis_host_version": (I32, []),
    "wsis_host_last_error": (c_char_p, []),
    "wsis_host_voxelize_idx_map": (I32, [P, I64, P, P, P]),
    "wsis_host_voxelize_idx_fill": (I32, [P, I64, P, I64, I32, P, P]),
    "wsis_host_bfs_cluster_count": (I32, [P, P, P, I64, I32, P, P, P, P]),
    "wsis_host_bfs_cluster_fill": (I32, [P, P, I64, I64, I64, P, P]),
    "wsis_host_graph_bfs": (I32, [P, P, I32, P, P, P, P, I64, P, P]),
}

_HIP_SIGS = {
    "wsis_version": (I32, []),
    "wsis_last_error": (c_char_p, []),
    "wsis_device_count": (I32, []),
    "wsis_voxelize_fwd": (I32, [P, P, P, I64, I32, I32, I32, P]),
    "wsis_voxelize_bwd": (I32, [P, P, P, I64, I32, I32, I32, P]),
    "wsis_voxelize_idx_workspace_bytes": (I64, [I64]),
    "wsis_voxelize_idx_map": (I32, [P, I64, P, P, P, I64, P]),
    "wsis_voxelize_idx_fill": (I32, [P, I64, I64, I32, P, P, P, I64, P]),
    "wsis_hash_build": (I32, [P, I64, P, P, P, I64, P]),
    "wsis_rulebook_subm": (I32, [P, I64, P, P, P, P, P, I64, P, P, P]),
    "wsis_rulebook_down_ncand": (I64, [I64, P, P, P]),
    "wsis_rulebook_down_workspace_bytes": (I64, [I64]),
    "wsis_rulebook_down_keys": (I32, [P, I64, P, P, P, P, P, P, P, P, P, I64, P]),
    "wsis_rulebook_down_fill": (I32, [P, I64, P, P, P, P, P, P, I64, P, P, P, I64, P, P, P, P, P]),
    "wsis_mask_order_workspace_bytes": (I64, [I64]),
    "wsis_mask_order": (I32, [P, I64, P, P, I64, P]),
    "wsis_tile_order_workspace_bytes": (I64, [I64]),
    "wsis_tile_order": (I32, [P, P, I64, I32, P, P, I64, P]),
    "wsis_tile_order_batch_workspace_bytes": (I64, [I64]),
    "wsis_tile_order_batch": (I32, [I32, P, P, P, I32, I32, P, P, I64, P]),
    "wsis_spconv_fwd_workspace_bytes": (I64, [I64, I32, I32, I32]),
    "wsis_spconv_fwd": (I32, [P, P, P, P, P, P, P, I64, I64, I32, I32, I32, P, I64, P]),
    "wsis_spconv_fwd_t_supported": (I32, [I32, I32, I32]),
    "wsis_spconv_fwd_t_workspace_bytes": (I64, [I64, I32, I32, I32]),
    "wsis_spconv_fwd_t": (I32, [P, P, P, P, I32, P, P, P, P, I64, I64, I32, I32, I32, P, I64, P, P]),
    "wsis_spconv_fwd_t_slabs": (I32, [I64, I32, I32, I32]),
    "wsis_bn_bwd_from_partials": (I32, [P, I64, P, P, P, P, P, P, F32, I32, P, P, P, P, I64, I32, P, I64, P, P]),
    "wsis_sync_bytes": (I64, []),
    "wsis_run_ops_part": (I32, [P, I32, P, I64, P, P, I32]),
    "wsis_bn_bwd_apply": (I32, [P, P, P, P, P, P, P, P, F32, I32, P, P, I64, I32, P]),
    "wsis_spconv_fwd_t_bn": (I32, [P, P, P, P, I32, P, P, P, P, P, P, P, F32, I32, I64, I64, I32, I32, I32, P, I64, P, P]),
    "wsis_bn_stats_finalize_workspace_bytes": (I64, [I64, I32]),
    "wsis_bn_stats_finalize": (I32, [P, I64, I64, I32, P, P, P, P, F32, P, I64, P, P]),
    "wsis_bn_stats_finalize_apply": (I32, [P, I64, I64, I32, P, P, P, P, F32, P, P, P, F32, I32, P, P, I64, P, P]),
    "wsis_weight_transpose": (I32, [P, P, I32, I32, I32, I32, P]),
    "wsis_spconv_dw_workspace_bytes": (I64, [I64, I32, I32, I32]),
    "wsis_spconv_dw": (I32, [P, P, P, P, P, I64, I64, I32, I32, I32, P, I64, P]),
    "wsis_spconv_dw_bn_supported": (I32, [I32, I32, I32]),
    "wsis_spconv_dw_bn_workspace_bytes": (I64, [I64, I32, I32, I32]),
    "wsis_spconv_dw_bn": (I32, [P, P, P, P, P, F32, I32, P, P, I32, P, P, I64, I64, I32, I32, I32, P, I64, P]),
    "wsis_rulebook_pyramid_layout": (I64, [I64, P, I32, P]),
    "wsis_rulebook_pyramid": (I32, [P, I64, P, P, I32, I32, I32, P, I64, P]),
    "wsis_rulebook_pack": (I32, [P, P, P, I64, I32, P]),
    "wsis_rulebook_pack_batch": (I32, [I32, P, P, P, P, P, P]),
    "wsis_prof_enable": (I32, [I32]),
    "wsis_prof_summary": (I32, [I32, P, P]),
    "wsis_prof_records": (I32, [I32, P, P, I64, P]),
    "wsis_bn_workspace_bytes": (I64, [I64, I32]),
    "wsis_bn_stats": (I32, [P, I64, I32, P, P, P, P, F32, P, I64, P]),
    "wsis_bn_apply": (I32, [P, P, P, P, P, F32, I32, P, I64, I32, P]),
    "wsis_bn_bwd": (I32, [P, P, P, P, P, P, F32, I32, I32, P, P, P, P, I64, I32, P, I64, P]),
    "wsis_segment_csr_workspace_bytes": (I64, [I64, I64]),
    "wsis_segment_csr": (I32, [P, I64, I64, P, P, P, I64, P]),
    "wsis_segment_csr_batch_workspace_bytes": (I64, [I64]),
    "wsis_segment_csr_batch": (I32, [I32, P, P, P, P, P, P, I64, P]),
    "wsis_segment_reduce_fwd": (I32, [P, P, P, P, P, I64, I64, I32, I32, P]),
    "wsis_segment_reduce_bwd": (I32, [P, P, P, P, P, I64, I64, I32, I32, P]),
    "wsis_gather_rows": (I32, [P, P, I32, P, I64, I32, P]),
    "wsis_edge_affinity_fwd": (I32, [P, P, P, P, P, P, P, P, F32, P, P, I64, I64, I32, P]),
    "wsis_edge_affinity_bwd": (I32, [P] * 11 + [F32] + [P] * 7 + [I64, I64, I64, I32, P]),
    "wsis_ecc_message_fwd": (I32, [P, P, P, P, P, P, I64, I64, I32, P]),
    "wsis_ecc_message_bwd": (I32, [P, P, P, P, P, P, P, P, P, I64, I64, I32, P]),
    "wsis_ecc_contract_fwd": (I32, [P, P, P, P, P, I64, I64, P]),
    "wsis_ecc_contract_bwd": (I32, [P, P, P, P, P, P, P, I64, I64, P]),
    "wsis_ecc_contract_bwd_acc": (I32, [P, P, P, P, P, P, P, I64, I64, I32, P]),
    "wsis_ecc_contract_bwd_mean": (I32, [P, P, P, P, P, P, P, P, P, I64, I64, I32, P]),
    "wsis_ecc_u_fwd": (I32, [P, P, P, I64, P]),
    "wsis_gru_cell_workspace_bytes": (I64, [I64]),
    "wsis_gru_cell_fwd": (I32, [P] * 9 + [I64, I32, P]),
    "wsis_gru_cell_bwd": (I32, [P] * 17 + [I64, I32, P, I64, P]),
    "wsis_colsum_workspace_bytes": (I64, [I64, I32]),
    "wsis_colsum": (I32, [P, I64, I32, P, P, I64, P, P]),
    "wsis_pos_enc_workspace_bytes": (I64, [I64]),
    "wsis_pos_enc_fwd": (I32, [P, P, P, P, P, P, P, P, I64, P]),
    "wsis_pos_enc_bwd": (I32, [P, P, P, P, P, P, P, P, P, P, P, I64, P, I64, P]),
    "wsis_heads_workspace_bytes": (I64, [I64, I32, I32]),
    "wsis_heads_fwd": (I32, [P, P, I64, F32, F32, I32, P, P, I64, P]),
    "wsis_heads_bwd": (I32, [P, P, I64, I32, P, P, P, I64, P]),
    "wsis_gru_cell_bwd_seq": (I32, [P] * 10 + [I64] + [P] * 8 + [I64, I32, I32, I32, I32, P, I64, P]),
    "wsis_gru_cell_fwd_mean": (I32, [P] * 12 + [I64, I32, P]),
    "wsis_affinity_dense_build": (I32, [P, P, P, I64, P, I64, P]),
    "wsis_affinity_transition": (I32, [P, P, P, P, P, I32, F32, P, I64, P]),
    "wsis_dgemm": (I32, [P, P, P, I64, I64, I64, P]),
    "wsis_affinity_colmax": (I32, [P, P, I32, P, P, I64, P]),
    "wsis_affinity_propagate_sparse_workspace_bytes": (I64, [I64, I32]),
    "wsis_affinity_propagate_sparse": (I32, [P, P, P, P, P, P, P, I32, I32, F32, I32, I64, P, P, I64, P, P, P, I64, P]),
    "wsis_ballquery_workspace_bytes": (I64, [I64]),
    "wsis_ballquery_count": (I32, [P, P, P, I64, I32, F32, P, P, P, I64, P]),
    "wsis_ballquery_fill": (I32, [P, P, P, I64, I32, F32, P, P, I64, P, I64, P]),
    "wsis_cc_same_label": (I32, [P, P, P, I64, P, P, P, P]),
    "wsis_bfs_order": (I32, [P, P, P, P, P, I64, P, P, P, P]),
    "wsis_semantic_loss_workspace_bytes": (I64, [I64]),
    "wsis_semantic_loss_fwd": (I32, [P, P, I64, I32, I64, P, P, P, I64, P]),
    "wsis_semantic_loss_bwd": (I32, [P, P, I64, I32, I64, P, P, P, P]),
    "wsis_sp_ce_loss_workspace_bytes": (I64, [I64]),
    "wsis_sp_ce_loss_fwd": (I32, [P, P, I64, I32, I64, P, P, I64, P, P]),
    "wsis_sp_ce_loss_bwd": (I32, [P, P, I64, I32, I64, P, P, P, P]),
    "wsis_loss_sum": (I32, [P, I32, ctypes.c_uint32, P, P]),
    "wsis_sp_regression_loss_fwd": (I32, [P] * 8 + [I64, I64, P, P]),
    "wsis_sp_regression_loss_bwd": (I32, [P] * 8 + [I64, I64] + [P] * 8 + [P]),
    "wsis_disc_loss_saved_floats": (I32, []),
    "wsis_disc_loss_fwd": (I32, [P, P, P, I64, I32, I32, I64, F32, F32, F32, F32, F32, P, P, P]),
    "wsis_disc_loss_bwd": (I32, [P, P, P, I64, I32, I32, I64, F32, F32, F32, F32, F32, P, P, P, P]),
    "wsis_adamw_segment_bytes": (I32, []),
    "wsis_adamw_chunk": (I32, []),
    "wsis_adamw_step": (I32, [P, P, I64, F64, F64, F64, F64, F64, P]),
    "wsis_run_ops_workspace_bytes": (I64, [P, I32]),
    "wsis_run_ops": (I32, [P, I32, P, I64, P, P]),
    "wsis_run_ops_marked": (I32, [P, I32, P, I64, P, P, I32, P]),
    "wsis_experimental": (I32, []),
    "wsis_warm_streams": (I32, [P]),
    "wsis_hint_batch_rows": (I32, [I64]),
}

# entry points of the EXPERIMENTAL build only (make -C 3d-wsis_amd/csrc EXPERIMENTAL=1; include/wsis_hip.h guards them):
# the retired designs of DESIGN.md section 8.  Bound when the loaded library reports wsis_experimental() == 1.
_HIP_SIGS_EXPERIMENTAL = {
    "wsis_spconv_fwd_f_workspace_bytes": (I64, [I64, I32, I32, I32]),
    "wsis_spconv_fwd_f": (I32, [P, P, P, P, P, I32, P, P, P, P, P, I32, I64, I64, I32, I32, I32, P, I64, P, P]),
    "wsis_deep_launches": (I64, []),
    "wsis_deep_phases": (I64, []),
}


def _bind(lib, sigs):
    for name, (res, args) in sigs.items():
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    return lib


def host():
    """libwsis_host.so (never touches the GPU runtime)."""
    global _host
    if _host is None:
        _host = _bind(_load("libwsis_host.so"), _HOST_SIGS)
    return _host


def hip():
    """libwsis_hip.so (device operators)."""
    global _hip
    if _hip is None:
        # torch first: it ships its own libamdhip64; loaded afterwards, libwsis_hip.so resolves against that copy.
        # The other order leaves two HIP runtimes in the process (this library then sees zero devices).
        import torch  # noqa: F401
        _hip = _bind(_load("libwsis_hip.so"), _HIP_SIGS)
        if _hip.wsis_experimental():
            _bind(_hip, _HIP_SIGS_EXPERIMENTAL)
    return _hip


def experimental():
    """True when the loaded libwsis_hip.so is the EXPERIMENTAL build (the retired designs of DESIGN.md section 8)"""
    return bool(hip().wsis_experimental())


def require_experimental(what):
    if not experimental():
        raise WsisError(f"{what} is part of the EXPERIMENTAL build only: make -C 3d-wsis_amd/csrc EXPERIMENTAL=1 "
                        f"(or WSIS_EXPERIMENTAL=1 python -c 'import __graft_entry__ as g; g.build()')")


def declared_symbols(experimental=False):
    hip_names = sorted(list(_HIP_SIGS) + (list(_HIP_SIGS_EXPERIMENTAL) if experimental else []))
    return sorted(_HOST_SIGS), hip_names


def check_host(status, what):
    if status != 0:
        raise WsisError(f"{what} failed ({status}): {host().wsis_host_last_error().decode()}")


def check(status, what):
    if status != 0:
        raise WsisError(f"{what} failed ({status}): {hip().wsis_last_error().decode()}")


# ---- torch helpers ------------------------------------------------------------------------------

def ptr(t):
    """data_ptr of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def stream_ptr():
    """raw hipStream_t of torch's current stream on the current device"""
    import torch
    return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())


_SYNC = {}


def sync_block(device=None):
    """zero-filled sync block (wsis_sync_bytes: 64 slots of 4 KiB) of (device, current stream): the cross-workgroup
    words of the one-launch reductions live in caller memory; every launch leaves its slot zero again, launches of one
    stream share the block, another stream gets its own (include/wsis_hip.h)"""
    import torch
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    if dev is None:
        dev = torch.cuda.current_device()
    key = (dev, torch._C._cuda_getCurrentRawStream(dev))
    t = _SYNC.get(key)
    if t is None:
        t = _SYNC[key] = torch.zeros(int(hip().wsis_sync_bytes()), dtype=torch.uint8, device=torch.device("cuda", dev))
    return t


def sync_err_words(device):
    """views of the error words (word 19 of every slot) of the sync blocks of ``device`` -- device tensors, not read here"""
    import torch
    dev = torch.device(device)
    return [t.view(torch.int32).view(-1, 1024)[:, 19] for (d, _), t in _SYNC.items() if d == dev.index]


def sync_errors():
    """slots whose bounded wait ran out (word 19 of a slot): a list of (device, stream, slot); reads the device"""
    bad = []
    for (dev, st), t in _SYNC.items():
        w = t.view(__import__("torch").int32).view(-1, 1024)[:, 19]
        for i in w.nonzero().flatten().tolist():
            bad.append((dev, st, int(i)))
    return bad


HEADS_MAX = 8          # WSIS_HEADS_MAX of include/wsis_hip.h


class Heads(ctypes.Structure):
    """``wsis_heads`` of include/wsis_hip.h: the pointer tables of wsis_heads_fwd / wsis_heads_bwd (host memory)"""
    _fields_ = [("n_heads", c_int32), ("n_lin", c_int32), ("cout", c_int32 * HEADS_MAX)] + \
               [(name, c_void_p * HEADS_MAX) for name in
                ("W1", "b1", "gamma", "beta", "W2", "b2", "running_mean", "running_var", "hidden", "out", "dout",
                 "dW1", "db1", "dgamma", "dbeta", "dW2", "db2")]


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise WsisError("this operator runs on the MI355X only: got a CPU tensor (there is no CPU fallback)")


def i32x3(v):
    """host int32[3] array from an int or a 3-sequence."""
    if isinstance(v, int):
        v = (v, v, v)
    v = [int(x) for x in v]
    assert len(v) == 3
    return (c_int32 * 3)(*v)
