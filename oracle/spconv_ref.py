"""Oracle for spconv [UPSTREAM llijiang/spconv (fork of traveller59/spconv v1.0), un-vendored submodule
modules/lib/spconv, no commit pin; algorithm restated in SURVEY.md App. A.1].
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  parity unpinned at this boundary; checked against
dense F.conv3d / F.conv_transpose3d in tests/test_oracle_spconv.py.

Mirrors the upstream algorithm: hash-map rulebook with per-offset pair lists, then per offset
index_select -> mm -> index_add_ (call sites modules/model/sparse_unet3d.py:130,261,292)."""
import numpy as np
import torch


def _triple(v):
    if isinstance(v, (list, tuple, np.ndarray)):
        return [int(x) for x in v]
    return [int(v)] * 3


def conv_output_size(in_shape, k, s, p):
    return [(int(in_shape[j]) + 2 * p[j] - (k[j] - 1) - 1) // s[j] + 1 for j in range(3)]


def subm_pairs(indices, spatial_shape, ksize, padding):
    """-> list over K flat offsets of (in_rows int64, out_rows int64); out rows == in rows.
    Pair (i,o) under kappa iff coord_i = coord_o - pad + kappa (cross-correlation)."""
    idx = np.asarray(indices, dtype=np.int64)
    k, p = _triple(ksize), _triple(padding)
    S = [int(s) for s in spatial_shape]
    table = {tuple(r): i for i, r in reversed(list(enumerate(map(tuple, idx.tolist()))))}
    pairs = []
    for a in range(k[0]):
        for b in range(k[1]):
            for c in range(k[2]):
                ins, outs = [], []
                for o, (bb, x, y, z) in enumerate(idx.tolist()):
                    q = (x - p[0] + a, y - p[1] + b, z - p[2] + c)
                    if not (0 <= q[0] < S[0] and 0 <= q[1] < S[1] and 0 <= q[2] < S[2]):
                        continue
                    i = table.get((bb,) + q)
                    if i is not None:
                        ins.append(i)
                        outs.append(o)
                pairs.append((np.asarray(ins, np.int64), np.asarray(outs, np.int64)))
    return pairs


def down_pairs(indices, spatial_shape, ksize, stride, padding):
    """SparseConv3d rulebook -> (out_indices int64 [M_out,4] ascending linear index, out_shape, pairs)."""
    idx = np.asarray(indices, dtype=np.int64)
    k, s, p = _triple(ksize), _triple(stride), _triple(padding)
    out_shape = conv_output_size(spatial_shape, k, s, p)
    cand = []   # (lin, kappa_flat, in_row)
    for i, (bb, x, y, z) in enumerate(idx.tolist()):
        kf = 0
        for a in range(k[0]):
            for b in range(k[1]):
                for c in range(k[2]):
                    t = (x + p[0] - a, y + p[1] - b, z + p[2] - c)
                    ok = all(t[j] >= 0 and t[j] % s[j] == 0 and t[j] // s[j] < out_shape[j] for j in range(3))
                    if ok:
                        o = [t[j] // s[j] for j in range(3)]
                        lin = ((bb * out_shape[0] + o[0]) * out_shape[1] + o[1]) * out_shape[2] + o[2]
                        cand.append((lin, kf, i))
                    kf += 1
    lins = sorted(set(c[0] for c in cand))
    row_of = {l: r for r, l in enumerate(lins)}
    out_idx = np.zeros((len(lins), 4), dtype=np.int64)
    for r, l in enumerate(lins):
        z = l % out_shape[2]
        l //= out_shape[2]
        y = l % out_shape[1]
        l //= out_shape[1]
        x = l % out_shape[0]
        out_idx[r] = (l // out_shape[0], x, y, z)
    K = k[0] * k[1] * k[2]
    ins = [[] for _ in range(K)]
    outs = [[] for _ in range(K)]
    for lin, kf, i in cand:
        ins[kf].append(i)
        outs[kf].append(row_of[lin])
    pairs = [(np.asarray(a, np.int64), np.asarray(b, np.int64)) for a, b in zip(ins, outs)]
    return out_idx, out_shape, pairs


def pairs_conv(features, weight, pairs, M_out, bias=None):
    """out = sum_k index_add(out_rows, features[in_rows] @ W[k]); differentiable (torch autograd).
    weight [k0,k1,k2,Cin,Cout]."""
    Cin, Cout = weight.shape[-2], weight.shape[-1]
    W = weight.reshape(-1, Cin, Cout)
    out = torch.zeros((M_out, Cout), dtype=features.dtype, device=features.device)
    for kf, (i_rows, o_rows) in enumerate(pairs):
        if len(i_rows) == 0:
            continue
        i_t = torch.as_tensor(i_rows, device=features.device)
        o_t = torch.as_tensor(o_rows, device=features.device)
        out.index_add_(0, o_t, features.index_select(0, i_t) @ W[kf])
    if bias is not None:
        out = out + bias
    return out


def inverse_pairs(pairs):
    """SparseInverseConv3d reuses the coupled rulebook with the two sides swapped."""
    return [(o, i) for (i, o) in pairs]


def pairs_to_table(pairs, M_rows):
    """gather-table form nbr[K, M_rows] (row of the other side or -1), rows = the pairs' second member."""
    nbr = -np.ones((len(pairs), M_rows), dtype=np.int32)
    for kf, (i_rows, o_rows) in enumerate(pairs):
        nbr[kf, o_rows] = i_rows
    return nbr


# ---- vectorised rulebooks (same results as the dict versions above, used for large scenes / CPU baseline) ----
def _lin(idx, shape):
    return ((idx[:, 0] * shape[0] + idx[:, 1]) * shape[1] + idx[:, 2]) * shape[2] + idx[:, 3]


def subm_pairs_fast(indices, spatial_shape, ksize, padding):
    idx = np.asarray(indices, dtype=np.int64)
    k, p = _triple(ksize), _triple(padding)
    S = [int(s) for s in spatial_shape]
    lin = _lin(idx, S)
    order = np.argsort(lin, kind="stable")
    slin = lin[order]
    pairs = []
    rows = np.arange(idx.shape[0])
    for a in range(k[0]):
        for b in range(k[1]):
            for c in range(k[2]):
                q = idx[:, 1:] + np.array([a - p[0], b - p[1], c - p[2]])
                ok = np.all((q >= 0) & (q < np.array(S)), axis=1)
                qlin = ((idx[:, 0] * S[0] + q[:, 0]) * S[1] + q[:, 1]) * S[2] + q[:, 2]
                pos = np.searchsorted(slin, qlin)
                pos_c = np.minimum(pos, len(slin) - 1) if len(slin) else pos
                hit = ok & (pos < len(slin))
                if len(slin):
                    hit &= slin[pos_c] == qlin
                pairs.append((order[pos_c[hit]].astype(np.int64), rows[hit].astype(np.int64)))
    return pairs


def down_pairs_fast(indices, spatial_shape, ksize, stride, padding):
    idx = np.asarray(indices, dtype=np.int64)
    k, s, p = _triple(ksize), _triple(stride), _triple(padding)
    out_shape = conv_output_size(spatial_shape, k, s, p)
    sa, oa = np.array(s), np.array(out_shape)
    cand = []
    for a in range(k[0]):
        for b in range(k[1]):
            for c in range(k[2]):
                t = idx[:, 1:] + np.array([p[0] - a, p[1] - b, p[2] - c])
                ok = np.all((t >= 0) & (t % sa == 0) & (t // sa < oa), axis=1)
                o = t // sa
                olin = ((idx[:, 0] * oa[0] + o[:, 0]) * oa[1] + o[:, 1]) * oa[2] + o[:, 2]
                cand.append((np.nonzero(ok)[0], olin[ok]))
    all_lin = np.concatenate([c[1] for c in cand]) if cand else np.zeros(0, np.int64)
    lins = np.unique(all_lin)
    out_idx = np.zeros((len(lins), 4), dtype=np.int64)
    l = lins.copy()
    out_idx[:, 3] = l % oa[2]
    l //= oa[2]
    out_idx[:, 2] = l % oa[1]
    l //= oa[1]
    out_idx[:, 1] = l % oa[0]
    out_idx[:, 0] = l // oa[0]
    pairs = [(rows.astype(np.int64), np.searchsorted(lins, ol).astype(np.int64)) for rows, ol in cand]
    return out_idx, out_shape, pairs
