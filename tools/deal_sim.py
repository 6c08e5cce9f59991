"""What the item deal of the forward / dIn kernel (spconv2.hip: item_of) gives each CU, computed from a level's real
rulebook (GPU box: the rulebook is built on the device, the rest is numpy):
   python tools/deal_sim.py [level] [NW]
Per CU (workgroup i -> CU i % 256 while everything is resident): steps summed over its items -- of whole items, of the
busiest wave under ownership by offset index (k % NW: the kernel's rule) and under a round-robin deal of the ACTIVE
offsets (WSIS_FWD2_DEAL=1) -- for the plain (slice, block) grid of round 4 and for the snake over the weight order."""
import importlib, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); importlib.import_module("3d-wsis_amd")
import numpy as np, torch, harness
from spconv import ops
level = int(sys.argv[1]) if len(sys.argv) > 1 else 2
NW = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = 'cuda:0'
b = harness.collate([harness.make_scene(1 + i) for i in range(int(os.environ.get('CONV2_SCENES', '1')))])
idx = b['voxel_locs'].int().to(dev).contiguous(); shape = [int(s) for s in b['spatial_shape']]
for l in range(level):
    rd = ops.build_down_rulebook(idx, shape, [2] * 3, [2] * 3, [0] * 3); idx, shape = rd.out_indices, rd.out_shape
rb = ops.build_subm_rulebook(idx, shape, [3] * 3, [1] * 3)
C = 32 * (level + 1); M = idx.shape[0]
nbr_p = rb.nbr_p.cpu().numpy()                      # [27, M] in tile order
gx = (M + 31) // 32
act = np.zeros((gx, 27), dtype=bool)
for s in range(gx):
    act[s] = (nbr_p[:, s * 32:(s + 1) * 32] >= 0).any(1)
w_slice = act.sum(1)
print(f"level {level}: {M} rows, {gx} slices, C = {C}; slice weights first 24: {w_slice[:24].tolist()} ... last 8: {w_slice[-8:].tolist()}")
print(f"descending: {bool((np.diff(w_slice[:gx - 1]) <= 0).all())}")
nchunk, gy = C // 32, C // 32
own = np.stack([act[:, w::NW].sum(1) for w in range(NW)], 1)                       # [gx, NW] offsets by index
rank = np.cumsum(act, 1) - 1
deal = np.stack([(act & (rank % NW == w)).sum(1) for w in range(NW)], 1)           # active offsets round-robin
n_items = gx * gy


def loads(item_slice, per_wave, cu=None):
    cu = np.arange(n_items) % 256 if cu is None else cu
    tot = np.bincount(cu, weights=w_slice[item_slice] * nchunk, minlength=256)
    # waves of a workgroup sit on different SIMDs; the SIMD of wave w varies from workgroup to workgroup: bound it by the
    # busiest wave of every item (pessimistic) and by the mean wave (optimistic)
    busiest = np.bincount(cu, weights=per_wave[item_slice].max(1) * nchunk, minlength=256)
    w0 = np.bincount(cu, weights=per_wave[item_slice][:, 0] * nchunk, minlength=256)
    return tot, busiest, w0


i = np.arange(n_items)
plain = i % gx                                                     # round 4: workgroup (bx, by), x fastest
bnd = i // 256; c = i % 256
width = np.minimum(256, n_items - bnd * 256)
j = np.where(bnd % 2 == 1, bnd * 256 + width - 1 - c, i)
snake = j // gy
# the last (partial) band always towards the lighter CUs: after any number of full snake bands of a convex weight curve the
# low-numbered CUs carry more
nb_full = n_items // 256
j2 = np.where((bnd == nb_full) | (bnd % 2 == 1), bnd * 256 + width - 1 - c, i)
cu2 = np.where(bnd == nb_full, 255 - c, c)
snake2 = j2 // gy
# greedy bound: heaviest item to the least loaded CU
ld = np.zeros(256)
for wv in np.sort(np.repeat(w_slice, gy))[::-1] * nchunk:
    ld[np.argmin(ld)] += wv
print(f"least-loaded-first bound: whole items max {ld.max():.0f}")
_cu_override = {}
for name, sl in (("plain (slice, block) grid", plain), ("snake over the weight order", snake), ("snake, last band reversed", None)):
    for rule, pw in (("k % NW", own), ("active dealt", deal)):
        if sl is None:      # workgroup i (CU i % 256) takes item: full bands as the snake, the last band mirrored
            ii = np.arange(n_items); bb = ii // 256; cc = ii % 256
            wd = np.minimum(256, n_items - bb * 256)
            last = bb == n_items // 256
            # CU cc of the last band takes the item that the mirrored position would: position p = 255 - cc must be < width
            jj = np.where(last, bb * 256 + (255 - cc), np.where(bb % 2 == 1, bb * 256 + wd - 1 - cc, ii))
            ok = jj < n_items
            # (workgroups of the last band whose mirrored position has no item stay empty; the launch needs 256 * bands workgroups)
            tot = np.bincount(cc[ok], weights=(w_slice[jj[ok] // gy] * nchunk), minlength=256)
            busiest = np.bincount(cc[ok], weights=pw[jj[ok] // gy].max(1) * nchunk, minlength=256)
            w0 = np.bincount(cc[ok], weights=pw[jj[ok] // gy][:, 0] * nchunk, minlength=256)
            # the mirrored band leaves positions 0 .. 255 - width empty: those items must exist -> emulate with a full last band
            n_last = n_items - (n_items // 256) * 256
            cc_last = 255 - np.arange(n_last)
            jl = (n_items // 256) * 256 + np.arange(n_last)
            sel = ~last
            tot = np.bincount(np.concatenate([cc[sel], cc_last]), weights=np.concatenate([w_slice[jj[sel] // gy], w_slice[jl // gy]]) * nchunk, minlength=256)
            busiest = np.bincount(np.concatenate([cc[sel], cc_last]), weights=np.concatenate([pw[jj[sel] // gy].max(1), pw[jl // gy].max(1)]) * nchunk, minlength=256)
            w0 = np.bincount(np.concatenate([cc[sel], cc_last]), weights=np.concatenate([pw[jj[sel] // gy][:, 0], pw[jl // gy][:, 0]]) * nchunk, minlength=256)
        else:
            tot, busiest, w0 = loads(sl, pw)
        print(f"{name:28s} {rule:12s}: steps per CU  whole items max {tot.max():.0f} mean {tot.mean():.1f} | busiest waves max "
              f"{busiest.max():.0f} mean {busiest.mean():.1f} | wave 0 max {w0.max():.0f} mean {w0.mean():.1f}")

# ---- two-group snake: the R CUs that hold one item more (the partial last round) get the LIGHTEST (F + 1) R items, the
# other 256 - R CUs the heaviest F (256 - R); a snake inside each group
F, R = n_items // 256, n_items % 256
nB = 256 - R
ii = np.arange(n_items); cu = ii % 256; r = ii // 256
inB = cu >= R
b = cu - R
itemB = np.where(r % 2 == 0, r * nB + b, r * nB + (nB - 1 - b))
a = cu
itemA = F * nB + np.where(r % 2 == 0, r * R + a, r * R + (R - 1 - a))
item = np.where(inB, itemB, itemA)
assert sorted(item.tolist()) == list(range(n_items)), "not a permutation"
for rule, pw in (("k % NW", own), ("active dealt", deal)):
    tot = np.bincount(cu, weights=w_slice[item // gy] * nchunk, minlength=256)
    busiest = np.bincount(cu, weights=pw[item // gy].max(1) * nchunk, minlength=256)
    print(f"{'two-group snake':28s} {rule:12s}: steps per CU  whole items max {tot.max():.0f} mean {tot.mean():.1f} | busiest waves max {busiest.max():.0f} mean {busiest.mean():.1f}")
