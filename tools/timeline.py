"""Kernel timeline of one training step from a rocprofv3 --kernel-trace CSV:
   python tools/timeline.py <kernel_trace.csv> [out.txt]
Takes the window between the last two optimizer launches and lists every kernel (start offset us, duration us, gap to
the previous kernel end on the same queue, queue), then sums busy time / idle gaps per queue and per kernel name."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
for r in rows:
    r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
opt = [i for i, r in enumerate(rows) if "adamw" in r["Kernel_Name"].lower()]
lo, hi = opt[-2] + 1, opt[-1] + 1
win = rows[lo:hi]
t0 = win[0]["s"]
last_end = {}
busy = collections.Counter(); gaps = collections.Counter(); byname = collections.Counter(); cnt = collections.Counter()
for r in win:
    q = r.get("Queue_Id", "0")
    nm = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").replace("wsis::", "")
    nm = nm.split("(")[0][:70]
    gap = (r["s"] - last_end[q]) / 1e3 if q in last_end else 0.0
    print(f"{(r['s'] - t0) / 1e3:9.1f} {(r['e'] - r['s']) / 1e3:7.1f} gap {gap:7.1f} q{q} {nm}", file=out)
    last_end[q] = max(last_end.get(q, 0), r["e"])
    busy[q] += (r["e"] - r["s"]) / 1e3; gaps[q] += max(gap, 0.0); byname[nm] += (r["e"] - r["s"]) / 1e3; cnt[nm] += 1
span = (max(r["e"] for r in win) - t0) / 1e3
print(f"# step window {span:.1f} us, {len(win)} kernels", file=out)
for q in busy: print(f"# queue {q}: busy {busy[q]:.1f} us, gaps {gaps[q]:.1f} us", file=out)
# union busy time over all queues
iv = sorted((r["s"], r["e"]) for r in win); u = 0; cs, ce = iv[0]
for s, e in iv[1:]:
    if s > ce: u += ce - cs; cs, ce = s, e
    else: ce = max(ce, e)
u += ce - cs
print(f"# GPU busy (union of all queues) {u / 1e3:.1f} us = {u / 1e3 / span * 100:.1f} % of the window", file=out)
for nm, t in byname.most_common(60): print(f"# {t:8.1f} us {cnt[nm]:4d}x {nm}", file=out)
for q in busy:
    bq = collections.Counter(); cq = collections.Counter()
    for r in win:
        if r.get("Queue_Id", "0") == q:
            nm = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").replace("wsis::", "").split("(")[0][:70]
            bq[nm] += (r["e"] - r["s"]) / 1e3; cq[nm] += 1
    print(f"# ---- queue {q}", file=out)
    for nm, t in bq.most_common(25): print(f"#   {t:8.1f} us {cq[nm]:4d}x {nm}", file=out)
