"""Launches of a rocprofv3 --kernel-trace CSV that lasted far longer than their kernel's norm, with their context:
   python tools/trace_outliers.py <kernel_trace.csv> [out.txt] [factor=20] [min_us=200]
For every launch whose duration exceeds `factor` x the median of its kernel name AND `min_us`: its position in the run
(index, step = optimizer launches seen so far), what ran before it on its own queue (end of the predecessor -> its start),
what overlapped it on other queues, and whether the NEXT kernel of its queue started late too (a stall of the queue or
of the clock) or right behind its end (the launch itself was long).  Written for VERDICT r05 weak #4 (one 15.8 ms launch of
a 7 us kernel in r05_c3_kernel_stats_overlap.csv whose trace was not kept)."""
import collections
import csv
import statistics
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
factor = float(sys.argv[3]) if len(sys.argv) > 3 else 20.0
min_us = float(sys.argv[4]) if len(sys.argv) > 4 else 200.0


def short(n):
    return n.replace("void ", "").replace("(anonymous namespace)::", "").replace("wsis::", "").split("(")[0][:60]


for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    r["n"] = short(r["Kernel_Name"])
    r["q"] = r.get("Queue_Id", "0")
rows.sort(key=lambda r: r["s"])
dur = collections.defaultdict(list)
for r in rows:
    dur[r["n"]].append((r["e"] - r["s"]) / 1e3)
med = {n: statistics.median(v) for n, v in dur.items()}
step = 0
per_q = collections.defaultdict(list)
for i, r in enumerate(rows):
    r["i"], r["step"] = i, step
    if "adamw" in r["n"].lower():
        step += 1
    per_q[r["q"]].append(r)
for q in per_q:
    for j, r in enumerate(per_q[q]):
        r["qj"] = j
found = 0
print(f"# {len(rows)} launches, {len(med)} kernels, {step} optimizer steps, {len(per_q)} queues; outlier = > {factor} x median "
      f"and > {min_us} us", file=out)
for r in rows:
    d = (r["e"] - r["s"]) / 1e3
    if d <= min_us or d <= factor * med[r["n"]]:
        continue
    found += 1
    qs = per_q[r["q"]]
    prev = qs[r["qj"] - 1] if r["qj"] > 0 else None
    nxt = qs[r["qj"] + 1] if r["qj"] + 1 < len(qs) else None
    print(f"\n## launch {r['i']} (step {r['step']}): {r['n']}  {d:.1f} us (median {med[r['n']]:.1f} us, "
          f"{len(dur[r['n']])} launches), queue {r['q']}, grid {r.get('Grid_Size', '?')} wg {r.get('Workgroup_Size', '?')} "
          f"lds {r.get('LDS_Block_Size', '?')} scratch {r.get('Scratch_Size', r.get('Private_Segment_Size', '?'))}", file=out)
    if prev is not None:
        print(f"   predecessor on its queue: {prev['n']} {(prev['e'] - prev['s']) / 1e3:.1f} us, ended "
              f"{(r['s'] - prev['e']) / 1e3:.1f} us before this start", file=out)
    else:
        print("   FIRST launch on its queue", file=out)
    if nxt is not None:
        print(f"   successor on its queue: {nxt['n']} starts {(nxt['s'] - r['e']) / 1e3:.1f} us after this end, lasts "
              f"{(nxt['e'] - nxt['s']) / 1e3:.1f} us (median {med[nxt['n']]:.1f})", file=out)
    first_of_name = next(x for x in rows if x["n"] == r["n"])
    print(f"   first launch of this kernel in the run: {'THIS one' if first_of_name is r else 'launch %d' % first_of_name['i']}",
          file=out)
    over = [x for x in rows if x is not r and x["s"] < r["e"] and x["e"] > r["s"]]
    byq = collections.defaultdict(list)
    for x in over:
        byq[x["q"]].append(x)
    for q, xs in byq.items():
        t = sum(min(x["e"], r["e"]) - max(x["s"], r["s"]) for x in xs) / 1e3
        names = collections.Counter(x["n"] for x in xs).most_common(4)
        print(f"   overlapping on queue {q}: {len(xs)} launches, {t:.1f} us inside the window: "
              + ", ".join(f"{c}x {n}" for n, c in names), file=out)
    if not over:
        print("   nothing else ran on the GPU during it", file=out)
    # anything else stretched in the same window?  (a clock / power event stretches everything)
    stretched = [x for x in over if (x["e"] - x["s"]) / 1e3 > 5 * med[x["n"]] and (x["e"] - x["s"]) / 1e3 > 50]
    print(f"   other launches in the window at > 5 x their median: {len(stretched)}"
          + ("" if not stretched else " (" + ", ".join(sorted({x['n'] for x in stretched})[:5]) + ")"), file=out)
print(f"\n# {found} outlier launch(es)", file=out)
