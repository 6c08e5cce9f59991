#!/bin/bash
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"   # the repo root (gpurun exports it; derived from the script path otherwise)
export GRAFT_REPO_ROOT
# per-kernel BN durations at several one-launch thresholds (rocprofv3 kernel stats of tools/bn_bench.py)
cd /tmp && export TMPDIR=/tmp
for r in "$@"; do
  export WSIS_BN_SMALL_ROWS=$r
  rm -rf /tmp/bnp_$r
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bnp_$r -- python3 $GRAFT_REPO_ROOT/tools/bn_bench.py > /tmp/bnp_$r.log 2>&1 || tail -5 /tmp/bnp_$r.log
  echo "== WSIS_BN_SMALL_ROWS=$r"
  python3 - <<PY
import csv,glob,collections
f=glob.glob('/tmp/bnp_$r/**/*kernel_trace.csv',recursive=True)[0]
d=collections.defaultdict(list)
for row in csv.DictReader(open(f)):
    n=row['Kernel_Name']
    if 'bn_' in n:
        d[(n[n.find('bn_'):n.find('(',n.find('bn_'))], int(row['Grid_Size_X']))].append((int(row['End_Timestamp'])-int(row['Start_Timestamp']))/1e3)
for k in sorted(d):
    v=sorted(d[k]); print(f"{k[0]:28s} grid {k[1]:7d} n {len(v):4d} median {v[len(v)//2]:6.1f} us")
PY
done
