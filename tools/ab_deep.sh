#!/bin/bash
# alternating same-box A/B of the resident deep-level kernel: WSIS_DEEP=0 / 1 (+ extra settings in "$@" for the B side)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for rep in 1 2; do
  for d in 0 1; do
    WSIS_DEEP=$d python bench.py --no-stages --no-cpu-baseline --steps 40 --warmup 5 --setup-steps 150 2>/dev/null | python -c "
import json,sys
l=[x for x in sys.stdin.read().split('\n') if x.startswith('{')][-1]
d=json.loads(l); r=d['roofline']
print('DEEP=$d rep$rep', 'ms/step', d['ms_per_step'], 'scenes/s', d['value'], 'frac', r['frac'], 'avg_us', r.get('avg_launch_us'), 'per_level', [(p['level'], p['us'], p['frac']) for p in r.get('per_level', [])])
"
  done
done
