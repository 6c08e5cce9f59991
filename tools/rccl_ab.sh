: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"
# What a live RCCL communicator costs the one-scene step (one rank, no wire): A/B of bench.py lines on one box.
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash tools/rccl_ab.sh'
cd $GRAFT_REPO_ROOT
B="python bench.py --gpus 1 --steps 30 --warmup 5 --setup-steps 150 --no-cpu-baseline --no-stages --profile-steps 0"
run() { echo "== $1"; env $2 timeout -k 10 200 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('scaling_baseline',{}).get('ms_per_step'))"; }
for spec in "$@"; do run "$spec" "$spec"; done
