: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"   # the repo root (gpurun exports it; derived from the script path otherwise)
export GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/st4
WSIS_DW_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st4 -- python3 $GRAFT_REPO_ROOT/bench.py --scenes-per-gpu 4 --steps 3 --warmup 1 --setup-steps 3 --no-cpu-baseline --no-stages > /tmp/$(basename $0).log 2>&1 || tail -20 /tmp/$(basename $0).log
cp $(find /tmp/st4 -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/r02_c3_kernel_stats.csv
