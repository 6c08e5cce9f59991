set -x
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py > gpurun_out/r01_bench_default.json 2> gpurun_out/r01_bench_default.err
tail -c 600 gpurun_out/r01_bench_default.json
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/st0 /tmp/st1
WSIS_DW_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st0 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --setup-steps 3 --no-cpu-baseline --no-stages > $GRAFT_REPO_ROOT/gpurun_out/r01_bench_under_rocprof.json 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --setup-steps 3 --no-cpu-baseline --no-stages > /dev/null 2>&1
cp $(find /tmp/st0 -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/r01_bench_kernel_stats.csv
cp $(find /tmp/st1 -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/r01_bench_kernel_stats_overlap.csv
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r01_stats11; cp $(find /tmp/st0 -name "*kernel_trace.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/r01_stats11/kernel_trace.csv
head -3 $GRAFT_REPO_ROOT/gpurun_out/r01_bench_kernel_stats.csv | cut -c1-200
