: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"   # the repo root (gpurun exports it; derived from the script path otherwise)
export GRAFT_REPO_ROOT
# Regenerates the evidence under profiles/ for round $R (default r03) on the GPU box:
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/refresh_profiles.sh'
# then copy gpurun_out/${R}_* into profiles/ (see profiles/README.md).
set -x
R=${R:-r06}
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py > gpurun_out/${R}_bench_default.json 2> gpurun_out/${R}_bench_default.err
tail -c 400 gpurun_out/${R}_bench_default.json
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/st0 /tmp/st1 /tmp/pmc_fetch /tmp/pmc_write
B="$GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --setup-steps 3 --no-cpu-baseline --no-stages"
WSIS_DW_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st0 -- python3 $B > $GRAFT_REPO_ROOT/gpurun_out/${R}_bench_under_rocprof.json 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st1 -- python3 $B > /tmp/$(basename $0).log 2>&1 || tail -20 /tmp/$(basename $0).log
cp $(find /tmp/st0 -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/${R}_bench_kernel_stats.csv
cp $(find /tmp/st1 -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/${R}_bench_kernel_stats_overlap.csv
python3 $GRAFT_REPO_ROOT/tools/timeline.py $(find /tmp/st1 -name "*kernel_trace.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/${R}_step_timeline.txt
python3 $GRAFT_REPO_ROOT/tools/trace_outliers.py $(find /tmp/st1 -name "*kernel_trace.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/${R}_step_outliers.txt
# the four-scene step (C3 / the per-GPU batch of C5), timed configuration: summary, timeline of one step, launches far beyond
# their kernel's norm with their context (round 5's summary held one 15.8 ms launch of a 7 us kernel and the trace was gone),
# and the trace itself (gzip) so that the next question can be asked of it
rm -rf /tmp/st4
B4="$GRAFT_REPO_ROOT/bench.py --scenes-per-gpu 4 --steps 3 --warmup 1 --setup-steps 3 --no-cpu-baseline --no-stages"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st4 -- python3 $B4 > /tmp/$(basename $0).log 2>&1 || tail -20 /tmp/$(basename $0).log
cp $(find /tmp/st4 -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/${R}_c3_kernel_stats_overlap.csv
T4=$(find /tmp/st4 -name "*kernel_trace.csv" | head -1)
python3 $GRAFT_REPO_ROOT/tools/timeline.py $T4 $GRAFT_REPO_ROOT/gpurun_out/${R}_c3_timeline.txt
python3 $GRAFT_REPO_ROOT/tools/trace_outliers.py $T4 $GRAFT_REPO_ROOT/gpurun_out/${R}_c3_outliers.txt
cut -d, -f1-20 $T4 | gzip -9 > $GRAFT_REPO_ROOT/gpurun_out/${R}_c3_kernel_trace.csv.gz
tail -3 $GRAFT_REPO_ROOT/gpurun_out/${R}_c3_outliers.txt
# HBM traffic of the dominant kernel family: separate counter passes (no trace domains besides --kernel-trace)
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_fetch -- python3 $B --profile-steps 0 > /tmp/$(basename $0).log 2>&1 || tail -20 /tmp/$(basename $0).log
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_write -- python3 $B --profile-steps 0 > /tmp/$(basename $0).log 2>&1 || tail -20 /tmp/$(basename $0).log
python3 $GRAFT_REPO_ROOT/profiles/collect_traffic.py /tmp/pmc_fetch /tmp/pmc_write > $GRAFT_REPO_ROOT/gpurun_out/${R}_conv_traffic.json
cp $GRAFT_REPO_ROOT/profiles/conv_traffic.json $GRAFT_REPO_ROOT/gpurun_out/${R}_conv_traffic_file.json
head -4 $GRAFT_REPO_ROOT/gpurun_out/${R}_bench_kernel_stats.csv | cut -c1-200
cat $GRAFT_REPO_ROOT/gpurun_out/${R}_conv_traffic.json
