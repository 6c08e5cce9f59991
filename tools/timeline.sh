cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl0
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl0 -- python3 $GRAFT_REPO_ROOT/tools/step_phases.py > /dev/null 2>&1
f=$(find /tmp/tl0 -name "*kernel_trace.csv" | head -1)
python3 $GRAFT_REPO_ROOT/tools/timeline.py $f $GRAFT_REPO_ROOT/gpurun_out/r02_timeline.txt
tail -130 $GRAFT_REPO_ROOT/gpurun_out/r02_timeline.txt
