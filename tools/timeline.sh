# kernel timeline of one training step: bash tools/timeline.sh [tag]   (env passes through to the python process)
cd /tmp && export TMPDIR=/tmp
tag=${1:-r02_timeline}
rm -rf /tmp/tl0
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl0 -- python3 $GRAFT_REPO_ROOT/tools/step_phases.py > /dev/null 2>&1
f=$(find /tmp/tl0 -name "*kernel_trace.csv" | head -1)
python3 $GRAFT_REPO_ROOT/tools/timeline.py $f $GRAFT_REPO_ROOT/gpurun_out/$tag.txt
grep "^#" $GRAFT_REPO_ROOT/gpurun_out/$tag.txt | head -${2:-45}
