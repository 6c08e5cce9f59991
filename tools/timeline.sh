: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"   # the repo root (gpurun exports it; derived from the script path otherwise)
export GRAFT_REPO_ROOT
# kernel timeline of one training step: bash tools/timeline.sh [tag]   (env passes through to the python process)
cd /tmp && export TMPDIR=/tmp
tag=${1:-r02_timeline}
rm -rf /tmp/tl0
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl0 -- python3 $GRAFT_REPO_ROOT/tools/step_phases.py > /tmp/$(basename $0).log 2>&1 || tail -20 /tmp/$(basename $0).log
f=$(find /tmp/tl0 -name "*kernel_trace.csv" | head -1)
python3 $GRAFT_REPO_ROOT/tools/timeline.py $f $GRAFT_REPO_ROOT/gpurun_out/$tag.txt
grep "^#" $GRAFT_REPO_ROOT/gpurun_out/$tag.txt | head -${2:-45}
