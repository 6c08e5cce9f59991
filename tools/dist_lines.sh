: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"
# The distributed lines of a round (one GPU box): one RCCL rank (1 scene, 4 scenes, 4 scenes with --sync-bn), two gloo
# ranks on the one GPU (per-rank statistics, --sync-bn), and the aten / host trace with and without a process group.
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash tools/dist_lines.sh'
set -x
R=${R:-r06}
cd $GRAFT_REPO_ROOT
O=gpurun_out
WSIS_FORCE_DIST=1 timeout -k 10 240 python bench.py --gpus 1 --steps 40 --warmup 5 --no-cpu-baseline --no-stages > $O/${R}_bench_rccl1.json 2> $O/${R}_bench_rccl1.err
tail -c 300 $O/${R}_bench_rccl1.json
WSIS_FORCE_DIST=1 timeout -k 10 240 python bench.py --gpus 1 --scenes-per-gpu 4 --steps 20 --warmup 5 --setup-steps 150 --no-cpu-baseline --no-stages > $O/${R}_bench_rccl1_spg4.json 2> $O/${R}_bench_rccl1_spg4.err
tail -c 300 $O/${R}_bench_rccl1_spg4.json
WSIS_FORCE_DIST=1 timeout -k 10 240 python bench.py --gpus 1 --scenes-per-gpu 4 --sync-bn --steps 20 --warmup 5 --setup-steps 150 --no-cpu-baseline --no-stages > $O/${R}_bench_rccl1_syncbn.json 2> $O/${R}_bench_rccl1_syncbn.err
tail -c 300 $O/${R}_bench_rccl1_syncbn.json
WSIS_DIST_BACKEND=gloo timeout -k 10 240 python bench.py --gpus 2 --small --steps 5 --warmup 2 --setup-steps 5 --no-cpu-baseline --no-stages > $O/${R}_bench_gloo2_small.json 2> $O/${R}_bench_gloo2_small.err
tail -c 300 $O/${R}_bench_gloo2_small.json
WSIS_DIST_BACKEND=gloo timeout -k 10 240 python bench.py --gpus 2 --small --sync-bn --steps 5 --warmup 2 --setup-steps 5 --no-cpu-baseline --no-stages > $O/${R}_bench_gloo2_syncbn_small.json 2> $O/${R}_bench_gloo2_syncbn_small.err
# (the gloo library prints its own connection lines on stdout: keep the JSON line alone in the .json, the rest in the .err)
for f in $O/${R}_bench_gloo2_small $O/${R}_bench_gloo2_syncbn_small; do
  [ -f $f.json ] && { grep -v '^{' $f.json >> $f.err; grep '^{' $f.json | tail -1 > $f.json.tmp; mv $f.json.tmp $f.json; }
done
tail -c 300 $O/${R}_bench_gloo2_syncbn_small.json
timeout -k 10 200 python tools/aten_trace.py > $O/${R}_aten_trace.txt 2>&1
AB_DIST=1 timeout -k 10 200 python tools/aten_trace.py > $O/${R}_aten_trace_dist.txt 2>&1
tail -30 $O/${R}_aten_trace.txt
