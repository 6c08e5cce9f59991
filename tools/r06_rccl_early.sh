# one RCCL rank, one scene per step: the early fork of the weight gradients (WSIS_DW_EARLY) with a live communicator
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_rccl_early.txt; : > $O
for e in 0 1 0 1; do
  WSIS_DW_EARLY=$e WSIS_FORCE_DIST=1 timeout -k 10 240 python bench.py --gpus 1 --steps 40 --warmup 5 --no-cpu-baseline --no-stages --profile-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('EARLY=$e rccl1', d['value'], d['ms_per_step'], d['scaling_baseline']['scenes_per_s'])" >> $O
done
AB_DIST=1 python tools/ab_step.py WSIS_DW_EARLY=0 WSIS_DW_EARLY=1 6 40 >> $O 2>&1
grep -v amdgpu.ids $O
