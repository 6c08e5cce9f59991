# the register-gather weight-gradient kernel (spconv_dw3_kernel, round 6) against spconv_dw2_kernel: value checks, per-layer
# times, per-wave stamps, in-process A/B of the step at one and four scenes
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_dw3.txt
python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize.py tests/test_gpu_dw_hint.py tests/test_gpu_degenerate.py -x -q -m gpu > gpurun_out/r06_dw3_tests.log 2>&1; tail -3 gpurun_out/r06_dw3_tests.log
echo "== WSIS_DW3=0" > $O; WSIS_DW3=0 python tools/dw2_bench.py x >> $O 2>&1
echo "== WSIS_DW3=1 H16=1" >> $O; WSIS_DW3=1 python tools/dw2_bench.py x >> $O 2>&1
for l in 0 1 2; do WSIS_DW3=1 python tools/dw2_stamps.py $l >> $O 2>&1; done
WSIS_DW3=0 python tools/dw2_stamps.py 0 >> $O 2>&1
python tools/ab_step.py WSIS_DW3=0 WSIS_DW3=1 6 40 >> $O 2>&1
AB_SCENES=4 python tools/ab_step.py WSIS_DW3=0 WSIS_DW3=1 6 30 >> $O 2>&1
grep -v amdgpu.ids $O
