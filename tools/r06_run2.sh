cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ops.py tests/test_gpu_network.py tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/r06_tests_c.log 2>&1; tail -3 gpurun_out/r06_tests_c.log
O=gpurun_out/r06_dw2_h16.txt
echo "== dw2 with the 16-byte header DMA" > $O; WSIS_DW3=0 python tools/dw2_bench.py x >> $O 2>&1
python tools/dw2_stamps.py 0 >> $O 2>&1
grep -v amdgpu.ids $O | tail -40
R=r06a bash tools/refresh_profiles.sh > gpurun_out/r06a_refresh.log 2>&1
cat gpurun_out/r06a_c3_outliers.txt | head -40; cat gpurun_out/r06a_step_outliers.txt | head -20
