cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_collate.py tests/test_gpu_ops.py tests/test_gpu_degenerate.py tests/test_gpu_network.py -x -q -m gpu > gpurun_out/r06_tests_d.log 2>&1; tail -3 gpurun_out/r06_tests_d.log
O=gpurun_out/r06_ab_csr.txt
python tools/ab_step.py WSIS_CSR_COUNTING=0 WSIS_CSR_COUNTING=1 6 40 > $O 2>&1
AB_SCENES=4 python tools/ab_step.py WSIS_CSR_COUNTING=0 WSIS_CSR_COUNTING=1 6 30 >> $O 2>&1
grep -v amdgpu.ids $O
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/stc
B="$GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --setup-steps 3 --no-cpu-baseline --no-stages --profile-steps 0"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/stc -- python3 $B > /tmp/stc.log 2>&1 || tail -20 /tmp/stc.log
cp $(find /tmp/stc -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/r06_csr_kernel_stats.csv
grep -c rocprim $GRAFT_REPO_ROOT/gpurun_out/r06_csr_kernel_stats.csv
