set -x
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_network.py tests/test_gpu_ops.py tests/test_gpu_pyramid.py tests/test_gpu_dw_hint.py -x -q -m gpu > gpurun_out/r06_tests_b.log 2>&1; tail -3 gpurun_out/r06_tests_b.log
R5="WSIS_DW2_RED=0,WSIS_DW_BATCH_REDUCE=0"; K="WSIS_DW2_RED=1,WSIS_DW_BATCH_REDUCE=0"; B="WSIS_DW2_RED=1,WSIS_DW_BATCH_REDUCE=1"
AB_SCENES=4 python tools/ab_step.py $R5 $B 6 30 > gpurun_out/r06_ab_reduce_c3.txt 2>&1
AB_SCENES=4 python tools/ab_step.py $K $B 6 30 >> gpurun_out/r06_ab_reduce_c3.txt 2>&1
cat gpurun_out/r06_ab_reduce_c3.txt
python tools/ab_step.py $R5 $B 6 40 > gpurun_out/r06_ab_reduce_c2.txt 2>&1
cat gpurun_out/r06_ab_reduce_c2.txt
