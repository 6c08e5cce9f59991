# (needs EBATCH = 16 in csrc/ecc.hip built as libwsis_hip_new.so and the committed build as libwsis_hip_old.so)
# in-edges staged per trip of the ECC contraction kernels: 8 (old build) against 16 (new build), two default-flavour
# libraries swapped on one box: kernel times (tools/ecc_bench.py), the GNN tests, the step
cd $GRAFT_REPO_ROOT
L=3d-wsis_amd
O=gpurun_out/r06_ebatch.txt; : > $O
use() { cp $L/libwsis_hip_$1.so $L/libwsis_hip.so; }
use new
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py tests/test_network_golden.py -q -k "ecc or gnn or network or golden" > gpurun_out/r06_ebatch_tests.log 2>&1; tail -2 gpurun_out/r06_ebatch_tests.log >> $O
for v in old new; do use $v; echo "== $v: ecc_bench" >> $O; python tools/ecc_bench.py 2>&1 | grep -v amdgpu.ids >> $O; done
for rep in 1 2; do for v in old new; do use $v
  echo "== $v: one scene" >> $O; python tools/ab_step.py WSIS_X=0 WSIS_X=1 3 40 2>&1 | grep mean >> $O
done; done
use new
cat $O
