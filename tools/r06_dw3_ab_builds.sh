# (needs the build under test as libwsis_hip_new.so and the build before it as libwsis_hip_old.so beside libwsis_hip.so)
# dw3 with the activity-balanced offset partition (new) against the build before it (old): two default-flavour libraries swapped
# between runs on one box.  Values (fp64 tests), per-layer times alone, the step at one and four scenes (A B A B).
cd $GRAFT_REPO_ROOT
L=3d-wsis_amd
O=gpurun_out/r06_dw3_bal.txt; : > $O
use() { cp $L/libwsis_hip_$1.so $L/libwsis_hip.so; }
use new
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -q -k "weight_gradient or conv or dw" > gpurun_out/r06_dw3_bal_tests.log 2>&1; tail -2 gpurun_out/r06_dw3_bal_tests.log >> $O
for v in old new; do use $v; echo "== $v: per layer, alone" >> $O; python tools/dw2_bench.py 2>&1 | grep -E "subm|1x1|down|estimated" >> $O; done
for rep in 1 2; do for v in old new; do use $v
  echo "== $v: one scene" >> $O; python tools/ab_step.py WSIS_X=0 WSIS_X=1 4 40 2>&1 | grep mean >> $O
  echo "== $v: four scenes" >> $O; AB_SCENES=4 python tools/ab_step.py WSIS_X=0 WSIS_X=1 4 20 2>&1 | grep mean >> $O
done; done
use new
cat $O
