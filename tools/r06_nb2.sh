# two output blocks per work item at level 1 (WSIS_FWD2_NB=2, read once per process): per layer, then the step
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_nb2.txt; : > $O
for nb in 1 2; do echo "== WSIS_FWD2_NB=$nb" >> $O; WSIS_FWD2_NB=$nb python tools/conv_ab.py 2>&1 | grep -E "^L[01] |estimated" >> $O; done
for rep in 1 2; do for nb in 1 2; do echo "== WSIS_FWD2_NB=$nb: one scene" >> $O; WSIS_FWD2_NB=$nb python tools/ab_step.py WSIS_X=0 WSIS_X=1 3 40 2>&1 | grep mean >> $O; done; done
cat $O
