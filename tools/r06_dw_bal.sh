# The activity-balanced offset partition of the weight-gradient kernel (WSIS_DW_BAL, EXPERIMENTAL build: live) and the
# round-down rule of its workgroup count (WSIS_DW2_PFLOOR): per layer alone, then the step in-process at one / four scenes
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_dw_bal.txt; : > $O
for s in "WSIS_DW_BAL=0 WSIS_DW2_PFLOOR=0" "WSIS_DW_BAL=1 WSIS_DW2_PFLOOR=0" "WSIS_DW_BAL=0 WSIS_DW2_PFLOOR=1" "WSIS_DW_BAL=1 WSIS_DW2_PFLOOR=1"; do
  echo "== $s: per layer, alone" >> $O; env $s python tools/dw2_bench.py 2>&1 | grep -E "subm.*x[48]:|estimated" | head -6 >> $O
done
echo "== one scene" >> $O
python tools/ab_step.py WSIS_DW_BAL=0 WSIS_DW_BAL=1 8 40 2>&1 | grep mean >> $O
WSIS_DW_BAL=1 python tools/ab_step.py WSIS_DW2_PFLOOR=0 WSIS_DW2_PFLOOR=1 6 40 2>&1 | grep mean >> $O
echo "== four scenes" >> $O
AB_SCENES=4 python tools/ab_step.py WSIS_DW_BAL=0 WSIS_DW_BAL=1 8 20 2>&1 | grep mean >> $O
AB_SCENES=4 WSIS_DW_BAL=1 python tools/ab_step.py WSIS_DW2_PFLOOR=0 WSIS_DW2_PFLOOR=1 6 20 2>&1 | grep mean >> $O
echo "== two scenes" >> $O
AB_SCENES=2 python tools/ab_step.py WSIS_DW_BAL=0 WSIS_DW_BAL=1 6 30 2>&1 | grep mean >> $O
cat $O
