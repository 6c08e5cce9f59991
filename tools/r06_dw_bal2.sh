cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_dw_bal2.txt; : > $O
for s in "WSIS_DW_BAL=0 WSIS_DW2_PFLOOR=0" "WSIS_DW_BAL=1 WSIS_DW2_PFLOOR=0" "WSIS_DW_BAL=1 WSIS_DW2_PFLOOR=1"; do
  echo "== $s: per layer, alone (dw3)" >> $O; env $s python tools/dw2_bench.py 2>&1 | grep -E "subm.*x[48]:|estimated" | tail -6 >> $O
done
echo "== one scene, WSIS_DW_BAL=1" >> $O
WSIS_DW_BAL=1 python tools/ab_step.py WSIS_DW2_PFLOOR=0 WSIS_DW2_PFLOOR=1 10 40 2>&1 | grep mean >> $O
echo "== one scene, both" >> $O
python tools/ab_step.py WSIS_DW_BAL=0,WSIS_DW2_PFLOOR=0 WSIS_DW_BAL=1,WSIS_DW2_PFLOOR=1 10 40 2>&1 | grep mean >> $O
cat $O
