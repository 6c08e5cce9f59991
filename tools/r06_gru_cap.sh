# A/B of the GRU launches' workgroup cap (EXPERIMENTAL build: the knob is live there), four scenes and one scene per step
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_ab_grucap.txt; : > $O
echo "== four scenes per step" >> $O
AB_SCENES=4 python tools/ab_step.py WSIS_GRU_MAXB=256 WSIS_GRU_MAXB=512 6 20 2>&1 | grep mean >> $O
AB_SCENES=4 python tools/ab_step.py WSIS_GRU_MAXB=256 WSIS_GRU_MAXB=1024 4 20 2>&1 | grep mean >> $O
echo "== one scene per step" >> $O
python tools/ab_step.py WSIS_GRU_MAXB=256 WSIS_GRU_MAXB=512 6 40 2>&1 | grep mean >> $O
cat $O
