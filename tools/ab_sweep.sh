# process-level sweep of launch-plan switches with the in-process timer of tools/ab_step.py (one line per setting)
#   SWEEP="DUMMY=0 WSIS_FWD2_NB=2 ..." [AB_SCENES=4] [AB_BLOCKS=2] [AB_STEPS=40] bash tools/ab_sweep.sh
cd $GRAFT_REPO_ROOT
for cfg in ${SWEEP:-"DUMMY=0"}; do
  echo -n "$cfg  "
  env $(echo $cfg | tr ',' ' ') python tools/ab_step.py DUMMY=0 DUMMY=1 ${AB_BLOCKS:-2} ${AB_STEPS:-40} 2>&1 | tail -1 | cut -c42-100
done
