# in-process A/Bs of the BatchNorm fusions restricted to the small levels (EXPERIMENTAL build), VERDICT r05 item 3
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_ab_bnfuse.txt
D="WSIS_FUSE_BN_APPLY=99,WSIS_FUSE_BN_FIN=0,WSIS_FUSE_BN_FIN_LVL=0"
python tools/ab_step.py $D "WSIS_FUSE_BN_APPLY=3,WSIS_FUSE_BN_FIN=0,WSIS_FUSE_BN_FIN_LVL=0" 6 40 > $O 2>&1
python tools/ab_step.py $D "WSIS_FUSE_BN_APPLY=99,WSIS_FUSE_BN_FIN=1,WSIS_FUSE_BN_FIN_LVL=3" 6 40 >> $O 2>&1
python tools/ab_step.py $D "WSIS_FUSE_BN_APPLY=3,WSIS_FUSE_BN_FIN=1,WSIS_FUSE_BN_FIN_LVL=3" 6 40 >> $O 2>&1
python tools/ab_step.py $D "WSIS_FUSE_BN_APPLY=2,WSIS_FUSE_BN_FIN=1,WSIS_FUSE_BN_FIN_LVL=2" 6 40 >> $O 2>&1
python tools/ab_step.py "WSIS_DW2_XCD=0" "WSIS_DW2_XCD=100000" 6 40 >> $O 2>&1
AB_SCENES=4 python tools/ab_step.py "WSIS_DW2_XCD=0" "WSIS_DW2_XCD=100000" 6 30 >> $O 2>&1
AB_SCENES=4 python tools/ab_step.py "WSIS_DW2_XCD=0" "WSIS_DW2_XCD=20000" 6 30 >> $O 2>&1
AB_SCENES=4 python tools/ab_step.py "WSIS_DW2_XCD=0,WSIS_DW2_XSH=6" "WSIS_DW2_XCD=100000,WSIS_DW2_XSH=4" 6 30 >> $O 2>&1
AB_SCENES=4 python tools/ab_step.py "WSIS_DW2_XCD=0,WSIS_DW2_XSH=6" "WSIS_DW2_XCD=100000,WSIS_DW2_XSH=8" 6 30 >> $O 2>&1
grep -v amdgpu.ids $O
