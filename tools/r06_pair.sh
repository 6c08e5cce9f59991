# (needs profiles/r06_pair_kernel.patch applied: the paired launch was measured and not kept)
# Paired launch of the one-wave level-0 items (WSIS_FWD2_PAIR; EXPERIMENTAL build: the knob is live): per layer, bit
# identity, then the step in-process
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_pair.txt; : > $O
python tools/conv_ab.py WSIS_FWD2_PAIR=0 WSIS_FWD2_PAIR=1 2>&1 | grep -v amdgpu.ids >> $O
echo "== one scene" >> $O
python tools/ab_step.py WSIS_FWD2_PAIR=0 WSIS_FWD2_PAIR=1 10 40 2>&1 | grep mean >> $O
echo "== one scene, dW side stream off (the products alone)" >> $O
WSIS_DW_STREAM=0 python tools/ab_step.py WSIS_FWD2_PAIR=0 WSIS_FWD2_PAIR=1 6 40 2>&1 | grep mean >> $O
cat $O
