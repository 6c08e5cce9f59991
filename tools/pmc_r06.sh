: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"   # the repo root (gpurun exports it; derived from the script path otherwise)
export GRAFT_REPO_ROOT
# PMC passes on the conv kernels of the deeper levels (round 6: every level, the kernels of the final build; FETCH_SIZE / WRITE_SIZE in passes of their own):
#   /usr/local/graft/bin/gpurun --timeout 1100 -- 'bash tools/pmc_r06.sh'
# each counter group in its own rocprofv3 --pmc pass, only --kernel-trace beside it
cd /tmp && export TMPDIR=/tmp
for lvl in 0 1 2 3 4; do
  for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT" "FETCH_SIZE" "WRITE_SIZE"; do
    d=/tmp/pmc6_${lvl}_$(echo $grp | cut -d' ' -f1)
    rm -rf $d
    timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $d -- python3 $GRAFT_REPO_ROOT/tools/conv_pmc.py $lvl > /tmp/$(basename $0).log 2>&1 || tail -20 /tmp/$(basename $0).log
    echo "== level $lvl: $grp"
    python3 $GRAFT_REPO_ROOT/tools/conv_pmc.py --parse $d | grep fwd2
  done
done > $GRAFT_REPO_ROOT/gpurun_out/r06_conv_pmc.txt 2>&1
tail -40 $GRAFT_REPO_ROOT/gpurun_out/r06_conv_pmc.txt
