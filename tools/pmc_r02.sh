: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"   # the repo root (gpurun exports it; derived from the script path otherwise)
export GRAFT_REPO_ROOT
# PMC passes on the round-2 conv kernels (forward SubM of one level, 10 launches each of the round-1 and round-2 kernel):
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/pmc_r02.sh'
cd /tmp && export TMPDIR=/tmp
for lvl in 0 1; do
  for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT"; do
    d=/tmp/pmc_${lvl}_$(echo $grp | cut -d' ' -f1)
    rm -rf $d
    timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $d -- python3 $GRAFT_REPO_ROOT/tools/conv_pmc.py $lvl > /tmp/$(basename $0).log 2>&1 || tail -20 /tmp/$(basename $0).log
    echo "== level $lvl: $grp"
    python3 $GRAFT_REPO_ROOT/tools/conv_pmc.py --parse $d
  done
done > $GRAFT_REPO_ROOT/gpurun_out/r02_conv_pmc.txt 2>&1
tail -60 $GRAFT_REPO_ROOT/gpurun_out/r02_conv_pmc.txt
