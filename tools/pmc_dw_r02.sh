: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"   # the repo root (gpurun exports it; derived from the script path otherwise)
export GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for lvl in 0 1; do
  for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT"; do
    d=/tmp/pmcdw_${lvl}_$(echo $grp | cut -d' ' -f1)
    rm -rf $d
    timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $d -- python3 $GRAFT_REPO_ROOT/tools/dw_pmc.py $lvl > /tmp/$(basename $0).log 2>&1 || tail -20 /tmp/$(basename $0).log
    echo "== level $lvl: $grp"
    python3 $GRAFT_REPO_ROOT/tools/dw_pmc.py --parse $d
  done
done > $GRAFT_REPO_ROOT/gpurun_out/r02_dw_pmc.txt 2>&1
grep "dw2_kernel" $GRAFT_REPO_ROOT/gpurun_out/r02_dw_pmc.txt
