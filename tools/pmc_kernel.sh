# usage: bash tools/pmc_kernel.sh <kernel-name-pattern> [more patterns]   (SQ counters of the bench workload's kernels)
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD" "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR"; do
  n=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --output-format csv -d /tmp/pmc_$n -o p -- python $GRAFT_REPO_ROOT/tools/pmc_probe.py > /dev/null 2>&1
  f=$(find /tmp/pmc_$n -name "*counter_collection.csv" | head -1)
  python $GRAFT_REPO_ROOT/tools/pmc_sq_summarise.py $f "$@"
done
