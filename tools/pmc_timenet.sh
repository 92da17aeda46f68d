# SQ counters of the TimeNet probe's kernels: bash tools/pmc_timenet.sh [pattern ...]
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE"; do
  n=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rm -rf /tmp/pmc_$n
  rocprofv3 --pmc $grp --output-format csv -d /tmp/pmc_$n -o p -- python $GRAFT_REPO_ROOT/tools/timenet_probe.py 10 > /dev/null 2>&1
  f=$(find /tmp/pmc_$n -name "*counter_collection.csv" | head -1)
  python $GRAFT_REPO_ROOT/tools/pmc_sq_summarise.py $f "$@"
done
