# usage: bash tools/pmc_tcc.sh <kernel-name-pattern> ...   (L2 <-> memory request counters of the bench workload's kernels)
cd /tmp && export TMPDIR=/tmp
for grp in "TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum" "TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum" "TCC_REQ_sum TCC_HIT_sum" "TCC_MISS_sum TCC_WRITE_sum"; do
  n=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --output-format csv -d /tmp/pmc_$n -o p -- python $GRAFT_REPO_ROOT/tools/pmc_probe.py > /dev/null 2>&1
  f=$(find /tmp/pmc_$n -name "*counter_collection.csv" | head -1)
  python $GRAFT_REPO_ROOT/tools/pmc_sq_summarise.py $f "$@"
done
