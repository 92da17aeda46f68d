cd $GRAFT_REPO_ROOT
export DIMO_EXEC_STREAMS=0
for d in 0 1 2 3 4 9; do
  export DIMO_BIN_DBG=$d
  echo "DIMO_BIN_DBG=$d"
  bash tools/kstats_all.sh $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-dropin --sustained-steps 0 2>&1 | grep -i "bucket_sort_batched\|level2_batched\|level1_"
done
