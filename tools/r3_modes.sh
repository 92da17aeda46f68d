cd $GRAFT_REPO_ROOT
for mode in "default" "DIMO_MAIN_CHAIN=0" "DIMO_EXEC_STREAMS=0" "DIMO_JOINT_BWD=0" "DIMO_JOINT_LOSSES=1"; do
  for rep in 1 2; do
    if [ "$mode" = "default" ]; then v=$(python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-dropin --no-live-pmc --sustained-steps 0 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(r['value']), round(r['ms_per_step'],4))");
    else v=$(env $mode python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-dropin --no-live-pmc --sustained-steps 0 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(r['value']), round(r['ms_per_step'],4))"); fi
    echo "$mode: $v"
  done
done
