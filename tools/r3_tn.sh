cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_timenet.py -x -q -m gpu 2>&1 | tail -2
for m in 0 1; do
  if [ $m = 1 ]; then export DIMO_TIMENET_8WAVES=1; fi
  echo "DIMO_TIMENET_8WAVES=$m"
  bash tools/kstats_all.sh $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-dropin --sustained-steps 0 --no-live-pmc 2>&1 | grep -i "timenet\|wgrad"
  python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-dropin --sustained-steps 0 --no-live-pmc | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], r['ms_per_step'])"
done
