set -u
cd $GRAFT_REPO_ROOT
o=gpurun_out/r4h; mkdir -p $o
export TMPDIR=/tmp
( time python -m pytest tests/test_gpu_losses.py tests/test_gpu_deform.py tests/test_gpu_trains.py tests/test_gpu_ops.py tests/test_gpu_bench.py -x -q -m gpu ) > $o/t.log 2>&1
echo "rc=$?" >> $o/t.log
for rep in 1 2; do
for mode in "DIMO_SPLIT_ADAM=1" "DIMO_SPLIT_ADAM=0"; do
  env $mode timeout 300 python bench.py --steps 150 --warmup 10 --no-cpu-baseline --sustained-steps 0 --no-live-pmc --no-dropin 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$mode', round(d['value']), round(d['ms_per_step'], 4), 'synced', round(d['synced_step_ms']['median'],4), 'skipped', d['skipped_steps'])
" >> $o/modes.txt
done; done
cat $o/modes.txt; tail -n 6 $o/t.log
