set -u
cd $GRAFT_REPO_ROOT
o=gpurun_out/r4d; mkdir -p $o
export TMPDIR=/tmp
( time python -m pytest tests/test_gpu_ops.py tests/test_gpu_losses.py tests/test_gpu_executor.py tests/test_gpu_reference_loop.py tests/test_gpu_batched_render.py tests/test_gpu_trains.py tests/test_gpu_deform.py -x -q -m gpu ) > $o/t.log 2>&1
echo "rc=$?" >> $o/t.log
( DIMO_XSTREAM=value timeout 600 python -m pytest tests/test_gpu_losses.py tests/test_gpu_trains.py -x -q -m gpu ) > $o/t_value.log 2>&1
echo "rc=$?" >> $o/t_value.log
for rep in 1 2; do
for mode in "DIMO_SKIN_IN_ORDER=1" "DIMO_SKIN_IN_ORDER=0" "DIMO_XSTREAM=value" "DIMO_XSTREAM=value DIMO_MAIN_CHAIN=1"; do
  env $mode timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --sustained-steps 0 --no-live-pmc --no-dropin 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$mode', round(d['value']), round(d['ms_per_step'], 4), 'serial bwd', round(d['roofline']['avg_ms'], 4), 'sched bwd', round(d['roofline']['timed_region']['avg_ms'], 4), 'skipped', d['skipped_steps'])
" >> $o/modes.txt
done; done
bash tools/step_timeline.sh > $o/timeline.txt 2>&1
DIMO_XSTREAM=value bash tools/step_timeline.sh > $o/timeline_value.txt 2>&1
cat $o/modes.txt; tail -n 6 $o/t.log; tail -n 4 $o/t_value.log
