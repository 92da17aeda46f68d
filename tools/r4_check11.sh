set -u
cd $GRAFT_REPO_ROOT
o=gpurun_out/r4k; mkdir -p $o
export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_gpu_timenet.py -x -q -m gpu ) > $o/t8.log 2>&1; echo "rc=$?" >> $o/t8.log
timeout 120 python tools/timenet_probe.py 50 > $o/probe8.txt 2>&1
DIMO_TIMENET_ROWS=16 timeout 120 python tools/timenet_probe.py 50 > $o/probe16.txt 2>&1
tail -n 5 $o/t8.log; grep -v amdgpu $o/probe8.txt; grep -v amdgpu $o/probe16.txt
( timeout 600 python -m pytest tests/test_gpu_losses.py tests/test_gpu_trains.py tests/test_gpu_deform.py -x -q -m gpu ) > $o/t.log 2>&1; echo "rc=$?" >> $o/t.log
tail -n 4 $o/t.log
for rep in 1 2; do
for mode in "DIMO_TIMENET_ROWS=8" "DIMO_TIMENET_ROWS=16"; do
  env $mode timeout 300 python bench.py --steps 150 --warmup 10 --no-cpu-baseline --sustained-steps 0 --no-live-pmc --no-dropin 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$mode', round(d['value']), round(d['ms_per_step'], 4), 'skipped', d['skipped_steps'])
" >> $o/modes.txt
done; done
cat $o/modes.txt
