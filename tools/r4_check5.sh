set -u
cd $GRAFT_REPO_ROOT
o=gpurun_out/r4e; mkdir -p $o
export TMPDIR=/tmp
python tools/host_probe.py 200 > $o/host_probe.txt 2>&1
DIMO_REPORT=0 DIMO_ZERO_NEXT=0 python tools/host_probe.py 200 > $o/host_probe_old.txt 2>&1
for rep in 1 2; do
for mode in "DIMO_SKIN_IN_ORDER=1" "DIMO_REPORT=0" "DIMO_ZERO_NEXT=0" "DIMO_REPORT=0 DIMO_ZERO_NEXT=0 DIMO_SKIN_IN_ORDER=0"; do
  env $mode timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --sustained-steps 0 --no-live-pmc --no-dropin 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$mode', round(d['value']), round(d['ms_per_step'], 4), 'synced', d['synced_step_ms']['median'], 'skipped', d['skipped_steps'])
" >> $o/modes.txt
done; done
( time python -m pytest tests/test_gpu_batched_render.py tests/test_gpu_bench.py -x -q -m gpu ) > $o/t.log 2>&1
echo "rc=$?" >> $o/t.log
cat $o/modes.txt; tail -n 4 $o/t.log; cat $o/host_probe.txt $o/host_probe_old.txt | grep -v amdgpu
