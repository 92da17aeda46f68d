set -u
cd $GRAFT_REPO_ROOT
o=gpurun_out/r4o; mkdir -p $o
export TMPDIR=/tmp
for mode in "DIMO_BWD_ORDER=0" "DIMO_BWD_ORDER=1" "DIMO_BWD_ORDER=2" "DIMO_BWD_ORDER=3" "DIMO_BWD_ORDER=4" "DIMO_BWD_GRID=8192" "DIMO_BWD_GRID=12288" "DIMO_BWD_GRID=4096" "DIMO_BWD_ORDER=4 DIMO_BWD_GRID=8192"; do
  env $mode timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --sustained-steps 0 --no-live-pmc --no-dropin 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$mode', round(d['value']), round(d['ms_per_step'], 4), 'serial bwd', round(d['roofline']['avg_ms'], 4), 'sched bwd', round(d['roofline']['timed_region']['avg_ms'], 4))
" >> $o/modes.txt
done
cat $o/modes.txt
