set -u
cd $GRAFT_REPO_ROOT
o=gpurun_out/r4m; mkdir -p $o
export TMPDIR=/tmp
for mode in "DIMO_TIMENET_ROWS=8" "DIMO_TIMENET_ROWS=16"; do
  env $mode timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --sustained-steps 0 --no-live-pmc --no-dropin 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['kernels_ms_per_launch']
print('$mode', round(d['value']), round(d['ms_per_step'], 4), {n: round(1e3*v,1) for n, v in k.items() if v})
" >> $o/modes.txt
done
cat $o/modes.txt
DIMO_XSTREAM=event DIMO_TIMENET_ROWS=8 bash tools/step_timeline.sh > $o/timeline8.txt 2>&1
DIMO_XSTREAM=event DIMO_TIMENET_ROWS=16 bash tools/step_timeline.sh > $o/timeline16.txt 2>&1
