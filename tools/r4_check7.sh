set -u
cd $GRAFT_REPO_ROOT
o=gpurun_out/r4g; mkdir -p $o
export TMPDIR=/tmp
for rep in 1 2; do
for mode in "DIMO_SIDE_KNN=1" "DIMO_SIDE_KNN=0"; do
  env $mode timeout 300 python bench.py --steps 150 --warmup 10 --no-cpu-baseline --sustained-steps 0 --no-live-pmc --no-dropin 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$mode', round(d['value']), round(d['ms_per_step'], 4), 'synced', round(d['synced_step_ms']['median'],4), 'skipped', d['skipped_steps'])
" >> $o/modes.txt
done; done
bash tools/step_timeline.sh > $o/timeline.txt 2>&1
( time python -m pytest tests -x -q -m gpu ) > $o/t_all.log 2>&1
echo "rc=$?" >> $o/t_all.log
cat $o/modes.txt; tail -n 6 $o/t_all.log
