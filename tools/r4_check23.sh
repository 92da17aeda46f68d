set -u
cd $GRAFT_REPO_ROOT
o=gpurun_out/r4x; mkdir -p $o
export TMPDIR=/tmp
for rep in 1 2; do
for mode in "DIMO_EXEC_STREAMS=-2 PG=2,2,2" "DIMO_EXEC_STREAMS=-2 PG=4,2,1" "DIMO_EXEC_STREAMS=-3 PG=4,2,1" "DIMO_EXEC_STREAMS=-4 PG=4,2,1" "DIMO_EXEC_STREAMS=-2 PG=1,2,4" "DIMO_EXEC_STREAMS=-2 PG=2,4,1"; do
  pg=${mode##*PG=}
  env ${mode%% PG=*} timeout 200 python bench.py --per-gpu $pg --steps 150 --warmup 10 --no-cpu-baseline --sustained-steps 0 --no-live-pmc --no-dropin 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$mode', round(d['value']), round(d['ms_per_step'], 4), 'skipped', d['skipped_steps'])
" >> $o/modes.txt
done; done
cat $o/modes.txt
