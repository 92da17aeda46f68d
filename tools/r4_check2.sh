set -u
cd $GRAFT_REPO_ROOT
o=gpurun_out/r4b; mkdir -p $o
export TMPDIR=/tmp
python tools/literal_loop_profile.py --log 0 > $o/profile_nolog.txt 2>&1
python tools/literal_loop_profile.py --log 1 --steps 10 2>&1 | head -n 3 > $o/profile_log.txt
( time python -m pytest tests/test_gpu_reference_loop.py tests/test_gpu_trains.py -x -q -m gpu -s ) > $o/t_fix.log 2>&1
echo "rc=$?" >> $o/t_fix.log
for rep in 1 2; do
for mode in "DIMO_JOINT_BWD=0 DIMO_MAIN_CHAIN=1" "DIMO_JOINT_BWD=0 DIMO_MAIN_CHAIN=0" "DIMO_JOINT_BWD=1 DIMO_MAIN_CHAIN=1"; do
  env $mode python bench.py --steps 100 --warmup 10 --no-cpu-baseline --sustained-steps 0 --no-live-pmc --no-dropin 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$mode', round(d['value']), round(d['ms_per_step'], 4), 'serial bwd', round(d['roofline']['avg_ms'], 4), 'sched bwd', round(d['roofline']['timed_region']['avg_ms'], 4))
" >> $o/modes.txt
done; done
cat $o/modes.txt; tail -n 5 $o/t_fix.log; head -n 3 $o/profile_nolog.txt; cat $o/profile_log.txt
