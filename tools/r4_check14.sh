set -u
cd $GRAFT_REPO_ROOT
o=gpurun_out/r4n; mkdir -p $o
export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_timenet.py tests/test_gpu_deform.py -x -q -m gpu ) > $o/t.log 2>&1; echo "rc=$?" >> $o/t.log
tail -n 4 $o/t.log
for rep in 1 2; do
for mode in "DIMO_TIMENET_ROWS_FWD=16" "DIMO_TIMENET_ROWS_FWD=8" "DIMO_TIMENET_ROWS_FWD=8 DIMO_SIDE_KNN=0" "DIMO_TIMENET_ROWS=16"; do
  env $mode timeout 300 python bench.py --steps 150 --warmup 10 --no-cpu-baseline --sustained-steps 0 --no-live-pmc --no-dropin 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['kernels_ms_per_launch']
print('$mode', round(d['value']), round(d['ms_per_step'], 4), {n: round(1e3*v,1) for n, v in k.items() if v and n in ('knn','timenet_fwd','timenet_bwd','adam','deform_bwd')})
" >> $o/modes.txt
done; done
cat $o/modes.txt
