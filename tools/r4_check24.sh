set -u
cd $GRAFT_REPO_ROOT
o=gpurun_out/r4y; mkdir -p $o
export TMPDIR=/tmp
for rep in 1 2; do
for mode in "DIMO_X=0" "DIMO_TIMENET_ROWS_FWD=8 DIMO_KNN_WGS=128" "DIMO_TIMENET_ROWS_FWD=8 DIMO_KNN_WGS=256" "DIMO_TIMENET_ROWS_FWD=8 DIMO_KNN_WGS=384" "DIMO_KNN_WGS=256" "DIMO_KNN_WGS=512"; do
  env $mode timeout 200 python bench.py --steps 150 --warmup 10 --no-cpu-baseline --sustained-steps 0 --no-live-pmc --no-dropin 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['kernels_ms_per_launch']
print('$mode', round(d['value']), round(d['ms_per_step'], 4), {n: round(1e3*v,1) for n, v in k.items() if v and n in ('knn','timenet_fwd','timenet_bwd')})
" >> $o/modes.txt
done; done
cat $o/modes.txt
