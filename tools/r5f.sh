set -u
cd $GRAFT_REPO_ROOT
o=gpurun_out/r5f; mkdir -p $o
for rep in 1 2 3; do
  for k in "20 5" "100 10" "50 10"; do
    set -- $k
    DIMO_BENCH_TRACE=1 timeout 300 python bench.py --steps $1 --warmup $2 --no-cpu-baseline --sustained-steps 0 --no-live-pmc --no-dropin 2> $o/err_$1_$rep.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('steps $1 warmup $2', round(d['value']), round(d['ms_per_step'], 4), d['synced_step_ms'])
" | tee -a $o/k.txt
    grep "step-end host times" $o/err_$1_$rep.txt | cut -c1-400 >> $o/k.txt
  done
done
