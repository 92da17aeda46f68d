set -u
cd $GRAFT_REPO_ROOT
o=gpurun_out/r4aa; mkdir -p $o
export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_gpu_raster.py -x -q -m gpu -k "backward_parity or baseline_config" ) 2>&1 | tail -n 2 > $o/t.log
cat $o/t.log
for rep in 1 2 3; do
  timeout 200 python bench.py --steps 150 --warmup 10 --no-cpu-baseline --sustained-steps 0 --no-live-pmc --no-dropin 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['kernels_ms_per_launch']
print('bwd waves variant', round(d['value']), round(d['ms_per_step'], 4), {n: round(1e3*v,1) for n, v in k.items() if v and n in ('blend_fwd','blend_bwd')}, 'roof', round(d['roofline']['avg_ms'],4))
" >> $o/modes.txt
done
cat $o/modes.txt
