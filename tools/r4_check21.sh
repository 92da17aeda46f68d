set -u
cd $GRAFT_REPO_ROOT
o=gpurun_out/r4v; mkdir -p $o
export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_gpu_raster.py tests/test_gpu_executor.py tests/test_gpu_kat.py -x -q -m gpu ) > $o/t.log 2>&1; echo "rc=$?" >> $o/t.log
tail -n 6 $o/t.log
for rep in 1 2; do
for mode in "DIMO_X=0" "DIMO_FWD_MFMA=0" "DIMO_WGRAD_LINEAR=1" "DIMO_WGRAD_LINEAR=1 DIMO_WGRAD_WGS=1536"; do
  env $mode timeout 200 python bench.py --steps 150 --warmup 10 --no-cpu-baseline --sustained-steps 0 --no-live-pmc --no-dropin 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['kernels_ms_per_launch']
print('$mode', round(d['value']), round(d['ms_per_step'], 4), 'skipped', d['skipped_steps'], {n: round(1e3*v,1) for n, v in k.items() if v and n in ('blend_fwd','blend_bwd','timenet_bwd')}, 'roof', d['roofline']['avg_ms'])
" >> $o/modes.txt
done; done
cat $o/modes.txt
