set -u
cd $GRAFT_REPO_ROOT
o=gpurun_out/r4w; mkdir -p $o
export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_gpu_timenet.py tests/test_gpu_deform.py tests/test_gpu_determinism.py -x -q -m gpu ) > $o/t.log 2>&1; echo "rc=$?" >> $o/t.log
tail -n 4 $o/t.log
for mode in "DIMO_X=0" "DIMO_PACK_WGS=16" "DIMO_WGRAD_WGS=2048"; do
  echo "== $mode" >> $o/probe.txt
  env $mode timeout 100 python tools/timenet_probe.py 50 2>&1 | grep -v amdgpu >> $o/probe.txt
done
cat $o/probe.txt
for rep in 1 2 3; do
for mode in "DIMO_X=0" "DIMO_TIMENET_PREPARE=0" "DIMO_TIMENET_PREPARE=0 DIMO_PACK_WGS=16"; do
  env $mode timeout 200 python bench.py --steps 150 --warmup 10 --no-cpu-baseline --sustained-steps 0 --no-live-pmc --no-dropin 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['kernels_ms_per_launch']
print('$mode', round(d['value']), round(d['ms_per_step'], 4), 'skipped', d['skipped_steps'], {n: round(1e3*v,1) for n, v in k.items() if v and n in ('timenet_fwd','timenet_bwd','adam')})
" >> $o/modes.txt
done; done
cat $o/modes.txt
