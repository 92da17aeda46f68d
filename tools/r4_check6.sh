set -u
cd $GRAFT_REPO_ROOT
o=gpurun_out/r4f; mkdir -p $o
export TMPDIR=/tmp
for rep in 1 2; do
for mode in "DIMO_REPORT=1" "DIMO_REPORT=0" "DIMO_REPORT=1 DIMO_XSTREAM=value" "DIMO_REPORT=0 DIMO_XSTREAM=value"; do
  env $mode timeout 300 python bench.py --steps 150 --warmup 10 --no-cpu-baseline --sustained-steps 0 --no-live-pmc --no-dropin 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$mode', round(d['value']), round(d['ms_per_step'], 4), 'synced', round(d['synced_step_ms']['median'],4), 'skipped', d['skipped_steps'])
" >> $o/modes.txt
done; done
python tools/host_probe.py 200 > $o/host_probe.txt 2>&1
cat $o/modes.txt; grep -v amdgpu $o/host_probe.txt
