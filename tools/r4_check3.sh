set -u
cd $GRAFT_REPO_ROOT
o=gpurun_out/r4c; mkdir -p $o
export TMPDIR=/tmp
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/mfma4_rate.hip -o /tmp/mfma4_rate 2>/dev/null && /tmp/mfma4_rate > $o/mfma4_rate.txt 2>&1
( time python -m pytest tests/test_gpu_reference_loop.py tests/test_gpu_batched_render.py tests/test_gpu_trains.py tests/test_gpu_bench.py -x -q -m gpu ) > $o/t.log 2>&1
echo "rc=$?" >> $o/t.log
python tools/literal_loop_profile.py --log 0 > $o/profile_nolog.txt 2>&1
python bench.py --steps 50 --warmup 5 --no-cpu-baseline --sustained-steps 0 --no-live-pmc > $o/bench.json 2> $o/bench.err
bash tools/step_timeline.sh > $o/timeline.txt 2>&1
cat $o/mfma4_rate.txt; tail -n 6 $o/t.log; head -n 6 $o/profile_nolog.txt | tail -n 2
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4c/bench.json").read().strip().splitlines()[-1])
print(round(d["value"]), d["ms_per_step"], d["roofline"]["avg_ms"], d["roofline"]["timed_region"]["avg_ms"], d.get("dropin_frames_per_s"), d.get("dropin_detail"))
PY
