# round 4, first GPU pass: new host-side work (drop-in batching, reference loop, schedule) + first numbers
set -u
cd $GRAFT_REPO_ROOT
o=gpurun_out/r4a; mkdir -p $o
export TMPDIR=/tmp
( time python -m pytest tests/test_gpu_reference_loop.py tests/test_gpu_batched_render.py -x -q -m gpu ) > $o/t_dropin.log 2>&1
echo "rc=$?" >> $o/t_dropin.log
python tools/teacher_student.py --trace-every 40 > $o/teacher.json 2> $o/teacher.err
( time python -m pytest tests/test_gpu_executor.py tests/test_gpu_losses.py tests/test_gpu_bench.py -x -q -m gpu ) > $o/t_exec.log 2>&1
echo "rc=$?" >> $o/t_exec.log
python tools/determinism_probe.py > $o/determinism.json 2> $o/determinism.err
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --sustained-steps 0 --no-live-pmc > $o/bench_default.json 2> $o/bench_default.err
DIMO_JOINT_BWD=1 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --sustained-steps 0 --no-live-pmc --no-dropin > $o/bench_joint.json 2> $o/bench_joint.err
DIMO_MAIN_CHAIN=0 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --sustained-steps 0 --no-live-pmc --no-dropin > $o/bench_nomain.json 2> $o/bench_nomain.err
( time python -m pytest tests/test_gpu_trains.py -x -q -m gpu -s ) > $o/t_trains.log 2>&1
echo "rc=$?" >> $o/t_trains.log
tail -3 $o/t_dropin.log $o/t_exec.log $o/t_trains.log
python - <<'PY'
import json
for n in ("bench_default", "bench_joint", "bench_nomain"):
    try:
        d = json.loads(open(f"gpurun_out/r4a/{n}.json").read().strip().splitlines()[-1])
        print(n, round(d["value"]), d["ms_per_step"], d["roofline"]["avg_ms"], d["roofline"].get("timed_region", {}).get("avg_ms"), d.get("dropin_frames_per_s"), d.get("dropin_detail"))
    except Exception as e:
        print(n, "failed", e)
PY
cat $o/teacher.json | head -c 1500
