# round 6, third GPU call: adaptive chains on the GPU -- parity (both regimes), determinism, then both regimes through bench.py
set -u
cd $GRAFT_REPO_ROOT
o=gpurun_out/r6d; mkdir -p $o
export TMPDIR=/tmp
( time timeout 2400 python -m pytest -q -m gpu -x tests/test_gpu_executor.py tests/test_gpu_raster.py tests/test_gpu_deform.py tests/test_gpu_determinism.py tests/test_gpu_kat.py tests/test_gpu_trains.py ) > $o/pytest.log 2>&1
echo "rc=$?" >> $o/pytest.log; tail -n 8 $o/pytest.log
timeout 900 python bench.py --steps 20 --warmup 5 --no-dropin --no-cpu-baseline --sustained-steps 0 > $o/bench.json 2> $o/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r6d/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "synced", d.get("synced_frames_per_s"), "gap", d.get("value_behind_idle_gap"))
print(json.dumps(d["config"].get("regimes"), indent=1))
PY
