set -u
cd $GRAFT_REPO_ROOT
o=gpurun_out/r4q; mkdir -p $o
export TMPDIR=/tmp
( timeout 200 python -m pytest tests/test_gpu_determinism.py tests/test_gpu_losses.py -x -q -m gpu ) > $o/t.log 2>&1
echo "rc=$?" >> $o/t.log
tail -n 4 $o/t.log
timeout 1500 bash tools/collect_profiles_r04.sh core > $o/collect.log 2>&1
tail -n 12 $o/collect.log
python - <<'PY'
import json
d = json.loads(open("gpurun_out/profiles_r04/r04_bench_plain.json").read().strip().splitlines()[-1])
r = d["roofline"]
print(round(d["value"]), d["ms_per_step"], "roofline", r["frac"], r["avg_ms"], r["traffic"], r["timed_region"]["avg_ms"])
print("dropin", d.get("dropin_frames_per_s"), d.get("dropin_detail"))
print("sustained", d["sustained"]["frames_per_s"] if d.get("sustained") else None, "teacher", d.get("sustained_teacher"))
print("cpu", d.get("cpu_baseline", {}).get("value"), "s1", d.get("s1_frames_per_s"))
PY
