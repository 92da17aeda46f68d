# End-of-round verification + the profiles that changed with the last kernels (run on the GPU box).
set -u
R=$GRAFT_REPO_ROOT
cd $R
o=gpurun_out/r4final2; mkdir -p $o
export TMPDIR=/tmp
( time timeout 700 python -m pytest tests -x -q -m gpu ) > $o/pytest.log 2>&1; echo "rc=$?" >> $o/pytest.log
tail -n 6 $o/pytest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $o/smoke.log 2>&1; tail -n 1 $o/smoke.log
out=$R/gpurun_out/profiles_r04b; mkdir -p $out
cd /tmp
rm -rf /tmp/p1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dropin --sustained-steps 0 --no-live-pmc > /tmp/bench_prof.log 2>&1
grep "^{\"metric\"" /tmp/bench_prof.log | tail -1 > $out/r04_bench_under_rocprof.json
cp $(find /tmp/p1 -name "*kernel_stats.csv" | head -1) $out/r04_kernel_stats_bench_steps20.csv
python - $out/r04_kernel_stats_bench_steps20.csv > $out/r04_kernel_stats_top40.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("%-84s %7s %9s %6s" % ("kernel", "calls", "avg us", "%"))
for r in rows[:44]:
    print("%-84s %7s %9.1f %6.2f" % (r["Name"].split("(")[0][:84], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
cd $R
timeout 900 python bench.py > $out/r04_bench_plain.json 2> $out/r04_bench_plain.err
cd /tmp
DIMO_EXEC_STREAMS=0 timeout 300 bash $R/tools/kstats_all.sh $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-dropin --sustained-steps 0 --no-live-pmc > $out/r04_kernel_stats_serial_8renders.txt 2>&1
DIMO_XSTREAM=event timeout 300 bash $R/tools/step_timeline.sh > $out/r04_step_timeline.txt 2>&1
cd $R
python - <<'PY'
import json
d = json.loads(open("gpurun_out/profiles_r04b/r04_bench_plain.json").read().strip().splitlines()[-1])
r = d["roofline"]
print(round(d["value"]), d["ms_per_step"], "roofline", r["frac"], r["avg_ms"], r["traffic"], r["timed_region"]["avg_ms"])
print("dropin", d.get("dropin_frames_per_s"), d.get("dropin_detail"))
print("sustained", d["sustained"]["frames_per_s"] if d.get("sustained") else None, "teacher", d.get("sustained_teacher"))
print("cpu", d.get("cpu_baseline", {}).get("value"), "s1", d.get("s1_frames_per_s"))
PY
head -n 30 $out/r04_step_timeline.txt
