# Round-6 evidence for profiles/ (run on the GPU box: bash tools/collect_profiles_r06.sh [core|rest|all]).  Kernel
# traces and PMC passes are separate rocprofv3 runs (never combined).
set -u
tag=r06
part=${1:-all}   # all | core (bench under rocprof, PMC traffic, driver-flag runs, the plain default run) | rest
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/profiles_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
top40() {  # kernel_stats.csv -> table (anonymous-namespace kernels keep their names: round-5 review item 6)
python - "$1" <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("%-84s %7s %9s %6s" % ("kernel", "calls", "avg us", "%"))
for r in rows[:44]:
    name = r["Name"].replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*$", "", name) or r["Name"][:84]
    print("%-84s %7s %9.1f %6.2f" % (name[:84], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
}
if [ "$part" != "rest" ]; then
# 1. kernel stats of the bench command with the driver's flags (default schedule + the serial roofline pass: the joint
#    launch is blend_bwd_batched_kernel<true, true>, the per-motion launches <true, false>)
rm -rf /tmp/p1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dropin --sustained-steps 0 --no-live-pmc --no-regimes > /tmp/bench_prof.log 2>&1
grep "^{\"metric\"" /tmp/bench_prof.log | tail -1 > $out/${tag}_bench_under_rocprof.json
cp $(find /tmp/p1 -name "*kernel_stats.csv" | head -1) $out/${tag}_kernel_stats_bench_steps20.csv
top40 $out/${tag}_kernel_stats_bench_steps20.csv > $out/${tag}_kernel_stats_top40.txt
# ... and of the same command in the init regime
rm -rf /tmp/p1i
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1i -o bench -- python $R/bench.py --regime init --steps 20 --warmup 5 --no-cpu-baseline --no-dropin --sustained-steps 0 --no-live-pmc --no-regimes > /tmp/bench_prof_i.log 2>&1
grep "^{\"metric\"" /tmp/bench_prof_i.log | tail -1 > $out/${tag}_bench_under_rocprof_init.json
top40 $(find /tmp/p1i -name "*kernel_stats.csv" | head -1) > $out/${tag}_kernel_stats_top40_init.txt
# 2. HBM traffic of the roofline kernel (FETCH_SIZE / WRITE_SIZE, separate passes, 1 GiB copy calibration), both regimes
for rg in trained init; do
  rm -rf /tmp/p2 /tmp/p3
  timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/p2 -o f -- python $R/tools/pmc_probe.py --regime $rg > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/p3 -o w -- python $R/tools/pmc_probe.py --regime $rg > /dev/null 2>&1
  sfx=""; [ $rg = init ] && sfx="_init"
  python $R/tools/pmc_summarise.py $(find /tmp/p2 -name "*counter_collection.csv" | head -1) $(find /tmp/p3 -name "*counter_collection.csv" | head -1) $out/${tag}_pmc_fetch_write$sfx.json 8
done
# 3. the driver's command, plain, three times (fresh processes), then the plain default run (everything on: both regimes,
#    drop-in figures, render_fps, sustained + teacher, CPU baselines, live PMC)
cd $R
for i in 1 2 3; do timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dropin --sustained-steps 0 --no-live-pmc --no-regimes 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('bench.py --steps 20 --warmup 5: %.0f frames/s, %.4f ms/step (synced %.0f, behind the idle gap %.0f), roofline frac %.3f (avg %.4f ms), timed-region %.4f ms' % (d['value'], d['ms_per_step'], d['synced_frames_per_s'], d['value_behind_idle_gap'], d['roofline']['frac'], d['roofline']['avg_ms'], d['roofline']['timed_region']['avg_ms']))
"; done > $out/${tag}_bench_driver_flags.txt
timeout 1500 python bench.py > $out/${tag}_bench_plain.json 2> $out/${tag}_bench_plain.err
fi
if [ "$part" = "core" ]; then ls -la $out; exit 0; fi
cd /tmp
# 4. SQ counters
timeout 700 bash $R/tools/pmc_sq.sh $tag blend_bwd_batched blend_fwd_batched ssim_loss_tile lbs_bwd_batched accumulate_batched preprocess_bwd level1_batched bucket_sort_batched level2_fill_batched timenet_fwd_fused timenet_bwd_fused8 knn4 wgrad > $out/${tag}_sq.log 2>&1
cd /tmp
# 5. every stage ONE launch over the 8 renders, each kernel alone on the device: both regimes
DIMO_EXEC_STREAMS=0 timeout 300 bash $R/tools/kstats_all.sh $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-dropin --sustained-steps 0 --no-live-pmc --no-regimes > $out/${tag}_kernel_stats_serial_8renders.txt 2>&1
DIMO_EXEC_STREAMS=0 timeout 300 bash $R/tools/kstats_all.sh $R/bench.py --regime init --steps 10 --warmup 3 --no-cpu-baseline --no-dropin --sustained-steps 0 --no-live-pmc --no-regimes > $out/${tag}_kernel_stats_serial_8renders_init.txt 2>&1
# 6. step timeline of the default schedule (events instead of stream memory operations: those show up as kernels in a trace)
DIMO_XSTREAM=event timeout 300 bash $R/tools/step_timeline.sh > $out/${tag}_step_timeline.txt 2>&1
cd $R
# 7. what ONE rank of an 8-GPU node does per step, measured alone on this GPU (DESIGN section 6 derives its table from
#    these): the reference's b = 2 step sharded over 8 / 4 / 2 ranks = 2 / 4 / 8 renders per rank, b = 4 over 8 ranks = 16
for shape in "1,1,2" "1,2,2" "2,2,2" "2,2,4" "1,4,4"; do
  timeout 300 python bench.py --per-gpu $shape --steps 50 --warmup 10 --no-cpu-baseline --no-dropin --sustained-steps 0 --no-live-pmc --no-regimes 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
p = d['rank_phases_ms'][0]
print('--per-gpu %-6s %3d renders/step  %.4f ms/step  %6.0f frames/s   head %.3f  chains %.3f  tail %.3f ms' % ('$shape', d['config']['renders_per_step'], d['ms_per_step'], d['value'], p.get('head_ms', 0), p.get('chains_ms', 0), p.get('tail_ms', 0)))
"; done > $out/${tag}_per_rank_shapes.txt
# 8. the training evidence, the drop-in loop's host profile, determinism
timeout 300 python tools/teacher_student.py --trace-every 40 > $out/${tag}_teacher_student.json 2>/dev/null
timeout 300 python tools/literal_loop_profile.py --log 1 2>&1 | grep -v "amdgpu.ids\|UserWarning\|_warn_once\|ROCTracer" | head -n 120 > $out/${tag}_literal_loop_profile.txt
timeout 300 python tools/determinism_probe.py > $out/${tag}_determinism.json 2>/dev/null
# 9. BASELINE.md section 2's other shapes on one GPU: C2 (50 k, 256^2, 4 views), C5's shape, the reference's b = 2 step
timeout 400 python bench.py --num-pts 50000 --resolution 256 --per-gpu 1,4,1 --steps 50 --warmup 10 --no-dropin --sustained-steps 0 --no-live-pmc --no-regimes > $out/${tag}_bench_c2.json 2>/dev/null
timeout 400 python bench.py --num-pts 200000 --resolution 1024 --per-gpu 1,1,20 --steps 10 --warmup 3 --no-cpu-baseline --no-dropin --sustained-steps 0 --no-live-pmc --no-regimes > $out/${tag}_bench_c5_shape_1gpu.json 2>/dev/null
timeout 400 python bench.py --global-batch 2 --steps 20 --warmup 5 --no-cpu-baseline --no-dropin --sustained-steps 0 --no-live-pmc --no-regimes > $out/${tag}_bench_strong_b2_1gpu.json 2>/dev/null
# 10. blend backward: the trace build on both regimes (dimo_amd/csrc/variants/trace.so, if present)
if [ -f dimo_amd/csrc/variants/trace.so ]; then
  cp -f dimo_amd/csrc/libdimo_hip.so /tmp/libdimo_hip.keep.so
  cp -f dimo_amd/csrc/variants/trace.so dimo_amd/csrc/libdimo_hip.so
  timeout 300 python tools/bwd_trace.py 2>/dev/null | grep -v amdgpu > $out/${tag}_blend_bwd_trace.txt
  timeout 300 python tools/bwd_trace.py --regime init 2>/dev/null | grep -v amdgpu > $out/${tag}_blend_bwd_trace_init.txt
  cp -f /tmp/libdimo_hip.keep.so dimo_amd/csrc/libdimo_hip.so
fi
ls -la $out
