# Collects the rocprofv3 evidence kept under profiles/ (run on the GPU box: bash tools/collect_profiles.sh <tag>).
# Kernel trace/stats and the PMC passes are separate rocprofv3 runs (never combined).
set -u
tag=${1:-r02}
out=$GRAFT_REPO_ROOT/gpurun_out/profiles_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-dropin --sustained-steps 0 > /tmp/bench_prof.log 2>&1
grep "^{\"metric\"" /tmp/bench_prof.log | tail -1 > $out/${tag}_bench_under_rocprof.json
cp $(find /tmp/p1 -name "*kernel_stats.csv" | head -1) $out/${tag}_kernel_stats_bench_steps10.csv
python - $out/${tag}_kernel_stats_bench_steps10.csv > $out/${tag}_kernel_stats_top40.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("%-74s %7s %9s %6s" % ("kernel", "calls", "avg us", "%"))
for r in rows[:40]:
    print("%-74s %7s %9.1f %6.2f" % (r["Name"].split("(")[0][:74], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/p2 -o f -- python $GRAFT_REPO_ROOT/tools/pmc_probe.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/p3 -o w -- python $GRAFT_REPO_ROOT/tools/pmc_probe.py > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/pmc_summarise.py $(find /tmp/p2 -name "*counter_collection.csv" | head -1) $(find /tmp/p3 -name "*counter_collection.csv" | head -1) $out/${tag}_pmc_fetch_write.json 8
bash $GRAFT_REPO_ROOT/tools/pmc_sq.sh $tag blend_bwd_batched blend_fwd_batched ssim_fused lbs_bwd_batched preprocess_bwd image_loss level1_count_batched level1_scatter_batched bucket_sort_batched "level2_batched_kernel<false>" "level2_batched_kernel<true>" timenet_fwd_fused timenet_bwd_fused > $out/${tag}_sq.log 2>&1
# every stage ONE launch over the 8 renders, each kernel alone on the device (the "8 renders" column of DESIGN.md 4)
DIMO_EXEC_STREAMS=0 bash $GRAFT_REPO_ROOT/tools/kstats_all.sh $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-dropin --sustained-steps 0 > $out/${tag}_kernel_stats_serial_8renders.txt 2>&1
cd $GRAFT_REPO_ROOT
python $GRAFT_REPO_ROOT/bench.py > $out/${tag}_bench_plain.json 2>/dev/null
