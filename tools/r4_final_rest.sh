# Other shapes and the switch table on the final code (run on the GPU box after r4_final.sh).
set -u
R=$GRAFT_REPO_ROOT
cd $R
export TMPDIR=/tmp
out=$R/gpurun_out/profiles_r04b; mkdir -p $out
timeout 300 python bench.py --num-pts 200000 --resolution 1024 --per-gpu 1,1,20 --steps 10 --warmup 3 --no-cpu-baseline --no-dropin --sustained-steps 0 --no-live-pmc > $out/r04_bench_c5_shape_1gpu.json 2>/dev/null
timeout 300 python bench.py --global-batch 2 --steps 20 --warmup 5 --no-cpu-baseline --no-dropin --sustained-steps 0 --no-live-pmc > $out/r04_bench_strong_b2_1gpu.json 2>/dev/null
rm -f $out/r04_schedule_modes.txt
for mode in "DIMO_JOINT_BWD=0" "DIMO_JOINT_BWD=1" "DIMO_EXEC_STREAMS=0" "DIMO_XSTREAM=event" "DIMO_SIDE_KNN=0" "DIMO_SPLIT_ADAM=0" "DIMO_SKIN_IN_ORDER=0" "DIMO_REPORT=0 DIMO_ZERO_NEXT=0" "DIMO_TIMENET_ROWS=16" "DIMO_TIMENET_ROWS=8" "DIMO_WGRAD=32" "DIMO_PACK_WGS=16" "DIMO_FWD_MFMA=1"; do
  env $mode timeout 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --sustained-steps 0 --no-live-pmc --no-dropin 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-44s %6.0f frames/s  %.4f ms/step' % ('$mode', d['value'], d['ms_per_step']))
" >> $out/r04_schedule_modes.txt
done
cat $out/r04_schedule_modes.txt
( echo "default (forward 16 rows per workgroup, dgrad chain 8, wgrad64 + embedding backward in one launch, packs with 64 workgroups):"; timeout 100 python tools/timenet_probe.py 50 2>&1 | grep timenet; echo "DIMO_TIMENET_ROWS=16:"; DIMO_TIMENET_ROWS=16 timeout 100 python tools/timenet_probe.py 50 2>&1 | grep timenet; echo "DIMO_TIMENET_ROWS=8:"; DIMO_TIMENET_ROWS=8 timeout 100 python tools/timenet_probe.py 50 2>&1 | grep timenet; echo "DIMO_WGRAD=32 DIMO_PACK_WGS=16 (the kernels of the first collection):"; DIMO_WGRAD=32 DIMO_PACK_WGS=16 timeout 100 python tools/timenet_probe.py 50 2>&1 | grep timenet ) > $out/r04_timenet_rows_final.txt
cat $out/r04_timenet_rows_final.txt
python - <<'PY'
import json
for f in ("r04_bench_c5_shape_1gpu.json", "r04_bench_strong_b2_1gpu.json"):
    d = json.loads(open("gpurun_out/profiles_r04b/" + f).read().strip().splitlines()[-1])
    print(f, round(d["value"]), d["ms_per_step"], d["roofline"]["frac"], d.get("peak_memory_gb"))
PY
