# round 6, first GPU call: the init regime's parity tests, both regimes through bench.py, the backward's item trace and
# the serial kernel table in the init regime
set -u
cd $GRAFT_REPO_ROOT
o=gpurun_out/r6a; mkdir -p $o
export TMPDIR=/tmp
( time timeout 1500 python -m pytest -q -m gpu -x \
    "tests/test_gpu_raster.py::test_init_regime_at_c3_size_against_the_oracle" \
    "tests/test_gpu_executor.py::test_init_regime_batched_kernels_at_c3_against_the_oracle" \
    "tests/test_gpu_deform.py::test_train_step_at_benchmark_size_default_switches_matches_cpu_oracle_pipeline" ) > $o/pytest.log 2>&1
echo "rc=$?" >> $o/pytest.log; tail -n 15 $o/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-dropin --no-cpu-baseline --sustained-steps 0 > $o/bench.json 2> $o/bench.err
tail -n 3 $o/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r6a/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "synced", d.get("synced_frames_per_s"), "gap", d.get("value_behind_idle_gap"))
print(json.dumps(d.get("regimes"), indent=1)[:6000])
PY
( cd /tmp && DIMO_EXEC_STREAMS=0 timeout 300 bash $GRAFT_REPO_ROOT/tools/kstats_all.sh $GRAFT_REPO_ROOT/bench.py --regime init --steps 10 --warmup 3 --no-cpu-baseline --no-dropin --sustained-steps 0 --no-live-pmc --no-regimes ) > $o/kstats_serial_init.txt 2>&1
head -n 30 $o/kstats_serial_init.txt
cp -f dimo_amd/csrc/libdimo_hip.so dimo_amd/csrc/libdimo_hip.default.so
cp -f dimo_amd/csrc/variants/trace.so dimo_amd/csrc/libdimo_hip.so
timeout 300 python tools/bwd_trace.py --regime init > $o/bwd_trace_init.txt 2>&1
cp -f dimo_amd/csrc/libdimo_hip.default.so dimo_amd/csrc/libdimo_hip.so
grep -v Warning $o/bwd_trace_init.txt | tail -n 40
