# Per-queue timeline and idle analysis of the bench workload's steady-state step (run on the GPU box).
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl && timeout 280 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o t -- python $GRAFT_REPO_ROOT/tools/pmc_probe.py --no-calibration --default-schedule --steps 12 > /dev/null 2>&1
f=$(find /tmp/tl -name "*kernel_trace.csv" | head -1)
python $GRAFT_REPO_ROOT/tools/trace_gaps.py $f
python $GRAFT_REPO_ROOT/tools/trace_timeline.py $f all
