#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
rm -f gpurun_out/r3_mc.log
for i in 1 2 3; do timeout 600 python bench.py --no-cpu-baseline --no-live-pmc --no-dropin --sustained-steps 0 --steps 100 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> gpurun_out/r3_mc.log; done
timeout 900 python -m pytest tests/test_gpu_losses.py tests/test_gpu_raster.py -x -q 2>&1 | tail -2 >> gpurun_out/r3_mc.log
cat gpurun_out/r3_mc.log
