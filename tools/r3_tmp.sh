#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
{ python tools/s1_probe.py 100; timeout 1200 python -m pytest tests/test_gpu_raster.py tests/test_gpu_executor.py -x -q 2>&1 | tail -3; bash tools/kstats_all.sh $GRAFT_REPO_ROOT/tools/s1_probe.py 30 2>&1 | head -8
  timeout 600 python bench.py --no-cpu-baseline --no-live-pmc --no-dropin --sustained-steps 0 --steps 100 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; } > gpurun_out/r3_s1.log 2>&1
cat gpurun_out/r3_s1.log
