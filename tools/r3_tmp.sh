#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
{ timeout 1200 python -m pytest tests/test_gpu_raster.py tests/test_gpu_executor.py tests/test_gpu_kat.py -x -q 2>&1 | tail -3
  for i in 1 2; do timeout 600 python bench.py --no-cpu-baseline --no-live-pmc --no-dropin --sustained-steps 0 --steps 100 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernels_ms_per_launch'].get('blend_fwd'), d['kernels_ms_per_launch_isolated'].get('blend_fwd'), d['kernels_ms_per_launch'].get('place'))"; done
  DIMO_EXEC_STREAMS=0 timeout 600 python bench.py --no-cpu-baseline --no-live-pmc --no-dropin --sustained-steps 0 --steps 50 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('one stream', d['value'], d['ms_per_step'], d['kernels_ms_per_launch'].get('blend_fwd'))"; } > gpurun_out/r3_bf.log 2>&1
cat gpurun_out/r3_bf.log
