#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
{ python tools/timenet_probe.py 30; timeout 600 python -m pytest tests/test_gpu_timenet.py -x -q 2>&1 | tail -3; } > gpurun_out/r3_tnq.log 2>&1
cat gpurun_out/r3_tnq.log
