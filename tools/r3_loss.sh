#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_losses.py -x -q 2>&1 | tail -15 > gpurun_out/r3_loss.log
cat gpurun_out/r3_loss.log
