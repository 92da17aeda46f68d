#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
python tools/loss_probe.py 4 8 > gpurun_out/r3_loss.log 2>&1
cat gpurun_out/r3_loss.log
