#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 2000 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r3_mc.log
cat gpurun_out/r3_mc.log
