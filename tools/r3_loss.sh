#!/bin/bash
# the loss kernels alone + the whole GPU suite + a bench line (GPU box)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
{ python tools/loss_probe.py 4 8
  timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8; } > gpurun_out/r3_loss.log 2>&1
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r3_bench_loss.json 2> gpurun_out/r3_bench_loss.err
cat gpurun_out/r3_loss.log; head -c 600 gpurun_out/r3_bench_loss.json
