#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
rm -f gpurun_out/r3_mc.log
for i in 1 2; do for m in 1 2; do echo -n "DIMO_MAIN_CHAIN=$m " >> gpurun_out/r3_mc.log; DIMO_MAIN_CHAIN=$m timeout 600 python bench.py --no-cpu-baseline --no-live-pmc --no-dropin --sustained-steps 0 --steps 100 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> gpurun_out/r3_mc.log; done; done
cat gpurun_out/r3_mc.log
