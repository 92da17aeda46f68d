# correctness of the binning rewrite + serial kernel stats + a bench line (GPU box)
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_raster.py tests/test_gpu_executor.py tests/test_gpu_kat.py -x -q -m gpu 2>&1 | tail -15
export DIMO_EXEC_STREAMS=0
bash tools/kstats_all.sh $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-dropin --sustained-steps 0 > gpurun_out/r3_kstats_serial.txt 2>&1
unset DIMO_EXEC_STREAMS
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3_bench1.json 2> gpurun_out/r3_bench1.err
head -c 400 gpurun_out/r3_bench1.json
