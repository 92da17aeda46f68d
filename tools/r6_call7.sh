# round 6, seventh GPU call: the projection backward reads a per-Gaussian hit mask instead of the record flags -- parity,
# A/B against the flag loop (variant nohit) in both regimes
set -u
cd $GRAFT_REPO_ROOT
o=gpurun_out/r6j; mkdir -p $o
export TMPDIR=/tmp
( time timeout 2400 python -m pytest -q -m gpu -x tests/test_gpu_raster.py tests/test_gpu_executor.py tests/test_gpu_kat.py tests/test_gpu_deform.py tests/test_gpu_determinism.py tests/test_gpu_trains.py tests/test_gpu_batched_render.py ) > $o/pytest.log 2>&1
echo "rc=$?" >> $o/pytest.log; tail -n 6 $o/pytest.log
bash tools/ab.sh r6j -r 3 -s 100 -k - @nohit
mv gpurun_out/r6j/modes.txt gpurun_out/r6j/modes_trained.txt
bash tools/ab.sh r6j_init -r 1 -s 30 -k -a "--regime init" - @nohit
