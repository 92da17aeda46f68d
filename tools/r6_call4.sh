# round 6, fourth GPU call: the group-loop skinning backward -- parity, determinism, training, then A/B against the
# three-kernel chain of rounds 2-5 (variant oldlbs) and against 256-thread workgroups
set -u
cd $GRAFT_REPO_ROOT
o=gpurun_out/r6e; mkdir -p $o
export TMPDIR=/tmp
( time timeout 2400 python -m pytest -q -m gpu -x tests/test_gpu_deform.py tests/test_gpu_determinism.py tests/test_gpu_executor.py tests/test_gpu_trains.py tests/test_gpu_reference_loop.py tests/test_gpu_batched_render.py tests/test_gpu_bench.py tests/test_gpu_losses.py ) > $o/pytest.log 2>&1
echo "rc=$?" >> $o/pytest.log; tail -n 8 $o/pytest.log
bash tools/ab.sh r6e -r 3 -s 100 -k - @oldlbs @lbs256
