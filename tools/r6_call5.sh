# round 6, fifth GPU call: skinning backward with atomic control-point rows (no partial tables, no reduce kernel), groups in
# parallel again -- parity, determinism, A/B against rounds 2-5's chain (variant oldlbs)
set -u
cd $GRAFT_REPO_ROOT
o=gpurun_out/r6f; mkdir -p $o
export TMPDIR=/tmp
( time timeout 2400 python -m pytest -q -m gpu -x tests/test_gpu_deform.py tests/test_gpu_determinism.py tests/test_gpu_executor.py tests/test_gpu_trains.py tests/test_gpu_reference_loop.py tests/test_gpu_batched_render.py tests/test_gpu_binning_fuzz.py ) > $o/pytest.log 2>&1
echo "rc=$?" >> $o/pytest.log; tail -n 8 $o/pytest.log
bash tools/ab.sh r6f -r 3 -s 100 -k - @oldlbs
