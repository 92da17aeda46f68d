# round 6, ninth GPU call: the blend kernels request their list entries a round trip early -- parity, A/B (variant nopf)
set -u
cd $GRAFT_REPO_ROOT
o=gpurun_out/r6l; mkdir -p $o
export TMPDIR=/tmp
( time timeout 2400 python -m pytest -q -m gpu -x tests/test_gpu_raster.py tests/test_gpu_executor.py tests/test_gpu_kat.py ) > $o/pytest.log 2>&1
echo "rc=$?" >> $o/pytest.log; tail -n 6 $o/pytest.log
bash tools/ab.sh r6l -r 3 -s 100 -k - @nopf
mv gpurun_out/r6l/modes.txt gpurun_out/r6l/modes_trained.txt
bash tools/ab.sh r6l_init -r 2 -s 30 -k -a "--regime init" - @nopf
