# round 6, sixth GPU call: the blend backward's quadrant visits -- pairs of quadrants in one straight-line block, the
# rejection tests as a select chain -- A/B in both regimes (serial kernel tables), parity of the candidates; the fold's
# 16-byte form; the world-8 gloo bench
set -u
cd $GRAFT_REPO_ROOT
o=gpurun_out/r6g; mkdir -p $o
export TMPDIR=/tmp
bash tools/ab.sh r6g -r 2 -s 100 -k - @pairs @selchain @pairsel
mv gpurun_out/r6g/modes.txt gpurun_out/r6g/modes_trained.txt
bash tools/ab.sh r6g_init -r 1 -s 30 -k -a "--regime init" - @pairs @selchain @pairsel
libdir=dimo_amd/csrc
cp -f $libdir/libdimo_hip.so $libdir/libdimo_hip.default.so
for v in pairs pairsel; do
  cp -f $libdir/variants/$v.so $libdir/libdimo_hip.so
  ( timeout 900 python -m pytest -q -m gpu -x tests/test_gpu_executor.py tests/test_gpu_raster.py::test_backward_parity tests/test_gpu_raster.py::test_backward_parity_four_output_flavour tests/test_gpu_raster.py::test_init_regime_at_c3_size_against_the_oracle tests/test_gpu_kat.py ) > $o/pytest_$v.log 2>&1
  echo "== parity with $v: $(tail -n 1 $o/pytest_$v.log)"
done
cp -f $libdir/libdimo_hip.default.so $libdir/libdimo_hip.so
( time timeout 2400 python -m pytest -q -m gpu -x tests/test_gpu_bench.py tests/test_gpu_deform.py tests/test_gpu_determinism.py ) > $o/pytest.log 2>&1
echo "rc=$?" >> $o/pytest.log; tail -n 6 $o/pytest.log
