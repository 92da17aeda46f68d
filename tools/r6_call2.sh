# round 6, second GPU call: where the init regime's blend backward spends its time (ablation builds: timing only), what
# chained buckets buy there, and their parity
set -u
cd $GRAFT_REPO_ROOT
bash tools/ab.sh r6b -r 1 -s 30 -k -a "--regime init" - @chain2 @chain4 @novisit @noreduce @nockpt
bash tools/ab.sh r6c -r 2 -s 100 - @chain2
libdir=dimo_amd/csrc
cp -f $libdir/libdimo_hip.so $libdir/libdimo_hip.default.so
for v in chain2 chain4; do
  cp -f $libdir/variants/$v.so $libdir/libdimo_hip.so
  ( timeout 900 python -m pytest -q -m gpu -x "tests/test_gpu_executor.py::test_init_regime_batched_kernels_at_c3_against_the_oracle" "tests/test_gpu_executor.py::test_joint_backward_launch_against_the_oracle" "tests/test_gpu_raster.py::test_backward_parity" ) > gpurun_out/r6b/pytest_$v.log 2>&1
  echo "== parity with $v: $(tail -n 1 gpurun_out/r6b/pytest_$v.log)"
done
cp -f $libdir/libdimo_hip.default.so $libdir/libdimo_hip.so
