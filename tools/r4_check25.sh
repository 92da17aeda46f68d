set -u
cd $GRAFT_REPO_ROOT
o=gpurun_out/r4z; mkdir -p $o
export TMPDIR=/tmp
for mode in "DIMO_XSTREAM=event" "DIMO_EXEC_STREAMS=0" "DIMO_JOINT_BWD=1" "DIMO_WGRAD=32 DIMO_SPLIT_ADAM=0 DIMO_SIDE_KNN=0 DIMO_SKIN_IN_ORDER=0 DIMO_REPORT=0 DIMO_ZERO_NEXT=0"; do
  echo "== $mode" >> $o/t.log
  ( env $mode timeout 400 python -m pytest tests/test_gpu_losses.py tests/test_gpu_trains.py tests/test_gpu_determinism.py tests/test_gpu_reference_loop.py -x -q -m gpu 2>&1 | tail -n 3 ) >> $o/t.log
done
cat $o/t.log
