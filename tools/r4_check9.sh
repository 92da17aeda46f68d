set -u
cd $GRAFT_REPO_ROOT
o=gpurun_out/r4i; mkdir -p $o
export TMPDIR=/tmp
( python -m pytest tests/test_gpu_timenet.py -x -q -m gpu ) > $o/t8.log 2>&1; echo "rc=$?" >> $o/t8.log
python tools/timenet_probe.py 50 > $o/probe8.txt 2>&1
DIMO_TIMENET_ROWS=16 python tools/timenet_probe.py 50 > $o/probe16.txt 2>&1
tail -n 5 $o/t8.log; grep -v amdgpu $o/probe8.txt; grep -v amdgpu $o/probe16.txt
