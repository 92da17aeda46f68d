# round 6, eighth GPU call: hit mask or flag (not both) -- parity incl. the >64-tile case, A/B against the flag loop
set -u
cd $GRAFT_REPO_ROOT
o=gpurun_out/r6k; mkdir -p $o
export TMPDIR=/tmp
( time timeout 2400 python -m pytest -q -m gpu -x tests/test_gpu_raster.py tests/test_gpu_executor.py tests/test_gpu_kat.py tests/test_gpu_deform.py tests/test_gpu_determinism.py tests/test_gpu_trains.py ) > $o/pytest.log 2>&1
echo "rc=$?" >> $o/pytest.log; tail -n 6 $o/pytest.log
bash tools/ab.sh r6k -r 2 -s 100 -k - @nohit
mv gpurun_out/r6k/modes.txt gpurun_out/r6k/modes_trained.txt
bash tools/ab.sh r6k_init -r 2 -s 30 -k -a "--regime init" - @nohit
timeout 300 python bench.py --num-pts 512 --resolution 512 --steps 5 --warmup 2 --no-cpu-baseline --sustained-steps 0 --no-live-pmc --no-regimes 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('s1', d.get('s1_frames_per_s'), d.get('s1_what'))"
