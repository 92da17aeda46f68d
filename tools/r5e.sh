set -u
cd $GRAFT_REPO_ROOT
o=gpurun_out/r5e; mkdir -p $o
( timeout 300 python -m pytest tests/test_gpu_reference_loop.py tests/test_gpu_batched_render.py -q -m gpu ) > $o/pytest.log 2>&1; echo "rc=$?" >> $o/pytest.log
tail -n 4 $o/pytest.log
for rep in 1 2; do for m in select unbind; do timeout 200 python tools/literal_fps.py $m 2>/dev/null | tail -1 | tee -a $o/literal.txt; done; done
bash tools/ab.sh r5e_ab -r 2 -s 100 -k "-" "@noslp"
