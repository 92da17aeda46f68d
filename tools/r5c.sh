set -u
cd $GRAFT_REPO_ROOT
o=gpurun_out/r5c; mkdir -p $o
( timeout 600 python -m pytest tests/test_gpu_losses.py tests/test_gpu_ops.py -q -m gpu -k "one_pass or ssim or image_loss_kernel" ) > $o/pytest.log 2>&1; echo "rc=$?" >> $o/pytest.log
tail -n 8 $o/pytest.log
for w in 3 4; do DIMO_SSIM_WGS=$w timeout 120 python tools/loss_probe.py 4 8 2>&1 | grep "B =" | tee -a $o/probe.txt; done
bash tools/ab.sh r5c_ab -r 2 -s 100 "-" "DIMO_SSIM_WGS=4" "DIMO_FUSED_LOSS=1" "DIMO_FUSED_LOSS=1 DIMO_SSIM_WGS=4" "DIMO_TIMENET_ROWS_FWD=8 DIMO_KNN_WGS=64" "DIMO_TIMENET_ROWS_FWD=8 DIMO_KNN_WGS=128"
