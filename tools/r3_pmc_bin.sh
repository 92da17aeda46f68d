cd $GRAFT_REPO_ROOT
export DIMO_EXEC_STREAMS=0
bash tools/pmc_sq.sh r3bin level1_count_batched level1_scatter_batched bucket_sort_batched "level2_batched_kernel<false>" "level2_batched_kernel<true>" preprocess_fwd_batched > gpurun_out/r3_pmc_bin.log 2>&1
