# SQ-counter summary of the bench workload's dominant kernels (run on the GPU box):
#     bash tools/pmc_sq.sh <tag> [kernel-name pattern ...]
# Three rocprofv3 --pmc passes (8 SQ slots each, never combined with a trace) over tools/pmc_probe.py, merged into
# gpurun_out/profiles_<tag>/<tag>_sq_counters.json with the derived figures DESIGN.md quotes.
set -u
tag=${1:-r02}
shift || true
pats=${*:-blend_bwd_batched blend_fwd_batched ssim_fused}
out=$GRAFT_REPO_ROOT/gpurun_out/profiles_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
csvs=""
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_BRANCH SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE"; do
  i=$((i + 1))
  rm -rf /tmp/sq_$i
  timeout 200 rocprofv3 --pmc $grp --output-format csv -d /tmp/sq_$i -o p -- python $GRAFT_REPO_ROOT/tools/pmc_probe.py --no-calibration > /tmp/sq_$i.log 2>&1
  f=$(find /tmp/sq_$i -name "*counter_collection.csv" | head -1)
  csvs="$csvs $f"
done
python $GRAFT_REPO_ROOT/tools/pmc_sq_summarise.py --json $out/${tag}_sq_counters.json --csv $csvs -- $pats
