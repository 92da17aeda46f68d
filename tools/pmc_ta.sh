# usage: bash tools/pmc_ta.sh <kernel-name-pattern> ...   (texture-address / L1 counters of the bench workload's kernels:
# is a kernel bound by the RATE of its vector-memory instructions -- lanes of a load on 64 different lines -- rather than
# by bytes or latency?)
cd /tmp && export TMPDIR=/tmp
for grp in "TA_BUSY_avr TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCP_TAGRAM0_REQ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum"; do
  n=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rm -rf /tmp/pmc_$n
  timeout 200 rocprofv3 --pmc $grp --output-format csv -d /tmp/pmc_$n -o p -- python $GRAFT_REPO_ROOT/tools/pmc_probe.py --no-calibration > /dev/null 2>&1
  f=$(find /tmp/pmc_$n -name "*counter_collection.csv" | head -1)
  python $GRAFT_REPO_ROOT/tools/pmc_sq_summarise.py $f "$@"
done
