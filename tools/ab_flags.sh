# A/B of a compiler flag over the whole library on the GPU box, kernels serialised (DIMO_EXEC_STREAMS=0):
#     bash tools/ab_flags.sh "<extra flags>"
cd $GRAFT_REPO_ROOT
export DIMO_EXEC_STREAMS=0
echo "== baseline"
bash tools/kstats_probe.sh $GRAFT_REPO_ROOT/tools/pmc_probe.py --no-calibration 2>&1 | head -24
python - "$1" <<'PY'
import sys, re
p = "dimo_amd/csrc/build.py"
s = open(p).read()
s = s.replace('COMMON = ["-O3",', 'COMMON = %r + ["-O3",' % sys.argv[1].split())
open(p, "w").write(s)
PY
python -m dimo_amd.csrc.build --force > /dev/null 2>&1
echo "== with $1"
bash tools/kstats_probe.sh $GRAFT_REPO_ROOT/tools/pmc_probe.py --no-calibration 2>&1 | head -24
