#!/bin/bash
# One GPU call = targeted tests + A/B bench lines + (optionally) serial kernel tables, for any set of switches.
# Replaces the 26 one-off tools/r4_check*.sh of round 4.  Run on the GPU box (gpurun):
#   tools/ab.sh <tag> [-t "<pytest args>"] [-r reps] [-s steps] [-k] "<ENV=.. ENV=..>" "<ENV=..>" ...
#     -t  pytest arguments (quoted); default: none
#     -r  repetitions of every mode's bench line (interleaved: boxes drift); default 2
#     -s  bench steps per line; default 100
#     -a  extra bench.py arguments for every line (quoted), e.g. "--regime init"
#     -k  also a serial kernel table per mode (DIMO_EXEC_STREAMS=0 under rocprofv3: every kernel alone on the device)
#   a mode "-" means "no switch" (the defaults); a mode token "@name" runs that line with the alternative build
#   dimo_amd/csrc/variants/name.so (DIMO_BUILD_VARIANT=name python -m dimo_amd.csrc.build) in place of the library
set -u
cd $GRAFT_REPO_ROOT
tag=$1; shift
tests=""; reps=2; steps=100; kst=0; extra=""
while getopts "t:r:s:ka:" o; do
  case $o in t) tests=$OPTARG;; r) reps=$OPTARG;; s) steps=$OPTARG;; k) kst=1;; a) extra=$OPTARG;; esac
done
shift $((OPTIND - 1))
o=gpurun_out/$tag; mkdir -p $o
export TMPDIR=/tmp
libdir=dimo_amd/csrc
cp -f $libdir/libdimo_hip.so $libdir/libdimo_hip.default.so
if [ -n "$tests" ]; then
  ( time timeout 900 python -m pytest $tests -q -m gpu ) > $o/pytest.log 2>&1; echo "rc=$?" >> $o/pytest.log
  tail -n 25 $o/pytest.log
fi
: > $o/modes.txt
for rep in $(seq 1 $reps); do
  for mode in "$@"; do
    m=$mode; [ "$m" = "-" ] && m="DIMO_AB_NONE=1"
    cp -f $libdir/libdimo_hip.default.so $libdir/libdimo_hip.so
    for tok in $m; do case $tok in @*) cp -f $libdir/variants/${tok#@}.so $libdir/libdimo_hip.so;; esac; done
    m=$(echo "$m" | sed 's/@[A-Za-z0-9_]*//g'); [ -z "$(echo $m | tr -d ' ')" ] && m="DIMO_AB_NONE=1"
    env $m timeout 300 python bench.py --steps $steps --warmup 10 --no-cpu-baseline --sustained-steps 0 --no-live-pmc --no-dropin --no-regimes $extra 2>$o/bench.err | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read().strip().splitlines()[-1])
    k = d.get('kernels_ms_per_launch', {})
    print('$mode', round(d['value']), round(d['ms_per_step'], 4), 'skipped', d['skipped_steps'], {n: round(1e3 * v, 1) for n, v in k.items() if v})
except Exception as e:
    print('$mode', 'FAILED', e)
" >> $o/modes.txt
    tail -n 3 $o/bench.err >> $o/modes.err
  done
done
cp -f $libdir/libdimo_hip.default.so $libdir/libdimo_hip.so
cat $o/modes.txt
if [ $kst = 1 ]; then
  for mode in "$@"; do
    m=$mode; [ "$m" = "-" ] && m="DIMO_AB_NONE=1"
    cp -f $libdir/libdimo_hip.default.so $libdir/libdimo_hip.so
    for tok in $m; do case $tok in @*) cp -f $libdir/variants/${tok#@}.so $libdir/libdimo_hip.so;; esac; done
    m=$(echo "$m" | sed 's/@[A-Za-z0-9_]*//g'); [ -z "$(echo $m | tr -d ' ')" ] && m="DIMO_AB_NONE=1"
    f=$o/kstats_serial_$(echo "$mode" | tr ' =/@' '____').txt
    ( cd /tmp && env $m DIMO_EXEC_STREAMS=0 timeout 300 bash $GRAFT_REPO_ROOT/tools/kstats_all.sh $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-dropin --sustained-steps 0 --no-live-pmc --no-regimes $extra ) > $f 2>&1
    echo "== $mode"; head -n 24 $f
  done
fi
cp -f $libdir/libdimo_hip.default.so $libdir/libdimo_hip.so
