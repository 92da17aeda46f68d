set -u
cd $GRAFT_REPO_ROOT
o=gpurun_out/r4u; mkdir -p $o
export TMPDIR=/tmp
for mode in "DIMO_WGRAD_LINEAR=1" "DIMO_WGRAD_LINEAR=1 DIMO_WGRAD_BISECT=1" "DIMO_WGRAD_LINEAR=1 DIMO_WGRAD_BISECT=2" "DIMO_WGRAD_LINEAR=1 DIMO_WGRAD_BISECT=4" "DIMO_WGRAD_LINEAR=1 DIMO_WGRAD_BISECT=8" "DIMO_WGRAD_LINEAR=1 DIMO_WGRAD_BISECT=3" "DIMO_WGRAD_LINEAR=1 DIMO_WGRAD_BISECT=7" "DIMO_WGRAD_LINEAR=1 DIMO_WGRAD_BISECT=15" "DIMO_WGRAD_BISECT=8" "DIMO_WGRAD_BISECT=9"; do
  echo "== $mode" >> $o/probe.txt
  env $mode timeout 100 python tools/timenet_probe.py 50 2>&1 | grep timenet_bwd >> $o/probe.txt
done
cat $o/probe.txt
