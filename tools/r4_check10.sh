set -u
cd $GRAFT_REPO_ROOT
o=gpurun_out/r4j; mkdir -p $o
for b in 0 1 2; do echo "bisect $b" >> $o/probe.txt; DIMO_TIMENET_BISECT=$b python tools/timenet_probe.py 50 2>&1 | grep timenet_fwd >> $o/probe.txt; done
cat $o/probe.txt
