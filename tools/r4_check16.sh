set -u
cd $GRAFT_REPO_ROOT
o=gpurun_out/r4p; mkdir -p $o
export TMPDIR=/tmp
( time python -m pytest tests -x -q -m gpu ) > $o/t_all.log 2>&1
echo "rc=$?" >> $o/t_all.log
tail -n 6 $o/t_all.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $o/smoke.log 2>&1; tail -n 2 $o/smoke.log
bash tools/collect_profiles_r04.sh > $o/collect.log 2>&1
tail -n 30 $o/collect.log
