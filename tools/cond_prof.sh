cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out
for mode in -1 0; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cond_$mode -- python3 $R/tools/cond_once.py $mode > $OUT/cond_$mode.log 2>&1 || exit 1
  grep "call" $OUT/cond_$mode.log
done
