#!/bin/bash
set -u
OUT=gpurun_out/r06zj
mkdir -p $OUT
export TMPDIR=/tmp
for what in filtered hybrid; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_$what -o p -- python $GRAFT_REPO_ROOT/scripts/b1_profile.py $what 200 > $GRAFT_REPO_ROOT/$OUT/prof_$what.log 2>&1)
  f=$(find $OUT/prof_$what -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && python scripts/trim_stats.py "$f" $OUT/b1_${what}_kernel_stats.csv > /dev/null
  rm -rf $OUT/prof_$what
  head -30 $OUT/b1_${what}_kernel_stats.csv | cut -c1-160
done
