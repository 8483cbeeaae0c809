#!/bin/bash
set -u
OUT=gpurun_out/r06e
mkdir -p $OUT
export TMPDIR=/tmp
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_filtered -o filtered -- \
   python $GRAFT_REPO_ROOT/bench.py --workload hybrid --dirs 4 --steps 10 --warmup 2 --cpu-queries 0 --sub 0 > $GRAFT_REPO_ROOT/$OUT/prof_filtered.log 2>&1)
f=$(find $OUT/prof_filtered -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && python scripts/trim_stats.py "$f" $OUT/filtered_kernel_stats.csv
tail -c 1500 $OUT/prof_filtered.log
rm -rf $OUT/prof_filtered
