#!/bin/bash
# r02p: rocprofv3 kernel traces (hybrid + dense workloads) of the current build, trimmed to this library's kernels
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
for wl in hybrid dense; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$wl -o $wl -- python $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 5 --warmup 1 --cpu-queries 0 > $GRAFT_REPO_ROOT/gpurun_out/prof_$wl.log 2>&1); echo "rocprof $wl exit $?"
  f=$(find gpurun_out/prof_$wl -name "*kernel_stats.csv" | head -1)
  python scripts/trim_stats.py $f gpurun_out/${wl}_kernel_stats.csv | head -24
done
