#!/bin/bash
# HBM traffic of each workload's dominant kernel class from the PMC counters: one rocprofv3 pass with FETCH_SIZE only
# (--kernel-trace, no other trace domain), corrected as MI355X_MICROARCH.md "HBM" prescribes (x2: 128-byte requests are
# tallied at 64 bytes on gfx950; FETCH_SIZE is in KiB).  Writes gpurun_out/pmc_traffic.json; copy it to
# profiles/pmc_traffic.json, where bench.py picks it up for roofline.traffic.
set -u
mkdir -p gpurun_out/traffic
export TMPDIR=/tmp
cd /tmp
for wl in hybrid dense bm25; do
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/traffic/$wl -o p -- \
    python $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 3 --warmup 1 --cpu-queries 0 --sub 0 > $GRAFT_REPO_ROOT/gpurun_out/traffic/$wl.log 2>&1
  echo "$wl exit $?"
done
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/traffic/hybrid_dirs4 -o p -- \
  python $GRAFT_REPO_ROOT/bench.py --workload hybrid --dirs 4 --steps 3 --warmup 1 --cpu-queries 0 --sub 0 > $GRAFT_REPO_ROOT/gpurun_out/traffic/hybrid_dirs4.log 2>&1
echo "hybrid_dirs4 exit $?"
cd $GRAFT_REPO_ROOT
python scripts/traffic_summary.py > gpurun_out/pmc_traffic.json
head -c 3000 gpurun_out/pmc_traffic.json
