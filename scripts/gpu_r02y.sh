#!/bin/bash
# r02y: final refresh after the last source change: PMC traffic (digest-bound), then the four bench lines (traffic attached), determinism
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
bash scripts/gpu_traffic.sh > gpurun_out/traffic.log 2>&1; echo "traffic exit $?"
cp gpurun_out/pmc_traffic.json profiles/pmc_traffic.json
for wl in hybrid dense bm25; do
  timeout 600 python bench.py --workload $wl --steps 20 --warmup 3 > gpurun_out/bench_$wl.json 2> gpurun_out/bench_$wl.err; echo "bench $wl exit $?"
done
timeout 600 python bench.py --workload hybrid --variant okapi --steps 10 --warmup 2 > gpurun_out/bench_hybrid_okapi.json 2> gpurun_out/bench_hybrid_okapi.err; echo "bench hybrid okapi exit $?"
for wl in hybrid dense bm25; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$wl -o $wl -- python $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 5 --warmup 1 --cpu-queries 0 > $GRAFT_REPO_ROOT/gpurun_out/prof_$wl.log 2>&1); echo "rocprof $wl exit $?"
  f=$(find gpurun_out/prof_$wl -name "*kernel_stats.csv" | head -1)
  python scripts/trim_stats.py $f gpurun_out/${wl}_kernel_stats.csv > /dev/null
done
timeout 900 python scripts/determinism.py 20 > gpurun_out/determinism.log 2>&1; echo "determinism exit $?"; cat gpurun_out/determinism.log | grep -v amdgpu
for f in gpurun_out/bench_*.json; do echo $f; python - $f <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(round(r["value"]), r["ms_per_step"], r["roofline"]["frac"], r["roofline"].get("traffic"), r["roofline"].get("traffic_over_algorithmic"), r["kernel_ms_per_step"], (r.get("cpu_baseline") or {}).get("value"))
PY
done
