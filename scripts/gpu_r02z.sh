#!/bin/bash
# r02z: end-of-round measurement refresh: full GPU parity suite, smoke, bench lines, kernel stats, PMC traffic, PMC passes
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python - <<'PY' > gpurun_out/build.log 2>&1
import __graft_entry__ as g
g.build()
PY
tail -1 gpurun_out/build.log
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -6 gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"; tail -2 gpurun_out/smoke.log
for wl in hybrid dense bm25; do
  timeout 600 python bench.py --workload $wl --steps 20 --warmup 3 > gpurun_out/bench_$wl.json 2> gpurun_out/bench_$wl.err; echo "bench $wl exit $?"
done
timeout 600 python bench.py --workload hybrid --variant okapi --steps 10 --warmup 2 > gpurun_out/bench_hybrid_okapi.json 2> gpurun_out/bench_hybrid_okapi.err; echo "bench hybrid okapi exit $?"
for wl in hybrid dense bm25; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$wl -o $wl -- python $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 5 --warmup 1 --cpu-queries 0 > $GRAFT_REPO_ROOT/gpurun_out/prof_$wl.log 2>&1); echo "rocprof $wl exit $?"
  f=$(find gpurun_out/prof_$wl -name "*kernel_stats.csv" | head -1)
  python scripts/trim_stats.py $f gpurun_out/${wl}_kernel_stats.csv > /dev/null
done
bash scripts/gpu_traffic.sh > gpurun_out/traffic.log 2>&1; echo "traffic exit $?"; tail -3 gpurun_out/traffic.log
bash scripts/gpu_pmc.sh dense "--batch 1024" r02c > gpurun_out/pmc_dense_b1024.txt 2>&1; tail -5 gpurun_out/pmc_dense_b1024.txt
bash scripts/gpu_pmc.sh dense "" r02c256 > gpurun_out/pmc_dense_b256.txt 2>&1; tail -5 gpurun_out/pmc_dense_b256.txt
bash scripts/gpu_pmc.sh bm25 "" r02c > gpurun_out/pmc_bm25.txt 2>&1; tail -6 gpurun_out/pmc_bm25.txt
for f in gpurun_out/bench_*.json; do echo $f; python - $f <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(round(r["value"]), r["ms_per_step"], r["roofline"]["frac"], r["roofline"].get("traffic"), r["kernel_ms_per_step"], (r.get("cpu_baseline") or {}).get("value"))
PY
done
