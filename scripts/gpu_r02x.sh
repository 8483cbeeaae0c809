#!/bin/bash
# r02x: bench lines only (cpu_baseline now also reports the batched-GEMM dense variant)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
for wl in hybrid dense bm25; do
  timeout 900 python bench.py --workload $wl --steps 20 --warmup 3 > gpurun_out/bench_$wl.json 2> gpurun_out/bench_$wl.err; echo "bench $wl exit $?"; tail -2 gpurun_out/bench_$wl.err
done
timeout 900 python bench.py --workload hybrid --variant okapi --steps 10 --warmup 2 > gpurun_out/bench_hybrid_okapi.json 2> gpurun_out/bench_hybrid_okapi.err; echo "bench hybrid okapi exit $?"
timeout 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench default exit $?"
for f in gpurun_out/bench_*.json; do echo $f; python - $f <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
c = r.get("cpu_baseline") or {}
print(round(r["value"]), r["ms_per_step"], r["roofline"]["frac"], r["roofline"].get("traffic_over_algorithmic"), r["kernel_ms_per_step"], c.get("value"), (c.get("batched_dense") or {}).get("value"))
PY
done
