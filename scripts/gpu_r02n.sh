#!/bin/bash
# r02n: BM25 threshold-crossing path: sparse parity (all arms, small + full size), then kernel times
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_sparse_fusion.py tests/test_gpu_fullsize.py -m gpu -q -x --timeout 600 -p no:cacheprovider -k "bm25 or hybrid" > gpurun_out/pytest_sparse.log 2>&1; echo "pytest exit $?"; tail -8 gpurun_out/pytest_sparse.log
for wl in bm25 hybrid; do
  timeout 600 python bench.py --workload $wl --steps 20 --warmup 3 --cpu-queries 0 > gpurun_out/bench_$wl.json 2> gpurun_out/bench_$wl.err; echo "bench $wl exit $?"; python - <<PY
import json
r=json.loads(open("gpurun_out/bench_$wl.json").read().strip().splitlines()[-1])
print("$wl", r["value"], r["ms_per_step"], r["roofline"]["frac"], r["kernel_ms_per_step"])
PY
done
timeout 600 python bench.py --workload hybrid --variant okapi --steps 10 --warmup 2 --cpu-queries 0 > gpurun_out/bench_hybrid_okapi.json 2> gpurun_out/bench_hybrid_okapi.err; tail -c 400 gpurun_out/bench_hybrid_okapi.json
