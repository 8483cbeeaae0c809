#!/bin/bash
# r02o: epilogue max4 fast path, 128x256 seed tiles for small grids, BM25 crossing (template): parity subset + timings
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_dense.py -m gpu -q -x --timeout 420 -p no:cacheprovider -k "strict or gemv or lean" > gpurun_out/pytest_dense.log 2>&1; echo "pytest exit $?"; tail -4 gpurun_out/pytest_dense.log
timeout 800 python scripts/kbench.py bm25x 2>&1 | grep -v amdgpu.ids | tee gpurun_out/kbench_bm25x.log
for wl in dense hybrid; do
  timeout 600 python bench.py --workload $wl --steps 20 --warmup 3 --cpu-queries 0 > gpurun_out/bench_$wl.json 2> gpurun_out/bench_$wl.err; echo "bench $wl exit $?"; python - <<PY
import json
r=json.loads(open("gpurun_out/bench_$wl.json").read().strip().splitlines()[-1])
print("$wl", r["value"], r["ms_per_step"], r["roofline"]["frac"], r["kernel_ms_per_step"])
PY
done
