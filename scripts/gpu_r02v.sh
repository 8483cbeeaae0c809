#!/bin/bash
# r02v: per-stream workgroup sync + K-rotation in the strict ping-pong scan: parity (strict arms) + A/B
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_dense.py -m gpu -q -x --timeout 420 -p no:cacheprovider -k "strict" > gpurun_out/pytest_dense.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/pytest_dense.log
for opt in "--option dense_sync=0 --option dense_rot=0" "--option dense_sync=1 --option dense_rot=0" "--option dense_sync=1 --option dense_rot=-1" "--option dense_sync=0 --option dense_rot=-1" "--option dense_sync=0 --option dense_rot=0" "--option dense_sync=1 --option dense_rot=0" "--option dense_sync=1 --option dense_rot=-1" "--option dense_sync=0 --option dense_rot=-1"; do
  timeout 600 python bench.py --workload hybrid --steps 20 --warmup 3 --cpu-queries 0 $opt > gpurun_out/b.json 2> gpurun_out/b.err; python - "$opt" <<PY
import json, sys
r=json.loads(open("gpurun_out/b.json").read().strip().splitlines()[-1])
print(sys.argv[1], "|", round(r["value"]), r["ms_per_step"], r["roofline"]["frac"], r["kernel_ms_per_step"]["dense_scan"])
PY
done
