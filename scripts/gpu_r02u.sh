#!/bin/bash
# r02u: tiled chunk layout for the strict ping-pong scan: parity (strict arms) + A/B bench
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_dense.py -m gpu -q -x --timeout 420 -p no:cacheprovider -k "strict or gemv" > gpurun_out/pytest_dense.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/pytest_dense.log
for cfg in "hybrid:" "hybrid:--option dense_tiled=0" "dense:" "dense:--option dense_tiled=0" "hybrid:" "hybrid:--option dense_tiled=0" "dense:" "dense:--option dense_tiled=0"; do
  wl=${cfg%%:*}; opt=${cfg#*:}
  timeout 600 python bench.py --workload $wl --steps 20 --warmup 3 --cpu-queries 0 $opt > gpurun_out/b.json 2> gpurun_out/b.err; python - "$wl $opt" <<PY
import json, sys
r=json.loads(open("gpurun_out/b.json").read().strip().splitlines()[-1])
print(sys.argv[1], "|", round(r["value"]), r["ms_per_step"], r["roofline"]["frac"], r["kernel_ms_per_step"]["dense_scan"])
PY
done
