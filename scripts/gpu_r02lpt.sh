#!/bin/bash
# BM25 longest-first query order: parity (sparse + hybrid tests) + A/B
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_sparse_fusion.py tests/test_gpu_fullsize.py tests/test_gpu_retrievers.py -m gpu -q -x --timeout 600 -p no:cacheprovider > gpurun_out/pytest_sparse.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/pytest_sparse.log
for cfg in "hybrid:--option bm25_lpt=0" "hybrid:" "bm25:--option bm25_lpt=0" "bm25:" "hybrid:--option bm25_lpt=0" "hybrid:" "bm25:--option bm25_lpt=0" "bm25:" "hybrid:--variant okapi --option bm25_lpt=0" "hybrid:--variant okapi"; do
  wl=${cfg%%:*}; opt=${cfg#*:}
  timeout 600 python bench.py --workload $wl --steps 20 --warmup 3 --cpu-queries 0 $opt > gpurun_out/b.json 2> gpurun_out/b.err; python - "$wl $opt" <<PY
import json, sys
r=json.loads(open("gpurun_out/b.json").read().strip().splitlines()[-1])
print(sys.argv[1], "|", round(r["value"]), r["ms_per_step"], r["kernel_ms_per_step"]["bm25_scan"], r["kernel_ms_per_step"]["dense_scan"])
PY
done
