#!/bin/bash
set -u
OUT=gpurun_out/r06g
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_fullsize.py -k "dir_blocks" -m gpu -q --timeout 600 -p no:cacheprovider > $OUT/pytest_full.log 2>&1; echo "pytest exit $?"; grep -v amdgpu.ids $OUT/pytest_full.log | tail -5
for ql in fixed ref; do
  timeout 300 python bench.py --workload hybrid --dirs 4 --qlen $ql --steps 20 --warmup 5 --cpu-queries 0 --sub 0 > $OUT/bench_filtered_$ql.json 2>/dev/null
  python - $OUT/bench_filtered_$ql.json <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], round(r["ms_per_step"], 4), {k: round(v, 4) for k, v in r["kernel_ms_per_step"].items() if v}, {k: v for k, v in r["path"].items() if v})
PY
done
