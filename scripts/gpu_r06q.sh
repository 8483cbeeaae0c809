#!/bin/bash
set -u
OUT=gpurun_out/r06q
mkdir -p $OUT
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider -x > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?"; grep -v amdgpu.ids $OUT/pytest_gpu.log | tail -4
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_hybrid.json 2> $OUT/bench_hybrid.err; echo "bench exit $?"
python - <<'PY'
import json
r = json.loads(open("gpurun_out/r06q/bench_hybrid.json").read().strip().splitlines()[-1])
print(round(r["value"]), "q/s", round(r["ms_per_step"], 4), "ms", {k: round(v, 4) for k, v in r["kernel_ms_per_step"].items() if v})
for k, v in r["sub_benchmarks"].items():
    print(" ", k, round(v["ms_per_step"], 4), {a: round(b, 4) for a, b in v["kernel_ms_per_step"].items() if b})
PY
