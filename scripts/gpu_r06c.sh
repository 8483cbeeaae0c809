#!/bin/bash
# Round 6: anisotropic corpus parity at full size + the driver-style bench line with the new sub-benchmarks
set -u
OUT=gpurun_out/r06c
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_fullsize.py -k "clustered" -m gpu -q -s --timeout 600 -p no:cacheprovider > $OUT/pytest_clustered.log 2>&1; echo "pytest exit $?"; grep -v amdgpu.ids $OUT/pytest_clustered.log | tail -8
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_hybrid.json 2> $OUT/bench_hybrid.err; echo "bench exit $?"; tail -3 $OUT/bench_hybrid.err
python - <<'PY'
import json
r = json.loads(open("gpurun_out/r06c/bench_hybrid.json").read().strip().splitlines()[-1])
print("headline", round(r["value"]), "q/s", round(r["ms_per_step"], 4), "ms", r["kernel_ms_per_step"], r["path"])
for k, v in r["sub_benchmarks"].items():
    print(k, round(v["ms_per_step"], 4), {a: round(b, 4) for a, b in v["kernel_ms_per_step"].items() if b}, (v.get("roofline") or {}).get("frac"), v.get("path"), v.get("dense_candidates_per_query_last_step"))
PY
