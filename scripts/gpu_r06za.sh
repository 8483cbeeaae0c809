#!/bin/bash
set -u
OUT=gpurun_out/r06za
mkdir -p $OUT
export TMPDIR=/tmp
for i in 1 2; do
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-queries 0 > $OUT/bench_$i.json 2> $OUT/bench_$i.err; echo "bench exit $?"
python - $i <<'PY'
import json, sys
r = json.loads(open(f"gpurun_out/r06za/bench_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(round(r["value"]), round(r["ms_per_step"], 4))
for k in ("hybrid_b1024_dir_filter", "hybrid_b1_dir_filter_latency", "hybrid_b1_latency", "bm25_b256_top100"):
    v = r["sub_benchmarks"][k]
    print(" ", k, round(v["ms_per_step"], 4), {a: round(b, 4) for a, b in v["kernel_ms_per_step"].items() if b})
PY
done
