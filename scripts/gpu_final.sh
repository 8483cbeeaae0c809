#!/bin/bash
# After the LAST change to a kernel source: the FETCH_SIZE pass and the four bench lines on exactly those sources, one
# lease.  bench.py attaches roofline.traffic only when profiles/pmc_traffic.json carries the digest of the kernel sources
# it runs, so the table is installed before the lines are taken.  Copy gpurun_out/final/* to profiles/ afterwards
# (bench_W.json -> rNNz_bench_W.json, pmc_traffic.json, traffic.log -> rNNz_traffic.log).
set -u
mkdir -p gpurun_out/final
bash scripts/gpu_traffic.sh > gpurun_out/final/traffic.log 2>&1
cp gpurun_out/pmc_traffic.json profiles/pmc_traffic.json
cp gpurun_out/pmc_traffic.json gpurun_out/final/pmc_traffic.json
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/final/bench_hybrid.json 2> gpurun_out/final/bench_hybrid.err; echo "hybrid exit $?"
timeout 300 python bench.py --workload dense --steps 50 --warmup 5 --cpu-queries 0 > gpurun_out/final/bench_dense.json 2>/dev/null; echo "dense exit $?"
timeout 300 python bench.py --workload bm25 --steps 50 --warmup 5 --cpu-queries 0 > gpurun_out/final/bench_bm25.json 2>/dev/null; echo "bm25 exit $?"
timeout 300 python bench.py --workload hybrid --variant okapi --steps 20 --warmup 5 --cpu-queries 0 --sub 0 > gpurun_out/final/bench_hybrid_okapi.json 2>/dev/null; echo "okapi exit $?"
