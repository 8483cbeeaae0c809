#!/bin/bash
# Regenerates everything a round commits under profiles/ from the current sources, in ONE gpurun call:
#   bash scripts/gpu_round.sh r03z          (tag = file-name prefix under gpurun_out/round/; copy what is wanted to profiles/)
# GPU suite + smoke, the driver-style bench line (with sub_benchmarks), the per-workload lines, rocprofv3 kernel stats of
# the same commands, the BM25 / dense PMC passes, the FETCH_SIZE traffic table, the determinism screen, the BM25 kernel
# bench (section clocks) and the shim latency log.
# (the bench lines of this script are taken BEFORE its FETCH_SIZE pass: for lines with roofline.traffic attached run
# scripts/gpu_final.sh afterwards)
set -u
TAG=${1:-r00}
OUT=gpurun_out/round
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/${TAG}_build.log 2>&1; tail -1 $OUT/${TAG}_build.log
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > $OUT/${TAG}_pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -2 $OUT/${TAG}_pytest_gpu.log
timeout 300 python __graft_entry__.py smoke >> $OUT/${TAG}_pytest_gpu.log 2>&1; echo "smoke exit $?"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/${TAG}_bench_hybrid.json 2> $OUT/${TAG}_bench_hybrid.err; echo "bench hybrid exit $?"
timeout 300 python bench.py --workload dense --steps 50 --warmup 5 --cpu-queries 0 > $OUT/${TAG}_bench_dense.json 2>/dev/null; echo "bench dense exit $?"
timeout 300 python bench.py --workload bm25 --steps 50 --warmup 5 --cpu-queries 0 > $OUT/${TAG}_bench_bm25.json 2>/dev/null; echo "bench bm25 exit $?"
timeout 300 python bench.py --workload hybrid --variant okapi --steps 20 --warmup 5 --cpu-queries 0 --sub 0 > $OUT/${TAG}_bench_hybrid_okapi.json 2>/dev/null; echo "bench okapi exit $?"
for wl in hybrid dense bm25; do
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_$wl -o $wl -- \
     python $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 10 --warmup 2 --cpu-queries 0 --sub 0 > $GRAFT_REPO_ROOT/$OUT/prof_$wl.log 2>&1)
  f=$(find $OUT/prof_$wl -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && python scripts/trim_stats.py "$f" $OUT/${TAG}_${wl}_kernel_stats.csv > /dev/null
done
bash scripts/gpu_pmc.sh bm25 "--batch 1024 --sub 0" $TAG > $OUT/${TAG}_pmc_bm25.txt 2>&1
bash scripts/gpu_pmc.sh dense "--batch 1024 --sub 0" $TAG > $OUT/${TAG}_pmc_dense_b1024.txt 2>&1
bash scripts/gpu_pmc.sh dense "--batch 256 --sub 0" ${TAG}b > $OUT/${TAG}_pmc_dense_b256.txt 2>&1
bash scripts/gpu_traffic.sh > $OUT/${TAG}_traffic.log 2>&1; cp gpurun_out/pmc_traffic.json $OUT/pmc_traffic.json
timeout 600 python scripts/determinism.py 20 > $OUT/${TAG}_determinism.log 2>&1; echo "determinism exit $?"; tail -3 $OUT/${TAG}_determinism.log
timeout 600 python scripts/kbench.py bm25a > $OUT/${TAG}_kbench_bm25a.log 2>&1
timeout 400 python scripts/shim_latency.py 100000 > $OUT/${TAG}_shim_latency.log 2>&1
timeout 300 python scripts/small_batch.py > $OUT/${TAG}_small_batch.log 2>&1
timeout 200 scripts/ubench/scan_sync > $OUT/${TAG}_ubench_scan_sync.log 2>&1
ls $OUT | head -50
