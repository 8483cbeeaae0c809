#!/bin/bash
# Round-6 evidence in ONE gpurun lease, in the order that lets the bench lines carry their counters:
#   1. full -m gpu suite + smoke
#   2. PMC passes (dense at 1024 / 256 queries, BM25 at 1024 / 256, the FILTERED hybrid step) and the FETCH_SIZE traffic pass
#      -> profiles/pmc_counters.json, profiles/pmc_traffic.json ON THE BOX (keyed by the kernel digest), copies under gpurun_out/round/
#   3. FETCH_SIZE calibration against kernels of known volume (scripts/ubench/fetch_calib.hip)
#   4. the driver-style bench line (sub_benchmarks, cpu_baseline) and the per-workload lines -- with roofline.traffic / counters
#      attached -- and the rocprofv3 --kernel-trace --stats summaries of the same commands, the filtered step included
#   5. determinism screen, small-batch latencies
#   bash scripts/gpu_round6.sh r06z      (copy gpurun_out/round/<tag>_* and the two json tables to profiles/ afterwards)
set -u
TAG=${1:-r06z}
WHAT=${2:-all}
OUT=gpurun_out/round
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/${TAG}_build.log 2>&1; tail -1 $OUT/${TAG}_build.log
if [[ "$WHAT" == "all" || "$WHAT" == "tests" ]]; then
  timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > $OUT/${TAG}_pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -2 $OUT/${TAG}_pytest_gpu.log
  timeout 300 python __graft_entry__.py smoke >> $OUT/${TAG}_pytest_gpu.log 2>&1; echo "smoke exit $?"
fi
if [[ "$WHAT" == "all" || "$WHAT" == "pmc" ]]; then
  bash scripts/gpu_pmc.sh dense "--batch 1024 --sub 0" $TAG > $OUT/${TAG}_pmc_dense_b1024.txt 2>&1
  bash scripts/gpu_pmc.sh dense "--batch 256 --sub 0" ${TAG}b > $OUT/${TAG}_pmc_dense_b256.txt 2>&1
  bash scripts/gpu_pmc.sh bm25 "--batch 1024 --sub 0" $TAG > $OUT/${TAG}_pmc_bm25.txt 2>&1
  bash scripts/gpu_pmc.sh bm25 "--batch 256 --sub 0" ${TAG}b > $OUT/${TAG}_pmc_bm25_b256.txt 2>&1
  bash scripts/gpu_pmc.sh hybrid "--dirs 4 --sub 0" ${TAG}f > $OUT/${TAG}_pmc_filtered.txt 2>&1
  python scripts/pmc_summary.py dense_b1024=$OUT/${TAG}_pmc_dense_b1024.txt dense_b256=$OUT/${TAG}_pmc_dense_b256.txt \
         bm25_b1024=$OUT/${TAG}_pmc_bm25.txt bm25_b256=$OUT/${TAG}_pmc_bm25_b256.txt hybrid_dirs4_b1024=$OUT/${TAG}_pmc_filtered.txt \
         > profiles/pmc_counters.json; cp profiles/pmc_counters.json $OUT/pmc_counters.json
  bash scripts/gpu_traffic.sh > $OUT/${TAG}_traffic.log 2>&1; cp gpurun_out/pmc_traffic.json profiles/pmc_traffic.json; cp gpurun_out/pmc_traffic.json $OUT/pmc_traffic.json
  bash scripts/gpu_fetch_calib.sh $TAG > $OUT/${TAG}_fetch_calib.log 2>&1; tail -6 $OUT/${TAG}_fetch_calib.log
fi
if [[ "$WHAT" == "all" || "$WHAT" == "bench" ]]; then
  timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/${TAG}_bench_hybrid.json 2> $OUT/${TAG}_bench_hybrid.err; echo "bench hybrid exit $?"
  timeout 300 python bench.py --workload dense --steps 50 --warmup 5 --cpu-queries 0 > $OUT/${TAG}_bench_dense.json 2>/dev/null; echo "bench dense exit $?"
  timeout 300 python bench.py --workload bm25 --steps 50 --warmup 5 --cpu-queries 0 > $OUT/${TAG}_bench_bm25.json 2>/dev/null; echo "bench bm25 exit $?"
  timeout 300 python bench.py --workload hybrid --variant okapi --steps 20 --warmup 5 --cpu-queries 0 --sub 0 > $OUT/${TAG}_bench_hybrid_okapi.json 2>/dev/null; echo "bench okapi exit $?"
  timeout 300 python bench.py --workload hybrid --dirs 4 --steps 20 --warmup 5 --cpu-queries 0 --sub 0 > $OUT/${TAG}_bench_filtered.json 2>/dev/null; echo "bench filtered exit $?"
  timeout 300 python bench.py --workload hybrid --corpus clustered --qlen ref --steps 20 --warmup 5 --cpu-queries 0 --sub 0 > $OUT/${TAG}_bench_clustered_reflen.json 2>/dev/null; echo "bench clustered+ref lengths exit $?"
  for wl in hybrid dense bm25 filtered; do
    extra=""; w=$wl
    if [[ "$wl" == "filtered" ]]; then w=hybrid; extra="--dirs 4"; fi
    (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_$wl -o $wl -- \
       python $GRAFT_REPO_ROOT/bench.py --workload $w $extra --steps 10 --warmup 2 --cpu-queries 0 --sub 0 > $GRAFT_REPO_ROOT/$OUT/prof_$wl.log 2>&1)
    f=$(find $OUT/prof_$wl -name "*kernel_stats.csv" | head -1)
    [ -n "$f" ] && python scripts/trim_stats.py "$f" $OUT/${TAG}_${wl}_kernel_stats.csv > /dev/null
  done
  rm -rf $OUT/prof_hybrid $OUT/prof_dense $OUT/prof_bm25 $OUT/prof_filtered
  timeout 600 python scripts/determinism.py 20 > $OUT/${TAG}_determinism.log 2>&1; echo "determinism exit $?"; tail -3 $OUT/${TAG}_determinism.log
  timeout 300 python scripts/small_batch.py > $OUT/${TAG}_small_batch.log 2>&1
fi
python - "$TAG" <<'PY'
import json, sys
tag = sys.argv[1]
for wl in ("hybrid", "dense", "bm25", "hybrid_okapi", "filtered", "clustered_reflen"):
    try:
        r = json.loads(open(f"gpurun_out/round/{tag}_bench_{wl}.json").read().strip().splitlines()[-1])
        ro = r.get("roofline") or {}
        print(wl, round(r["value"]), "q/s", round(r["ms_per_step"], 4), "ms; frac", round(ro.get("frac", 0), 4), "traffic x", ro.get("traffic_over_algorithmic"),
              "counters", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in (ro.get("counters") or {}).items() if k in ("mfma_busy", "l2_hit", "td_busy", "effective_clock_ghz")},
              "kernels", {k: round(v, 4) for k, v in r["kernel_ms_per_step"].items() if v}, "path", {k: v for k, v in (r.get("path") or {}).items() if v})
    except Exception as e:
        print(wl, "unreadable", e)
PY
ls $OUT | head -80
