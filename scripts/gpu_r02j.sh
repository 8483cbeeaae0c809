#!/bin/bash
# round-2 measurement refresh: bench lines, kernel stats, PMC traffic, PMC passes
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python - <<'PY' > gpurun_out/build.log 2>&1
import __graft_entry__ as g
g.build()
PY
tail -1 gpurun_out/build.log
for wl in hybrid dense bm25; do
  timeout 600 python bench.py --workload $wl --steps 20 --warmup 3 > gpurun_out/bench_$wl.json 2> gpurun_out/bench_$wl.err; echo "bench $wl exit $?"
done
timeout 600 python bench.py --workload hybrid --variant okapi --steps 10 --warmup 2 > gpurun_out/bench_hybrid_okapi.json 2> gpurun_out/bench_hybrid_okapi.err; echo "bench hybrid okapi exit $?"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_hybrid -o hybrid -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --cpu-queries 0 > $GRAFT_REPO_ROOT/gpurun_out/prof_hybrid.log 2>&1); echo "rocprof exit $?"
bash scripts/gpu_traffic.sh > gpurun_out/traffic.log 2>&1; echo "traffic exit $?"; tail -5 gpurun_out/traffic.log
bash scripts/gpu_pmc.sh dense "--batch 1024" r02 > gpurun_out/pmc_dense_b1024.txt 2>&1; tail -4 gpurun_out/pmc_dense_b1024.txt
bash scripts/gpu_pmc.sh bm25 "" r02 > gpurun_out/pmc_bm25.txt 2>&1; tail -6 gpurun_out/pmc_bm25.txt
for f in gpurun_out/bench_*.json; do echo $f; tail -c 900 $f; echo; done
