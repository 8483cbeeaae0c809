#!/bin/bash
# One gpurun call: GPU parity tests, smoke, the three bench workloads, and a rocprofv3 kernel trace.
# Usage (from the repo root on the GPU box): bash scripts/gpu_check.sh [tests|bench|prof|all]
set -u
what=${1:-all}
mkdir -p gpurun_out
export TMPDIR=/tmp
python - <<'PY' > gpurun_out/build.log 2>&1
import __graft_entry__ as g
g.build()
PY
tail -2 gpurun_out/build.log
if [[ "$what" == "all" || "$what" == "tests" ]]; then
  timeout 1500 python -m pytest tests -m gpu -q --timeout 420 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest exit $?"; tail -40 gpurun_out/pytest_gpu.log
  timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"; tail -3 gpurun_out/smoke.log
fi
if [[ "$what" == "all" || "$what" == "bench" ]]; then
  timeout 600 python bench.py --steps 10 --warmup 2 > gpurun_out/bench_hybrid.json 2> gpurun_out/bench_hybrid.err; echo "bench hybrid exit $?"
  tail -c 3000 gpurun_out/bench_hybrid.json; tail -5 gpurun_out/bench_hybrid.err
  timeout 400 python bench.py --workload dense --steps 20 --warmup 3 --cpu-queries 0 > gpurun_out/bench_dense.json 2> gpurun_out/bench_dense.err; echo "bench dense exit $?"
  tail -c 2000 gpurun_out/bench_dense.json; tail -5 gpurun_out/bench_dense.err
  timeout 400 python bench.py --workload bm25 --steps 10 --warmup 2 --cpu-queries 0 > gpurun_out/bench_bm25.json 2> gpurun_out/bench_bm25.err; echo "bench bm25 exit $?"
  tail -c 2000 gpurun_out/bench_bm25.json; tail -5 gpurun_out/bench_bm25.err
fi
if [[ "$what" == "all" || "$what" == "prof" ]]; then
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_hybrid -o hybrid -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --cpu-queries 0 > $GRAFT_REPO_ROOT/gpurun_out/prof_hybrid.log 2>&1); echo "rocprof exit $?"
  find gpurun_out/prof_hybrid -name "*stats*" | head; for f in $(find gpurun_out/prof_hybrid -name "*kernel_stats.csv"); do head -20 $f; done
fi
