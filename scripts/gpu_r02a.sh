#!/bin/bash
# round 2, first GPU call: parity suite (new BM25 scan, slots, filters), BM25 + ping-pong measurements, bench lines
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python - <<'PY' > gpurun_out/build.log 2>&1
import __graft_entry__ as g
g.build()
PY
tail -2 gpurun_out/build.log
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 600 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?"; tail -25 gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"; tail -3 gpurun_out/smoke.log
timeout 600 python scripts/kbench.py bm25w > gpurun_out/kbench_bm25w.log 2>&1; echo "kbench bm25w exit $?"; cat gpurun_out/kbench_bm25w.log
timeout 600 python scripts/kbench.py pp2 > gpurun_out/kbench_pp2.log 2>&1; echo "kbench pp2 exit $?"; cat gpurun_out/kbench_pp2.log
timeout 600 python bench.py --steps 10 --warmup 2 > gpurun_out/bench_hybrid.json 2> gpurun_out/bench_hybrid.err; echo "bench hybrid exit $?"
tail -c 2500 gpurun_out/bench_hybrid.json; tail -3 gpurun_out/bench_hybrid.err
