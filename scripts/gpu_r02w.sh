#!/bin/bash
# r02w: full GPU parity suite + smoke on the final build
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python - <<'PY' > gpurun_out/build.log 2>&1
import __graft_entry__ as g
g.build()
PY
tail -1 gpurun_out/build.log
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -4 gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"; tail -1 gpurun_out/smoke.log
timeout 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench default exit $?"; python - <<'PY'
import json
r = json.loads(open("gpurun_out/bench_default.json").read().strip().splitlines()[-1])
print(round(r["value"]), r["ms_per_step"], r["steps"], r["roofline"]["frac"], r["roofline"].get("traffic_over_algorithmic"), r["kernel_ms_per_step"])
PY
