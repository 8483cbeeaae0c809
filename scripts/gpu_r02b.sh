#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python - <<'PY' > gpurun_out/build.log 2>&1
import __graft_entry__ as g
g.build()
PY
tail -2 gpurun_out/build.log
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?"; tail -25 gpurun_out/pytest_gpu.log
timeout 600 python scripts/kbench.py pp2 > gpurun_out/kbench_pp2.log 2>&1; echo "kbench pp2 exit $?"; cat gpurun_out/kbench_pp2.log
