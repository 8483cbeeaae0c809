#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python - <<'PY' > gpurun_out/build.log 2>&1
import __graft_entry__ as g
g.build()
PY
tail -1 gpurun_out/build.log
timeout 1700 python -m pytest tests/test_gpu_dense.py -m gpu -q -x --timeout 900 -p no:cacheprovider -k "pingpong or golden" > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?"; tail -4 gpurun_out/pytest_gpu.log
KB_PP=2 timeout 600 python scripts/kbench.py pp2q > gpurun_out/kbench_pp2lean.log 2>&1; echo "kbench exit $?"; grep -E "pabl=0|pabl=7 " gpurun_out/kbench_pp2lean.log
