#!/bin/bash
# r02t: register-resident seed rows: parity subset + kernel trace
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_dense.py -m gpu -q -x --timeout 420 -p no:cacheprovider -k "strict or gemv or seed or budgets or sorted or speculation or filter" > gpurun_out/pytest_dense.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/pytest_dense.log
bash scripts/gpu_r02p.sh | grep -E "pp3|wscan|finalize|seed_select|store|rocprof"
