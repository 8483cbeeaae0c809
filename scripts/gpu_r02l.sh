#!/bin/bash
# r02l: strict-alternation ping-pong (dense_pp=3): parity arm + timing against the lean kernel
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_dense.py -m gpu -q -x --timeout 420 -p no:cacheprovider -k "strict" > gpurun_out/pytest_pp3.log 2>&1; echo "pytest exit $?"; tail -5 gpurun_out/pytest_pp3.log
timeout 800 python scripts/kbench.py p3 2>&1 | grep -v amdgpu.ids | tee gpurun_out/kbench_p3.log
