#!/bin/bash
set -u
OUT=gpurun_out/r06zm
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_gpu_dense.py tests/test_gpu_dense_dir_blocks.py tests/test_gpu_dense_tile384.py tests/test_gpu_retrievers.py -m gpu -q --timeout 600 -p no:cacheprovider -x > $OUT/pytest.log 2>&1; echo "pytest exit $?"; grep -v amdgpu.ids $OUT/pytest.log | tail -4
timeout 600 python scripts/b1_host_cost.py 300 > $OUT/b1_host_cost.log 2>&1; grep -v amdgpu.ids $OUT/b1_host_cost.log | grep -E "check per call"
