#!/bin/bash
set -u
OUT=gpurun_out/r06y
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_gpu_dense.py tests/test_gpu_dense_dir_blocks.py -k "cache_policy or mixed_batch" -m gpu -q --timeout 600 -p no:cacheprovider -x > $OUT/pytest.log 2>&1; echo "pytest exit $?"; grep -v amdgpu.ids $OUT/pytest.log | tail -3
