#!/bin/bash
set -u
OUT=gpurun_out/r06l
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_retrievers.py tests/test_reference_fixtures.py tests/test_gpu_sparse_fusion.py tests/test_gpu_dense_dir_blocks.py -m gpu -q --timeout 600 -p no:cacheprovider -x > $OUT/pytest.log 2>&1; echo "pytest exit $?"; grep -v amdgpu.ids $OUT/pytest.log | tail -3
timeout 900 python -m pytest tests/test_gpu_fullsize.py -k "filters_and_duplicate or dir_blocks or configs3_exact" -m gpu -q --timeout 600 -p no:cacheprovider > $OUT/pytest_full.log 2>&1; echo "pytest exit $?"; grep -v amdgpu.ids $OUT/pytest_full.log | tail -3
timeout 600 python scripts/b1_host_cost.py 300 > $OUT/b1_host_cost.log 2>&1; grep -v amdgpu.ids $OUT/b1_host_cost.log | grep -E "fused|dense top" 
