#!/bin/bash
set -u
OUT=gpurun_out/r06f
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_dense_dir_blocks.py tests/test_gpu_dense_tile384.py tests/test_gpu_dist.py -m gpu -q --timeout 600 -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest exit $?"; grep -v amdgpu.ids $OUT/pytest.log | tail -15
timeout 900 python -m pytest tests/test_gpu_fullsize.py -k "dir_blocks" -m gpu -q --timeout 600 -p no:cacheprovider > $OUT/pytest_full.log 2>&1; echo "pytest exit $?"; grep -v amdgpu.ids $OUT/pytest_full.log | tail -5
timeout 600 python scripts/ab.py --workload hybrid --batch 1024 --dirs 4 --dir-layout block --opt dense_group_sample=0,1 --reps 5 --steps 20 > $OUT/ab_group_sample_b1024.log 2>&1; grep -v amdgpu.ids $OUT/ab_group_sample_b1024.log | tail -5
timeout 600 python scripts/ab.py --workload hybrid --batch 512 --dirs 4 --dir-layout block --opt dense_group_sample=0,1 --reps 5 --steps 20 > $OUT/ab_group_sample_b512.log 2>&1; grep -v amdgpu.ids $OUT/ab_group_sample_b512.log | tail -5
timeout 600 python scripts/ab.py --workload hybrid --batch 1024 --dirs 12 --dir-layout block --opt dense_group_sample=0,1 --reps 5 --steps 20 > $OUT/ab_group_sample_dirs12.log 2>&1; grep -v amdgpu.ids $OUT/ab_group_sample_dirs12.log | tail -5
