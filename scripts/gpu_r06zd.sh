#!/bin/bash
set -u
OUT=gpurun_out/r06zd
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python scripts/ab.py --workload dense --batch 1024 --k 288 --opt dense_fin_wgs=2,3,4 --reps 7 --steps 20 > $OUT/ab_fin_wgs4_b1024.log 2>&1
grep -v amdgpu.ids $OUT/ab_fin_wgs4_b1024.log | grep -E "^dense_fin|delta" | cut -c1-230
timeout 600 python scripts/ab.py --workload dense --batch 256 --k 100 --opt dense_fin_wgs=2,3,4 --reps 7 --steps 30 > $OUT/ab_fin_wgs4_b256.log 2>&1
grep -v amdgpu.ids $OUT/ab_fin_wgs4_b256.log | grep -E "^dense_fin|delta" | cut -c1-200
timeout 600 python scripts/ab.py --workload dense --batch 512 --k 288 --opt dense_fin_wgs=2,3,4 --reps 7 --steps 30 > $OUT/ab_fin_wgs4_b512.log 2>&1
grep -v amdgpu.ids $OUT/ab_fin_wgs4_b512.log | grep -E "^dense_fin|delta" | cut -c1-200
timeout 600 python scripts/ab.py --workload hybrid --batch 1024 --dirs 4 --dir-layout block --opt dense_fin_wgs=2,3,4 --reps 7 --steps 20 > $OUT/ab_fin_wgs4_dirs4.log 2>&1
grep -v amdgpu.ids $OUT/ab_fin_wgs4_dirs4.log | grep -E "^dense_fin|delta" | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_dense.py -m gpu -q --timeout 600 -p no:cacheprovider -x > $OUT/pytest.log 2>&1; echo "pytest exit $?"; grep -v amdgpu.ids $OUT/pytest.log | tail -2
