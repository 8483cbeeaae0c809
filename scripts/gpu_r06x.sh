#!/bin/bash
set -u
OUT=gpurun_out/r06x
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python scripts/ab.py --workload dense --batch 1024 --k 288 --opt dense_fin_wgs=2,3 --reps 7 --steps 20 > $OUT/ab_fin_wgs_b1024.log 2>&1
grep -v amdgpu.ids $OUT/ab_fin_wgs_b1024.log | tail -5 | head -4 | cut -c1-230
timeout 600 python scripts/ab.py --workload dense --batch 256 --k 100 --opt dense_fin_wgs=2,3 --reps 7 --steps 30 > $OUT/ab_fin_wgs_b256.log 2>&1
grep -v amdgpu.ids $OUT/ab_fin_wgs_b256.log | tail -5 | head -4 | cut -c1-200
timeout 600 python scripts/ab.py --workload hybrid --batch 1024 --dirs 4 --dir-layout block --opt dense_fin_wgs=2,3 --reps 7 --steps 20 > $OUT/ab_fin_wgs_dirs4.log 2>&1
grep -v amdgpu.ids $OUT/ab_fin_wgs_dirs4.log | tail -5 | head -4 | cut -c1-300
