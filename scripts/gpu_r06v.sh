#!/bin/bash
set -u
OUT=gpurun_out/r06v
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python scripts/ab.py --workload dense --batch 256 --k 100 --opt dense_scan_nt=0,1 --reps 7 --steps 30 > $OUT/ab_scan_nt_b256.log 2>&1
grep -v amdgpu.ids $OUT/ab_scan_nt_b256.log | tail -5 | head -4 | cut -c1-200
timeout 600 python scripts/ab.py --workload dense --batch 128 --k 100 --opt dense_scan_nt=0,1 --reps 7 --steps 30 > $OUT/ab_scan_nt_b128.log 2>&1
grep -v amdgpu.ids $OUT/ab_scan_nt_b128.log | tail -5 | head -4 | cut -c1-200
timeout 600 python scripts/ab.py --workload hybrid --batch 1024 --dirs 4 --dir-layout block --opt dense_scan_nt=0,1 --reps 7 --steps 20 > $OUT/ab_scan_nt_dirs4_b1024.log 2>&1
grep -v amdgpu.ids $OUT/ab_scan_nt_dirs4_b1024.log | tail -5 | head -4 | cut -c1-300
