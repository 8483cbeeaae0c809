#!/bin/bash
set -u
OUT=gpurun_out/r06t
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python scripts/ab.py --workload hybrid --batch 1 --opt dense_gemv_nt=0,1 --reps 15 --steps 50 > $OUT/ab_gemv_nt_hybrid_b1.log 2>&1
grep -v amdgpu.ids $OUT/ab_gemv_nt_hybrid_b1.log | tail -5 | head -4 | cut -c1-250
timeout 900 python scripts/ab.py --workload hybrid --batch 1 --opt dense_gemv_nt=0,1 --opt2 hybrid_overlap=0,1,2 --reps 9 --steps 50 > $OUT/ab_gemv_nt_hybrid_b1_overlap.log 2>&1
grep -v amdgpu.ids $OUT/ab_gemv_nt_hybrid_b1_overlap.log | grep -E "^dense_gemv" | cut -c1-200
timeout 900 python scripts/ab.py --workload hybrid --batch 16 --opt dense_gemv_nt=0,1 --reps 9 --steps 50 > $OUT/ab_gemv_nt_hybrid_b16.log 2>&1
grep -v amdgpu.ids $OUT/ab_gemv_nt_hybrid_b16.log | tail -5 | head -4 | cut -c1-250
