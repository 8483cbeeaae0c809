#!/bin/bash
set -u
OUT=gpurun_out/r06s
mkdir -p $OUT
export TMPDIR=/tmp
for b in 2 4 8 12 16 24 48; do
  timeout 600 python scripts/ab.py --workload dense --batch $b --k 288 --opt dense_gemv_nt=0,1 --reps 5 --steps 50 > $OUT/ab_gemv_nt_b$b.log 2>&1
  grep -v amdgpu.ids $OUT/ab_gemv_nt_b$b.log | grep -E "^# A/B|delta"
done
timeout 600 python scripts/ab.py --workload hybrid --batch 1 --opt dense_gemv_nt=0,1 --reps 7 --steps 50 > $OUT/ab_gemv_nt_hybrid_b1.log 2>&1
grep -v amdgpu.ids $OUT/ab_gemv_nt_hybrid_b1.log | tail -5 | head -4
timeout 600 python scripts/ab.py --workload hybrid --batch 1 --dirs 4 --dir-layout block --opt dense_gemv_nt=0,1 --reps 7 --steps 50 > $OUT/ab_gemv_nt_hybrid_b1_dirs4.log 2>&1
grep -v amdgpu.ids $OUT/ab_gemv_nt_hybrid_b1_dirs4.log | tail -5 | head -4
