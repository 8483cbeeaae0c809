#!/bin/bash
set -u
OUT=gpurun_out/r06r
mkdir -p $OUT
export TMPDIR=/tmp
for b in 1 16 32 64; do
  timeout 600 python scripts/ab.py --workload dense --batch $b --k 288 --opt dense_gemv_nt=0,1 --reps 7 --steps 50 > $OUT/ab_gemv_nt_b$b.log 2>&1
  grep -v amdgpu.ids $OUT/ab_gemv_nt_b$b.log | tail -6
done
timeout 600 python scripts/ab.py --workload dense --batch 1 --k 288 --dirs 4 --dir-layout block --opt dense_gemv_nt=0,1 --reps 7 --steps 50 > $OUT/ab_gemv_nt_b1_dirs4.log 2>&1
grep -v amdgpu.ids $OUT/ab_gemv_nt_b1_dirs4.log | tail -6
