#!/bin/bash
set -u
OUT=gpurun_out/r06zh
mkdir -p $OUT
export TMPDIR=/tmp
for b in 256 128 64 16; do
timeout 600 python scripts/ab.py --workload hybrid --batch $b --opt hybrid_overlap=0,1,2 --reps 7 --steps 30 > $OUT/ab_overlap_b$b.log 2>&1
grep -v amdgpu.ids $OUT/ab_overlap_b$b.log | grep -E "^# A/B|^hybrid_overlap=[012] " | cut -c1-150
done
