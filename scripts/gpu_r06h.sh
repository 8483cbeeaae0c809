#!/bin/bash
set -u
OUT=gpurun_out/r06h
mkdir -p $OUT
export TMPDIR=/tmp
for B in 256 128 200; do
  timeout 600 python scripts/ab.py --workload dense --batch $B --k 100 --opt dense_selfseed=1,2 --reps 7 --steps 30 > $OUT/ab_selfseed_partial_b$B.log 2>&1; echo "== B=$B"; grep -v amdgpu.ids $OUT/ab_selfseed_partial_b$B.log | tail -4 | head -3
done
timeout 600 python scripts/ab.py --workload hybrid --batch 256 --opt dense_selfseed=1,2 --reps 5 --steps 30 > $OUT/ab_selfseed_partial_hybrid_b256.log 2>&1; echo "== hybrid B=256"; grep -v amdgpu.ids $OUT/ab_selfseed_partial_hybrid_b256.log | tail -4 | head -3
