#!/bin/bash
# Round 6: the grouped launch (one launch per stage over all dir groups) against the per-group pipelines and the filter column.
set -u
OUT=gpurun_out/r06a
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_fullsize.py -k "dir_blocks or configs1" -m gpu -q --timeout 600 -p no:cacheprovider > $OUT/pytest_fullsize.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/pytest_fullsize.log
for B in 1024 512 256 64; do
  timeout 600 python scripts/ab.py --workload hybrid --batch $B --dirs 4 --dir-layout block --opt dense_dir_blocks=0,2 --opt2 dense_group_launch=0,1 --reps 5 --steps 20 > $OUT/ab_dirs4_b$B.log 2>&1
  echo "== B=$B"; tail -12 $OUT/ab_dirs4_b$B.log
done
timeout 600 python scripts/ab.py --workload hybrid --batch 1024 --dirs 12 --dir-layout block --opt dense_dir_blocks=0,2 --opt2 dense_group_launch=0,1 --reps 5 --steps 20 > $OUT/ab_dirs12_b1024.log 2>&1
echo "== 12 dirs"; tail -12 $OUT/ab_dirs12_b1024.log
