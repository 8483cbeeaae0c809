#!/bin/bash
# Round 6: BM25 exact re-score + rank as a batch-wide kernel behind the scan (bm25_split_finish) -- parity arms, then interleaved A/B.
set -u
OUT=gpurun_out/r06b
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_sparse_fusion.py -k "split or packed-4byte" -m gpu -q --timeout 600 -p no:cacheprovider -x > $OUT/pytest_sparse.log 2>&1; echo "pytest exit $?"; tail -5 $OUT/pytest_sparse.log
for B in 1024 256; do
  timeout 600 python scripts/ab.py --workload bm25 --batch $B --k $([ $B = 1024 ] && echo 192 || echo 100) --opt bm25_split_finish=0,1 --reps 7 --steps 20 > $OUT/ab_bm25_split_b$B.log 2>&1
  echo "== bm25 B=$B"; tail -6 $OUT/ab_bm25_split_b$B.log
done
timeout 600 python scripts/ab.py --workload hybrid --batch 1024 --opt bm25_split_finish=0,1 --reps 5 --steps 20 > $OUT/ab_hybrid_split.log 2>&1
echo "== hybrid"; tail -6 $OUT/ab_hybrid_split.log
timeout 600 python scripts/ab.py --workload hybrid --batch 1024 --dirs 4 --dir-layout block --opt bm25_split_finish=0,1 --reps 5 --steps 20 > $OUT/ab_hybrid_dirs_split.log 2>&1
echo "== hybrid dirs"; tail -6 $OUT/ab_hybrid_dirs_split.log
