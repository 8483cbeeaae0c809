#!/bin/bash
# round 4, call a: configs[1] (256 queries) ablation round of the current scan + the micro-benchmark ceilings on the same box
set -u
OUT=gpurun_out/r04a
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python scripts/kbench.py b256 > $OUT/kbench_b256.log 2>&1; echo "kbench exit $?"
timeout 200 scripts/ubench/mfma_dma > $OUT/ubench_mfma_dma.log 2>&1; echo "mfma_dma exit $?"
timeout 200 scripts/ubench/stream > $OUT/ubench_stream.log 2>&1; echo "stream exit $?"
tail -40 $OUT/kbench_b256.log
