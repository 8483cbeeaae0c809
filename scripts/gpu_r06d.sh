#!/bin/bash
set -u
OUT=gpurun_out/r06d
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_sparse_fusion.py -k "reference_question_lengths or split" -m gpu -q --timeout 600 -p no:cacheprovider -x > $OUT/pytest_qlen.log 2>&1; echo "pytest exit $?"; grep -v amdgpu.ids $OUT/pytest_qlen.log | tail -4
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -k "reference_question_lengths" -m gpu -q -s --timeout 600 -p no:cacheprovider -x > $OUT/pytest_qlen_full.log 2>&1; echo "pytest exit $?"; grep -v amdgpu.ids $OUT/pytest_qlen_full.log | tail -4
timeout 1000 python scripts/bm25_qlen_probe.py > $OUT/bm25_qlen_probe_after.log 2>&1; grep -v amdgpu.ids $OUT/bm25_qlen_probe_after.log | head -9
