#!/bin/bash
set -u
OUT=gpurun_out/r06o
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_sparse_fusion.py -k "mixed or question_lengths or flood" -m gpu -q --timeout 600 -p no:cacheprovider -x > $OUT/pytest.log 2>&1; echo "pytest exit $?"; grep -v amdgpu.ids $OUT/pytest.log | tail -5
timeout 900 python -m pytest tests/test_gpu_fullsize.py -k "question_lengths" -m gpu -q --timeout 600 -p no:cacheprovider > $OUT/pytest_full.log 2>&1; echo "pytest exit $?"; grep -v amdgpu.ids $OUT/pytest_full.log | tail -5
timeout 800 python scripts/bm25_segs_probe.py mixed > $OUT/bm25_mixed_probe.log 2>&1; grep -v amdgpu.ids $OUT/bm25_mixed_probe.log | tail -20
