#!/bin/bash
set -u
OUT=gpurun_out/r06d
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1000 python scripts/bm25_qlen_probe.py 2>&1 | grep -v amdgpu.ids | head -1 | sed "s/^/long_shape=0 /"
for T in 16 24 31; do
  timeout 600 python scripts/bm25_qlen_probe.py $T 2>&1 | grep -v amdgpu.ids | head -1 | sed "s/^/long_tokens=$T /"
done
