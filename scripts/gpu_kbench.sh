#!/bin/bash
# one kbench mode on the GPU box: bash scripts/gpu_kbench.sh <mode> <tag>
set -u
MODE=$1; TAG=${2:-r04}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python scripts/kbench.py $MODE > $OUT/kbench_$MODE.log 2>&1; echo "kbench $MODE exit $?"
grep -v amdgpu.ids $OUT/kbench_$MODE.log | tail -60
