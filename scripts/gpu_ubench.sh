#!/bin/bash
# runs one micro-benchmark binary of scripts/ubench on the GPU box: bash scripts/gpu_ubench.sh <name> <tag>
set -u
NAME=$1; TAG=${2:-r04}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 300 scripts/ubench/$NAME > $OUT/ubench_$NAME.log 2>&1; echo "$NAME exit $?"
cat $OUT/ubench_$NAME.log
