#!/bin/bash
# runs an arbitrary command line on the GPU box with its output kept: bash scripts/gpu_cmd.sh <tag> <command ...>
set -u
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
( eval "$@" ) > $OUT/cmd.log 2>&1; echo "exit $?" | tee -a $OUT/cmd.log
grep -v amdgpu.ids $OUT/cmd.log | tail -${TAIL:-60}
