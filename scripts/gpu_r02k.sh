#!/bin/bash
# r02k: whole-chip LDS-DMA streaming by layout (scripts/ubench/stream) + ablations / phase clocks of the lean ping-pong scan
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 120 scripts/ubench/stream > gpurun_out/ubench_stream.log 2>&1; echo "stream exit $?"; cat gpurun_out/ubench_stream.log
KB_PP=2 timeout 600 python scripts/kbench.py pp2 > gpurun_out/kbench_pp2.log 2>&1; echo "kbench exit $?"; cat gpurun_out/kbench_pp2.log | grep -v amdgpu.ids
