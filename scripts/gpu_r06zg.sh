#!/bin/bash
set -u
OUT=gpurun_out/r06zg
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python scripts/ab.py --workload hybrid --batch 1024 --opt fuse_threads=512,256 --reps 7 --steps 20 > $OUT/ab_fuse_threads_b1024.log 2>&1
grep -v amdgpu.ids $OUT/ab_fuse_threads_b1024.log | grep -E "^fuse_threads" | cut -c1-330
timeout 600 python scripts/ab.py --workload hybrid --batch 1 --dirs 4 --dir-layout block --opt fuse_threads=512,256 --reps 9 --steps 50 > $OUT/ab_fuse_threads_b1_dirs4.log 2>&1
grep -v amdgpu.ids $OUT/ab_fuse_threads_b1_dirs4.log | grep -E "^fuse_threads" | cut -c1-330
timeout 600 python scripts/ab.py --workload hybrid --batch 256 --opt fuse_threads=512,256 --reps 7 --steps 30 > $OUT/ab_fuse_threads_b256.log 2>&1
grep -v amdgpu.ids $OUT/ab_fuse_threads_b256.log | grep -E "^fuse_threads" | cut -c1-330
timeout 900 python -m pytest tests/test_gpu_sparse_fusion.py -k "rrf or fusion or hybrid" -m gpu -q --timeout 600 -p no:cacheprovider -x > $OUT/pytest.log 2>&1; echo "pytest exit $?"; grep -v amdgpu.ids $OUT/pytest.log | tail -2
