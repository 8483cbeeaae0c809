#!/bin/bash
set -u
OUT=gpurun_out/r06ze
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_dense.py tests/test_gpu_dense_tile384.py -m gpu -q --timeout 600 -p no:cacheprovider -x > $OUT/pytest.log 2>&1; echo "pytest exit $?"; grep -v amdgpu.ids $OUT/pytest.log | tail -2
timeout 600 python scripts/ab.py --workload dense --batch 1024 --k 288 --opt dense_fin_wgs=3,4 --reps 5 --steps 20 > $OUT/ab_b1024.log 2>&1
grep -v amdgpu.ids $OUT/ab_b1024.log | grep -E "^dense_fin" | cut -c1-230
timeout 600 python scripts/ab.py --workload dense --batch 256 --k 100 --opt dense_fin_wgs=3,4 --reps 5 --steps 30 > $OUT/ab_b256.log 2>&1
grep -v amdgpu.ids $OUT/ab_b256.log | grep -E "^dense_fin" | cut -c1-230
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o p -- python $GRAFT_REPO_ROOT/scripts/b1_profile.py filtered 200 > $GRAFT_REPO_ROOT/$OUT/prof.log 2>&1)
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && python scripts/trim_stats.py "$f" $OUT/b1_filtered_kernel_stats.csv > /dev/null
rm -rf $OUT/prof
head -9 $OUT/b1_filtered_kernel_stats.csv | cut -c1-120
