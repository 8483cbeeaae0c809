#!/bin/bash
# Focused PMC comparison of the dense append-scan kernels at B=1024 (hybrid-sized batch, dense-only workload):
# HBM fetch bytes, L2 hit/miss, MFMA busy.  Usage: bash scripts/gpu_pmc2.sh "<bench options>" tag
set -u
OPTS=${1:-}
TAG=${2:-a}
mkdir -p gpurun_out/pmc2
export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --workload dense --batch 1024 --steps 2 --warmup 1 --cpu-queries 0 $OPTS"
cd /tmp
i=0
for set in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" \
           "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TD_TD_BUSY_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc2/$TAG$i -o p -- $CMD > $GRAFT_REPO_ROOT/gpurun_out/pmc2/$TAG$i.log 2>&1
  echo "pass $i exit $? ($set)"
done
cd $GRAFT_REPO_ROOT
python - "$TAG" <<'PY'
import csv, glob, collections, sys
tag = sys.argv[1]
for f in sorted(glob.glob(f'gpurun_out/pmc2/{tag}*/*counter_collection.csv')):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'dense_scan' not in k: continue
        k = 'pp' if '_pp_' in k else 'persist' if 'persist' in k else 'append' if 'append' in k else 'store'
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
    for k, d in agg.items():
        print(tag, k, {c: round(v / cnt[(k, c)], 1) for c, v in d.items()}, 'launches', max(cnt[(k, c)] for c in d))
PY
