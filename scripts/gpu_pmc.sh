#!/bin/bash
# PMC passes (each in its own rocprofv3 run, --kernel-trace only) over a short dense-only bench.
set -u
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
WL=${1:-dense}
CMD="python $GRAFT_REPO_ROOT/bench.py --workload $WL --steps 3 --warmup 1 --cpu-queries 0"
cd /tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ|TCC|TCP|GRBM|TA|TD)_[A-Za-z0-9_]+" | sort -u > $GRAFT_REPO_ROOT/gpurun_out/pmc/counters.txt
wc -l $GRAFT_REPO_ROOT/gpurun_out/pmc/counters.txt
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAVES" \
           "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc/p$i -o p$i -- $CMD > $GRAFT_REPO_ROOT/gpurun_out/pmc/p$i.log 2>&1
  echo "pass $i exit $? ($set)"
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/pmc/p*/*counter_collection.csv')):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if not any(s in k for s in ('dense_scan', 'bm25_scan', 'seed_select', 'dense_finalize', 'cand_refine', 'fuse_kernel')): continue
        k = 'append' if 'append' in k else 'store' if 'store' in k else k.split('(')[0][-40:]
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
    for k, d in agg.items():
        print(f, k, {c: round(v / cnt[(k, c)], 1) for c, v in d.items()})
PY
