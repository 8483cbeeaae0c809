#!/bin/bash
# PMC passes over the BM25-only workload: where do the scan kernel's wave cycles go?
set -u
mkdir -p gpurun_out/pmc3
export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --workload bm25 --steps 2 --warmup 1 --cpu-queries 0"
cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_FLAT SQ_INSTS_GDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc3/p$i -o p -- $CMD > $GRAFT_REPO_ROOT/gpurun_out/pmc3/p$i.log 2>&1
  echo "pass $i exit $? ($set)"
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/pmc3/p*/*counter_collection.csv')):
    agg = collections.defaultdict(float); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        if 'bm25_scan' not in r['Kernel_Name']: continue
        agg[r['Counter_Name']] += float(r['Counter_Value']); cnt[r['Counter_Name']] += 1
    print({c: round(v / cnt[c] / 1e6, 2) for c, v in agg.items()}, '(millions per launch)')
PY
