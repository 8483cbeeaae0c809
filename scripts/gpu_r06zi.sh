#!/bin/bash
# FETCH_SIZE / wave-cycle counters of the final kernel (dense_finalize_kernel) at 1024 queries, k = 288: what its row gathers move and how its waves spend their cycles
set -u
OUT=gpurun_out/r06zi
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --workload dense --batch 1024 --steps 2 --warmup 1 --cpu-queries 0 --sub 0"
cd /tmp
i=0
for set in "FETCH_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $GRAFT_REPO_ROOT/$OUT/p$i -o p -- $CMD > $GRAFT_REPO_ROOT/$OUT/p$i.log 2>&1
  echo "pass $i exit $? ($set)"
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import collections, csv, glob
for f in sorted(glob.glob("gpurun_out/r06zi/p*/*counter_collection.csv")):
    agg = collections.defaultdict(float); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        if "dense_finalize" not in r["Kernel_Name"]:
            continue
        agg[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
    print("dense_finalize_kernel", {c: round(v / cnt[c], 1) for c, v in agg.items()}, "launches", max(cnt.values()) if cnt else 0)
for f in sorted(glob.glob("gpurun_out/r06zi/p1/*kernel_trace.csv")):
    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if "dense_finalize" in r["Kernel_Name"]]
    print("durations ns", d)
PY
rm -rf $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4
