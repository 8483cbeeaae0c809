#!/bin/bash
# PMC passes over one bench workload (each counter set in its own rocprofv3 run with --kernel-trace only, as the
# MI355X guide prescribes): where the dominant kernels' time goes.
#   bash scripts/gpu_pmc.sh dense "--batch 1024" tag     dense scan kernels (HBM fetch, L2, MFMA busy, LDS, TA/TD)
#   bash scripts/gpu_pmc.sh bm25  ""             tag     BM25 scan kernel (wave cycles / waits, instruction mix, LDS)
set -u
WL=${1:-dense}
OPTS=${2:-}
TAG=${3:-a}
OUT=gpurun_out/pmc_$WL
mkdir -p $OUT
export TMPDIR=/tmp
# (WL "hybrid": the dense kernels of the hybrid step are summarised -- e.g. "--dirs 4 --sub 0", the filtered path of the reference's real calls)
CMD="python $GRAFT_REPO_ROOT/bench.py --workload $WL --steps 2 --warmup 1 --cpu-queries 0 $OPTS"
if [[ "$WL" == "bm25" ]]; then
  SETS=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE"
        "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_WAIT_INST_LDS"
        "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_FLAT SQ_INSTS_GDS"
        "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM"
        "FETCH_SIZE"
        "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TD_TD_BUSY_sum"
        "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum")
else
  SETS=("FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
        "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
        "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"
        "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TD_TD_BUSY_sum")
fi
cd /tmp
i=0
for set in "${SETS[@]}"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $GRAFT_REPO_ROOT/$OUT/$TAG$i -o p -- $CMD > $GRAFT_REPO_ROOT/$OUT/$TAG$i.log 2>&1
  echo "pass $i exit $? ($set)"
done
cd $GRAFT_REPO_ROOT
python - "$WL" "$TAG" "$OUT" <<'PY'
import collections, csv, glob, re, sys
wl, tag, out = sys.argv[1:4]
pat = re.compile(r"bm25_[wa]?scan" if wl == "bm25" else r"dense_(scan|gemv)")
def klass(name):
    if wl == "bm25":
        return "ascan" if "ascan" in name else ("wscan" if "wscan" in name else "scan")
    m = re.search(r"pp3_kernel(?:<\d+, |ILi\d+ELi)(\d+)", name)
    if m and int(m.group(1)) & 16:
        return "pp3_sample"                      # the sample pass (VAR bit 4; 48 / 56: of the grouped launch) has a timing class of its own
    for key in ("pp5", "pp3", "pp2", "_pp_", "persist", "append", "store", "gemv"):
        if key in name:
            return key.strip("_")
    return "other"
for f in sorted(glob.glob(f"{out}/{tag}*/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        if not pat.search(r["Kernel_Name"]):
            continue
        k = klass(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    for k, d in agg.items():
        print(tag, wl, k, {c: round(v / cnt[(k, c)], 1) for c, v in d.items()}, "launches", max(cnt[(k, c)] for c in d))
PY
