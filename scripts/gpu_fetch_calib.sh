#!/bin/bash
# FETCH_SIZE (and the TCC read-request counters) against kernels of known HBM volume: scripts/ubench/fetch_calib.hip.
# Each counter set in its own rocprofv3 run, --kernel-trace only.   bash scripts/gpu_fetch_calib.sh [tag]
set -u
TAG=${1:-r06}
OUT=gpurun_out/fetch_calib
mkdir -p $OUT
export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 scripts/ubench/fetch_calib.hip -o scripts/ubench/fetch_calib || exit 1
scripts/ubench/fetch_calib > $OUT/${TAG}_plain.log 2>&1; cat $OUT/${TAG}_plain.log
cd /tmp
i=0
for set in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $GRAFT_REPO_ROOT/$OUT/$TAG$i -o p -- $GRAFT_REPO_ROOT/scripts/ubench/fetch_calib > $GRAFT_REPO_ROOT/$OUT/$TAG$i.log 2>&1
  echo "pass $i exit $? ($set)"
done
cd $GRAFT_REPO_ROOT
python - "$OUT" "$TAG" <<'PY' | tee $OUT/${TAG}_fetch_calib.txt
import collections, csv, glob, sys
out, tag = sys.argv[1:3]
known = {"calib_stream16": 4 * 2**30, "calib_stream4": 4 * 2**30, "calib_stride128": 4 * 2**30, "calib_gather4": 4 * 2**30}
val = collections.defaultdict(dict)
for f in sorted(glob.glob(f"{out}/{tag}*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        for k in known:
            if k in r["Kernel_Name"]:
                val[k][r["Counter_Name"]] = val[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
print("kernel        known bytes (lines x 128)   FETCH_SIZE KiB   FETCH_SIZE x 1024 / known   EA read requests   of which 32 B   L2 hit share")
for k, kb in known.items():
    c = val.get(k, {})
    fs = c.get("FETCH_SIZE", float("nan"))
    rq, rq32 = c.get("TCC_EA0_RDREQ_sum", float("nan")), c.get("TCC_EA0_RDREQ_32B_sum", float("nan"))
    hit, miss = c.get("TCC_HIT_sum", 0.0), c.get("TCC_MISS_sum", 0.0)
    print(f"{k:14s} {kb:14d} {fs:18.0f} {fs * 1024 / kb:20.3f} {rq:22.0f} {rq32:14.0f} {hit / max(hit + miss, 1):12.3f}")
PY
