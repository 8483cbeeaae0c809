#!/bin/bash
# HBM traffic of each workload's dominant kernel class from the PMC counters: one rocprofv3 pass with FETCH_SIZE only
# (--kernel-trace, no other trace domain), corrected as MI355X_MICROARCH.md "HBM" prescribes (x2: 128-byte requests are
# tallied at 64 bytes on gfx950; FETCH_SIZE is in KiB).  Writes gpurun_out/pmc_traffic.json; copy it to
# profiles/pmc_traffic.json, where bench.py picks it up for roofline.traffic.
set -u
mkdir -p gpurun_out/traffic
export TMPDIR=/tmp
cd /tmp
for wl in hybrid dense bm25; do
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/traffic/$wl -o p -- \
    python $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 3 --warmup 1 --cpu-queries 0 --sub 0 > $GRAFT_REPO_ROOT/gpurun_out/traffic/$wl.log 2>&1
  echo "$wl exit $?"
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, json, collections, sys
sys.path.insert(0, ".")
from easyrag_amd import _build
out = {"_kernel_digest": _build._kernel_digest()}      # bench.py attaches the figures only to runs of exactly these kernels
import re
for wl, match, pat in (("hybrid", "dense_scan", r"dense_(scan|gemv)"), ("dense", "dense_scan", r"dense_(scan|gemv)"),
                       ("bm25", "bm25_scan", r"bm25_[wa]?scan")):
    f = glob.glob(f"gpurun_out/traffic/{wl}/**/*counter_collection.csv", recursive=True)
    if not f:
        continue
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if re.search(pat, r["Kernel_Name"]) and r["Counter_Name"] == "FETCH_SIZE":
            per[r["Kernel_Name"].split("(")[0][:80]].append(float(r["Counter_Value"]))
    vals = [v for vs in per.values() for v in vs]
    if not vals:
        continue
    out[wl] = {
        "kernel_class": match,
        "launches_profiled": len(vals),
        "fetch_size_kib_per_launch_raw": sum(vals) / len(vals),
        "hbm_bytes_per_launch": 2.0 * 1024.0 * sum(vals) / len(vals),
        "correction": "FETCH_SIZE [KiB] x 1024 x 2 (gfx950: 128-byte requests tallied at 64 bytes)",
        "per_kernel_kib": {k: sum(v) / len(v) for k, v in per.items()},
        "command": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python bench.py --workload {wl} --steps 3 --warmup 1 --cpu-queries 0 --sub 0",
    }
json.dump(out, open("gpurun_out/pmc_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
PY
