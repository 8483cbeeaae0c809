#!/bin/bash
set -u
OUT=gpurun_out/r06i
mkdir -p $OUT
export TMPDIR=/tmp
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o b1 -- python $GRAFT_REPO_ROOT/scripts/b1_profile.py filtered 200 > $GRAFT_REPO_ROOT/$OUT/prof.log 2>&1)
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
python scripts/trim_stats.py "$f" $OUT/b1_filtered_kernel_stats.csv | head -40
# the order and gaps of one call: the last call's kernels from the trace
t=$(find $OUT/prof -name "*kernel_trace.csv" | head -1)
python - "$t" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = rows[-40:]
t0 = int(last[0]["Start_Timestamp"])
for r in last:
    n = r["Kernel_Name"]
    n = n.split("(")[0][-60:]
    print(f'{(int(r["Start_Timestamp"]) - t0) / 1e3:9.1f} us  +{(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:7.1f} us  q{r.get("Queue_Id", "?")}  {n}')
PY
rm -rf $OUT/prof
