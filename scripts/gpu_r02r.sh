#!/bin/bash
# r02r: seed prefix snapped to scan rounds; BM25 segments per query at B = 1024
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
for cfg in "hybrid:" "hybrid:--option bm25_segs=2" "hybrid:--option bm25_segs=3" "dense:" "hybrid:--option dense_n0_auto=0" "hybrid:" ; do
  wl=${cfg%%:*}; opt=${cfg#*:}
  timeout 600 python bench.py --workload $wl --steps 20 --warmup 3 --cpu-queries 0 $opt > gpurun_out/b.json 2> gpurun_out/b.err; python - "$wl $opt" <<PY
import json, sys
r=json.loads(open("gpurun_out/b.json").read().strip().splitlines()[-1])
print(sys.argv[1], "|", round(r["value"]), r["ms_per_step"], r["roofline"]["frac"], r["kernel_ms_per_step"])
PY
done
