#!/usr/bin/env python
"""Register / scratch / occupancy table of the kernels of one translation unit (hipcc cross-compiles: no GPU needed).
  python scripts/resusage.py bm25.hip [regex over the demangled name] [extra hipcc flags, e.g. -DERH_MEASURE]"""
import os
import re
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 else "."
extra = sys.argv[3:]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
       "-I", os.path.join(root, "include"), "-I", os.path.join(root, "easyrag_amd", "csrc"), *extra, "-c",
       os.path.join(root, "easyrag_amd", "csrc", src), "-o", f"/tmp/resusage_{os.getpid()}.o", "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, capture_output=True, text=True)
rows, cur = {}, None
pat = re.compile(r"remark:\s*(Function Name|VGPRs|AGPRs|SGPRs Spill|VGPRs Spill|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]): (\S+)")
for line in out.stderr.splitlines():
    m = pat.search(line)
    if not m:
        continue
    k, v = m.groups()
    if k == "Function Name":
        cur = v
        rows[cur] = {}
    elif cur:
        rows[cur][k] = v
if out.returncode != 0 and not rows:
    sys.stderr.write(out.stderr[-4000:])
    sys.exit(1)
for name, r in rows.items():
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = dem.replace("(anonymous namespace)::", "").split("(")[0][:100]
    if not re.search(filt, dem):
        continue
    print("%-100s vgpr %3s agpr %3s sgpr-spill %3s vgpr-spill %3s scratch %4s occ %s" % (
        dem, r.get("VGPRs", "?"), r.get("AGPRs", "?"), r.get("SGPRs Spill", "?"), r.get("VGPRs Spill", "?"),
        r.get("ScratchSize [bytes/lane]", "?"), r.get("Occupancy [waves/SIMD]", "?")))
try:
    os.remove(f"/tmp/resusage_{os.getpid()}.o")
except OSError:
    pass
