#!/usr/bin/env python
"""PMC passes (scripts/gpu_pmc.sh text summaries) -> profiles/pmc_counters.json: the per-launch counter averages of the dense scan
kernels and what north_star asks to be reported against the gfx950 peaks -- MFMA-busy, L2 hit rate, the LDS-fill (TD) path --
keyed by the digest of the kernel sources, so that bench.py attaches them to `roofline.counters` only for runs of exactly the
profiled kernels (as it does with pmc_traffic.json).

  python scripts/pmc_summary.py dense_b1024=gpurun_out/round/r05z_pmc_dense_b1024.txt dense_b256=gpurun_out/round/r05z_pmc_dense_b256.txt \\
                                bm25_b1024=gpurun_out/round/r05z_pmc_bm25.txt > profiles/pmc_counters.json

Formulas (MI355X: 8 XCDs x 32 CUs x 4 SIMDs): GRBM_GUI_ACTIVE is summed over the XCDs, so cycles per XCD = GUI / 8;
mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GUI / 8); td_busy = TD_TD_BUSY_sum / 256 CUs / (GUI / 8);
l2_hit = TCC_HIT / (TCC_HIT + TCC_MISS); waves_waiting = SQ_WAIT_ANY / SQ_WAVE_CYCLES.  The effective shader clock needs the
kernel's duration: bench.py divides the cycles per XCD by its own HIP-event time of the class."""
import ast
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from easyrag_amd import _build  # noqa: E402

LINE = re.compile(r"^(\S+) (\S+) (\S+) (\{.*\}) launches (\d+)\s*$")


def parse(path):
    classes = {}
    for line in open(path):
        m = LINE.match(line.strip())
        if not m:
            continue
        _, wl, klass, d, n = m.groups()
        rec = classes.setdefault(klass, {"launches_profiled": int(n)})
        rec.update(ast.literal_eval(d))
    return classes


def derive(c):
    out = {}
    gui = c.get("GRBM_GUI_ACTIVE")
    if gui:
        per_xcd = gui / 8.0
        out["gui_cycles_per_xcd"] = per_xcd
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            out["mfma_busy"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / per_xcd
        if "SQ_BUSY_CYCLES" in c:
            out["sq_busy"] = c["SQ_BUSY_CYCLES"] / 32.0 / per_xcd if c["SQ_BUSY_CYCLES"] / 32.0 / per_xcd <= 1.5 else None
    if c.get("SQ_WAVE_CYCLES"):
        for k, name in (("SQ_WAIT_ANY", "waves_waiting"), ("SQ_WAIT_INST_ANY", "waves_issue_stalled"), ("SQ_ACTIVE_INST_ANY", "waves_issuing")):
            if k in c:
                out[name] = c[k] / c["SQ_WAVE_CYCLES"]
    if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c and c["TCC_HIT_sum"] + c["TCC_MISS_sum"] > 0:
        out["l2_hit"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    if "FETCH_SIZE" in c:
        out["hbm_bytes_per_launch"] = c["FETCH_SIZE"] * 1024.0 * 2.0          # KiB, x2: 128-byte requests tallied at 64 bytes on gfx950
    if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE"):
        out["lds_bank_conflict_share"] = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]
    return out


def main():
    out = {"_kernel_digest": _build._kernel_digest(),
           "_what": "rocprofv3 --kernel-trace --pmc <one counter set per run> over bench.py --steps 2 --warmup 1 (scripts/gpu_pmc.sh); per-launch averages"}
    for arg in sys.argv[1:]:
        key, path = arg.split("=", 1)
        classes = parse(path)
        if not classes:
            continue
        entry = {"source": os.path.basename(path), "classes": {}}
        for klass, c in classes.items():
            entry["classes"][klass] = {"raw": c, "derived": derive(c)}
        # TD / TA per class need the class's own GUI cycles (collected in another pass of the same kernels)
        for klass, rec in entry["classes"].items():
            per_xcd = rec["derived"].get("gui_cycles_per_xcd")
            if per_xcd and "TD_TD_BUSY_sum" in rec["raw"]:
                rec["derived"]["td_busy"] = rec["raw"]["TD_TD_BUSY_sum"] / 256.0 / per_xcd
            if per_xcd and "TA_TA_BUSY_sum" in rec["raw"]:
                rec["derived"]["ta_busy"] = rec["raw"]["TA_TA_BUSY_sum"] / 256.0 / per_xcd
        out[key] = entry
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
