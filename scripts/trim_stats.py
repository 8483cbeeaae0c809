#!/usr/bin/env python
"""Condense a rocprofv3 *_kernel_stats.csv to this library's kernels (drops torch's data-generation kernels)."""
import csv
import re
import sys

OURS = ("dense_scan", "dense_gemv", "dense_exact", "dense_bad", "dense_naive", "bm25_", "seed_select", "cand_refine",
        "dense_finalize", "fuse_kernel", "pack_topk", "unpack_topk", "csr_", "prep_queries", "convert_rows",
        "permute_rows", "row_norm", "widen_f32")


def short(name: str) -> str:
    m = re.search(r"(" + "|".join(OURS) + r")[A-Za-z0-9_]*(<[^>(]*>)?", name)
    if not m:
        return name[:60]
    s = m.group(0)
    if name.startswith("_ZN"):
        cfg = re.search(r"ScanCfgILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)", name)
        s = re.sub(r"(kernel).*", r"\1", s)
        if cfg:
            s += "<%s,%s,%s,%s>" % cfg.groups()
        tv = re.search(r"(?:pp3_kernel|finalize_kernel|store_kernel\w*)ILi(\d+)E(?:Li(\d+)E)?", name)
        if tv and "pp3" in s:                                   # <ablation mask, VAR>: VAR 32 / 40 = the grouped launch (round 6), 16 / 24 the sample pass
            s += "<%s,%s>" % (tv.group(1), tv.group(2))
        if "store_kernel" in s and "Lb1E" in name:
            s += "<grouped>"
    return s


def main(src, dst):
    rows = [r for r in csv.DictReader(open(src)) if any(k in r["Name"] for k in OURS)]
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "StdDev"])
        for r in rows:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["MinNs"], r["MaxNs"],
                        r["StdDev"]])
    print(open(dst).read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
