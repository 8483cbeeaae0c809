#!/usr/bin/env python
"""A/B of one library option on the PRODUCT library at the bench shapes -- the one protocol behind every "-x %" claim of
round 5: the arms are INTERLEAVED (a b a b ...), each visit times `--steps` calls with the library's
HIP-event classes and the wall clock, `--reps` visits per arm, and the report is median / min / max per arm and the
median-to-median delta.  Run-to-run drift on these boxes (clock / power state) is several per cent between consecutive
runs of the SAME configuration, so anything inside the spread printed here is not a result.  The arms' outputs are compared
bit for bit (an option must never change a result).

  python scripts/ab.py --workload bm25 --batch 1024 --k 192 --opt bm25_desc=0,1
  python scripts/ab.py --workload dense --batch 1024 --k 288 --opt dense_tile384=0,1 --reps 7
  python scripts/ab.py --workload hybrid --batch 1024 --opt bm25_desc=0,1 --opt2 dense_epi=0,1     (2 x 2 arms)
"""
import argparse
import json
import os
import statistics
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from easyrag_amd import synth  # noqa: E402
from easyrag_amd._lib import ERH_K_BM25_MERGE, ERH_K_BM25_SCAN, ERH_K_DENSE_SAMPLE, ERH_K_DENSE_SCAN, ERH_K_DENSE_SELECT, ERH_K_FUSE  # noqa: E402
from easyrag_amd.engine import RetrievalEngine, queries_to_csr  # noqa: E402
from easyrag_amd.index import BM25S, OKAPI, build_bm25_index_from_postings  # noqa: E402

CLASSES = (("dense_scan", ERH_K_DENSE_SCAN), ("dense_select", ERH_K_DENSE_SELECT), ("bm25_scan", ERH_K_BM25_SCAN),
           ("bm25_merge", ERH_K_BM25_MERGE), ("fuse", ERH_K_FUSE), ("dense_sample", ERH_K_DENSE_SAMPLE))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="hybrid", choices=["hybrid", "dense", "bm25"])
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--k", type=int, default=0, help="dense / bm25: top-k (default 288 / 192; hybrid always 288 + 192 -> 10)")
    ap.add_argument("--chunks", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=1024)
    ap.add_argument("--vocab", type=int, default=262_144)
    ap.add_argument("--variant", default="bm25s", choices=["bm25s", "okapi"])
    ap.add_argument("--opt", required=True, help="name=v0,v1[,v2...]")
    ap.add_argument("--opt2", default=None, help="second option, crossed with the first")
    ap.add_argument("--fixed", action="append", default=[], help="name=value set once for all arms")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--pool", type=int, default=4)
    ap.add_argument("--dirs", type=int, default=0, help="> 0: every query carries a `dir` filter (document i belongs to dir i %% D, query b asks for b %% D), "
                                                        "as every query of the reference's real workload does (src/data/question.jsonl)")
    ap.add_argument("--dir-layout", default="mod", choices=["mod", "block"], help="mod: document i in dir i %% D (interleaved); block: D contiguous blocks, "
                                                                                 "as the reference's loader produces them (it walks the directories one after the other)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    n, d, vocab, B = args.chunks, args.dim, args.vocab, args.batch
    eng = RetrievalEngine(0)
    for f in args.fixed:
        name, v = f.split("=")
        eng.set_option(name, int(v))
    q_pool, csr_pool = [], []
    if args.workload in ("hybrid", "dense"):
        x = synth.dense_corpus_torch(n, d, seed=2, device=dev)
        eng.set_dense(x)
        q_pool = [synth.dense_queries_torch(x, B, seed=1000 + p) for p in range(args.pool)]
    if args.workload in ("hybrid", "bm25"):
        indptr, doc, tf, lens, flat = synth.token_csr_torch(n, vocab, seed=3, device=dev)
        idx = build_bm25_index_from_postings(indptr, doc, tf, lens, BM25S if args.variant == "bm25s" else OKAPI, compute_payload=False)
        eng.set_bm25(idx, payload_on_device=True)
        csr_pool = [queries_to_csr(synth.token_queries(flat, lens, vocab, B, seed=2000 + p)) for p in range(args.pool)]
    filt = None
    if args.dirs > 0:
        eng.set_doc_meta(n, None, ((np.arange(n) % args.dirs) if args.dir_layout == "mod" else (np.arange(n) * args.dirs // n)).astype(np.int16))
        filt = (np.arange(B) % args.dirs).astype(np.int16)
    else:
        eng.set_doc_meta(n, None, None)
    k = args.k or (288 if args.workload == "dense" else 192)

    def call(p):
        if args.workload == "hybrid":
            return eng.hybrid_topk(q_pool[p], *csr_pool[p], k_dense=288, k_sparse=192, K=60, topk=10, device_out=True, filter_dir=filt)
        if args.workload == "dense":
            return eng.dense_topk(q_pool[p], k, device_out=True, filter_dir=filt)
        return eng.bm25_topk(*csr_pool[p], k, device_out=True, filter_dir=filt)

    def parse(spec):
        name, vals = spec.split("=")
        return name, [int(v) for v in vals.split(",")]

    n1, v1 = parse(args.opt)
    arms = [((n1, a),) for a in v1]
    if args.opt2:
        n2, v2 = parse(args.opt2)
        arms = [((n1, a), (n2, b)) for a in v1 for b in v2]
    label = lambda arm: " ".join(f"{nm}={v}" for nm, v in arm)                      # noqa: E731
    rec = {label(a): {"wall": [], **{c: [] for c, _ in CLASSES}} for a in arms}
    outs = {}
    for rep in range(args.reps + 1):                                               # visit 0 = warm-up of every arm (copies, caches), not recorded
        for arm in (arms if rep % 2 == 0 else arms[::-1]):                         # a b | b a | a b ...: no arm always runs behind the same neighbour
            for nm, v in arm:
                eng.set_option(nm, v)
            call(0)
            torch.cuda.synchronize()
            eng.set_profiling(True)
            eng.reset_kernel_time()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(args.steps):
                out = call(i % args.pool)
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / args.steps * 1e3
            eng.set_profiling(False)
            if args.workload != "bm25":
                eng.dense_check()
            if rep == 0:
                outs[label(arm)] = [t.cpu().numpy().copy() for t in call(0)]
                torch.cuda.synchronize()
                continue
            r = rec[label(arm)]
            r["wall"].append(wall)
            for c, cls in CLASSES:
                kt = eng.kernel_time(cls)
                if kt["launches"]:
                    r[c].append(kt["ms"] / args.steps)
    base = label(arms[0])
    print(f"# A/B {args.workload} B={B} k={k if args.workload != 'hybrid' else '288+192->10'} N={n} d={d} reps={args.reps} x steps={args.steps}, "
          f"interleaved; ms per step: median [min .. max]")
    for arm in arms:
        r = rec[label(arm)]
        parts = []
        for key in ("wall",) + tuple(c for c, _ in CLASSES):
            if r[key]:
                parts.append(f"{key} {statistics.median(r[key]):.4f} [{min(r[key]):.4f} .. {max(r[key]):.4f}]")
        print(f"{label(arm):32s} " + "  ".join(parts))
    for arm in arms[1:]:
        r, b = rec[label(arm)], rec[base]
        d_ = {key: round(100.0 * (statistics.median(r[key]) / statistics.median(b[key]) - 1.0), 2)
              for key in r if r[key] and b[key]}
        same = all(np.array_equal(a_.view(np.uint64) if a_.dtype == np.float64 else a_, b_.view(np.uint64) if b_.dtype == np.float64 else b_)
                   for a_, b_ in zip(outs[label(arm)], outs[base]))
        print(f"{label(arm)} vs {base}: delta of medians % {json.dumps(d_)}  results identical: {same}")
    print("stats " + json.dumps(eng.stats()))
    eng.close()


if __name__ == "__main__":
    main()
