#!/usr/bin/env python
"""Where do long token queries leave the fixed-point scan?  (Round 6: the reference's real questions have 4 ... 45 tokens; the bench's ten-token
queries never showed what a 30-token query costs.)  For each shape of the scan (bm25_small 2 packed 16-bit sums / 1 32-bit sums in 16384-document
tiles / 0 1024 threads) and for buckets of query length: ms per batch and (query, segment) pairs handed to the exact block scan."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from easyrag_amd import synth  # noqa: E402
from easyrag_amd.engine import RetrievalEngine, queries_to_csr  # noqa: E402
from easyrag_amd.index import BM25S, build_bm25_index_from_postings  # noqa: E402


def timed(eng, csr, k, reps=10):
    eng.bm25_topk(*csr, k, device_out=True)
    torch.cuda.synchronize()
    eng.reset_stats()
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.bm25_topk(*csr, k, device_out=True)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, eng.stat("bm25_redo_segments") / reps


def main():
    n, vocab, B, k = 1_000_000, 262_144, 1024, 192
    dev = torch.device("cuda", 0)
    indptr, doc, tf, lens, flat = synth.token_csr_torch(n, vocab, seed=3, device=dev)
    idx = build_bm25_index_from_postings(indptr, doc, tf, lens, BM25S, compute_payload=False)
    eng = RetrievalEngine(0)
    eng.set_bm25(idx, payload_on_device=True)
    qs = synth.token_queries(flat, lens, vocab, B, seed=4000, lengths=synth.REF_QUESTION_LENGTHS)
    ql = np.array([len(q) for q in qs])
    for name, v in (("split", int(sys.argv[1]) if len(sys.argv) > 1 else -1),):
        if v >= 0:
            eng.set_option("bm25_long_tokens", v)
    for shape in (2, 1, 0):
        eng.set_option("bm25_small", shape)
        ms, redo = timed(eng, queries_to_csr(qs), k)
        print(f"bm25_small={shape}: all {B} reference-length queries {ms:.3f} ms, {redo:.1f} redo segments per batch")
        for lo, hi in ((1, 8), (9, 12), (13, 16), (17, 22), (23, 31), (32, 64)):
            sel = [q for q, L in zip(qs, ql) if lo <= L <= hi]
            if not sel:
                continue
            rep = (sel * (B // len(sel) + 1))[:B]                # the bucket's queries repeated to a full batch
            ms, redo = timed(eng, queries_to_csr(rep), k)
            print(f"   lengths {lo:2d}..{hi:2d} ({len(sel):4d} distinct, batch of {B}): {ms:.3f} ms, {redo:.1f} redo segments per batch")
    fixed = synth.token_queries(flat, lens, vocab, B, seed=2000)
    eng.set_option("bm25_small", 2)
    print("ten-token queries, packed shape: %.3f ms, %.1f redo" % timed(eng, queries_to_csr(fixed), k))


if __name__ == "__main__":
    main()
