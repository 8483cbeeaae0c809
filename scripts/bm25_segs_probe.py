#!/usr/bin/env python
"""Reference-length token queries (4 ... 45 tokens): what do the document-range segments per query (option bm25_segs) and the scan shape do to
the batch time, and how much of it is the tail of the few long queries?  (Round 6, DESIGN K2: 'per-query segment counts' costed before built.)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from easyrag_amd import synth  # noqa: E402
from easyrag_amd.engine import RetrievalEngine, queries_to_csr  # noqa: E402
from easyrag_amd.index import BM25S, build_bm25_index_from_postings  # noqa: E402


def timed(eng, csr, k, reps=20):
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
    short = [q for q, L in zip(qs, ql) if L <= 28]
    capped = [q if len(q) <= 28 else short[i % len(short)] for i, q in enumerate(qs)]
    print(f"{B} queries, lengths {ql.min()}..{ql.max()}, mean {ql.mean():.1f}; {int((ql > 28).sum())} longer than 28 tokens, {int((ql > 16).sum())} longer than 16")
    if len(sys.argv) > 1 and sys.argv[1] == "mixed":
        # one launch, two bodies (bm25_mixed) against the whole batch on the 32-bit shape, interleaved
        csr = queries_to_csr(qs)
        for rnd in range(3):
            for mixed in (1, 0):
                eng.set_option("bm25_mixed", mixed)
                ms, redo = timed(eng, csr, k)
                print(f"round {rnd}: reference lengths, bm25_mixed={mixed}: {ms:.3f} ms, {redo:.1f} redo, mixed launches {eng.stat('bm25_mixed_launches')}")
        eng.set_option("bm25_mixed", 1)
        for ls in (1, 2, 3, 4, 6, 8):
            eng.set_option("bm25_long_segs", ls)
            ms, redo = timed(eng, csr, k)
            print(f"bm25_long_segs={ls}: {ms:.3f} ms, {redo:.1f} redo")
        eng.set_option("bm25_long_segs", 4)
        for lt in (16, 20, 24, 28, 31):
            eng.set_option("bm25_mixed", 1)
            eng.set_option("bm25_long_tokens", lt)
            ms, redo = timed(eng, csr, k)
            print(f"bm25_long_tokens={lt}: {ms:.3f} ms, {redo:.1f} redo ({int((ql > lt).sum())} queries on the 32-bit body)")
        eng.set_option("bm25_long_tokens", 28)
        return
    for name, batch in (("reference lengths", qs), ("the same, queries > 28 tokens replaced by short ones", capped)):
        csr = queries_to_csr(batch)
        for shape, long_tokens in ((2, 28), (2, 0), (1, 0)):
            eng.set_option("bm25_small", shape)
            eng.set_option("bm25_long_tokens", long_tokens)
            for segs in (0, 2, 3, 4):
                eng.set_option("bm25_segs", segs)
                ms, redo = timed(eng, csr, k)
                print(f"{name}: bm25_small={shape} long_tokens={long_tokens} segs={segs or 'auto'}: {ms:.3f} ms, {redo:.1f} redo")
    eng.set_option("bm25_segs", 0)
    eng.set_option("bm25_small", 2)
    eng.set_option("bm25_long_tokens", 28)


if __name__ == "__main__":
    main()
