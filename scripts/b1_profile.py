#!/usr/bin/env python
"""One query per call (the reference's call pattern) under rocprofv3 --kernel-trace --stats: which of the ~20 launches of a dense / fused
call take the time that is not the 2 GB stream.  python scripts/b1_profile.py [dense|hybrid|filtered] [calls]
(filtered: the fused call with its dir filter, four contiguous dirs -- the reference's REAL call)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from easyrag_amd import synth  # noqa: E402
from easyrag_amd.engine import RetrievalEngine, queries_to_csr  # noqa: E402
from easyrag_amd.index import BM25S, build_bm25_index_from_postings  # noqa: E402


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "dense"
    calls = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    dev = torch.device("cuda", 0)
    n, d, vocab = 1_000_000, 1024, 262_144
    eng = RetrievalEngine(0)
    x = synth.dense_corpus_torch(n, d, seed=2, device=dev)
    eng.set_dense(x)
    q = [synth.dense_queries_torch(x, 1, seed=7 + i) for i in range(4)]
    csr = None
    if what in ("hybrid", "filtered"):
        indptr, doc, tf, lens, flat = synth.token_csr_torch(n, vocab, seed=3, device=dev)
        idx = build_bm25_index_from_postings(indptr, doc, tf, lens, BM25S, compute_payload=False)
        eng.set_bm25(idx, payload_on_device=True)
        csr = [queries_to_csr(synth.token_queries(flat, lens, vocab, 1, seed=9 + i)) for i in range(4)]
    import numpy as np
    eng.set_doc_meta(n, None, (np.arange(n) * 4 // n).astype(np.int16) if what == "filtered" else None)
    filt = [np.array([i % 4], np.int16) for i in range(4)]
    for i in range(calls):
        if what == "filtered":
            eng.hybrid_topk(q[i % 4], *csr[i % 4], k_dense=288, k_sparse=192, K=60, topk=10, device_out=True, filter_dir=filt[i % 4])
        elif what == "hybrid":
            eng.hybrid_topk(q[i % 4], *csr[i % 4], k_dense=288, k_sparse=192, K=60, topk=10, device_out=True)
        else:
            eng.dense_topk(q[i % 4], 288, device_out=True)
        torch.cuda.synchronize()
    eng.close()


if __name__ == "__main__":
    main()
