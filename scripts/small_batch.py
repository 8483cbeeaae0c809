#!/usr/bin/env python
"""Latency of the small-batch dense path (skinny-GEMM stream, 1 / 2 / 4 column groups of 16 queries) at 1M x 1024:
B = 1 ... 64 and, for comparison, 65 ... 256 (the padded 256-query scan).  Library event timers per kernel class + wall time per call."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from easyrag_amd import synth  # noqa: E402
from easyrag_amd._lib import ERH_K_DENSE_SCAN, ERH_K_DENSE_SELECT  # noqa: E402
from easyrag_amd.engine import RetrievalEngine  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    n, d = 1_000_000, 1024
    eng = RetrievalEngine(0)
    x = synth.dense_corpus_torch(n, d, seed=2, device=dev)
    eng.set_dense(x)
    only_overlap = len(sys.argv) > 1 and sys.argv[1] == "overlap"      # just the side-stream sweep from 256 queries up
    for pipe in (() if only_overlap else (-1, 0, 1)):
        eng.set_option("dense_gemv_pipe", pipe)
        for B, k in ((1, 288), (1, 10), (4, 288), (16, 288), (17, 288), (32, 288), (48, 288), (64, 288), (65, 288), (128, 288), (256, 288)):
            if pipe >= 0 and (B > 64 or k == 10):
                continue
            q = synth.dense_queries_torch(x, B, seed=7)
            for _ in range(3):
                eng.dense_topk(q, k, device_out=True)
            torch.cuda.synchronize()
            eng.set_profiling(True)
            eng.reset_kernel_time()
            reps = 20
            t0 = time.perf_counter()
            for _ in range(reps):
                eng.dense_topk(q, k, device_out=True)
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / reps * 1e3
            eng.set_profiling(False)
            scan = eng.kernel_time(ERH_K_DENSE_SCAN)["ms"] / reps
            sel = eng.kernel_time(ERH_K_DENSE_SELECT)["ms"] / reps
            print(f"gemv_pipe={pipe:2d} B={B:3d} k={k:3d}: scan {scan:.3f} ms ({2.048 / scan:.2f} TB/s of the 2 GB matrix)  select {sel:.3f} ms  wall {wall:.3f} ms per call")
    eng.set_option("dense_gemv_pipe", -1)
    # sparse route and the fused call at B = 1 / 16: how the number of document-range segments (workgroups per query, then
    # one merge of their lists) trades scan time against merge time
    from easyrag_amd._lib import ERH_K_BM25_MERGE, ERH_K_BM25_SCAN, ERH_K_FUSE
    from easyrag_amd.engine import queries_to_csr
    from easyrag_amd.index import BM25S, build_bm25_index_from_postings
    vocab = 262_144
    indptr, doc, tf, lens, flat = synth.token_csr_torch(n, vocab, seed=3, device=dev)
    idx = build_bm25_index_from_postings(indptr, doc, tf, lens, BM25S, compute_payload=False)
    eng.set_bm25(idx, payload_on_device=True)
    eng.set_doc_meta(n, None, None)
    queries = synth.token_queries(flat, lens, vocab, 16, seed=9)
    for B in (() if only_overlap else (1, 16)):
        q = synth.dense_queries_torch(x, B, seed=7)
        qi, qt = queries_to_csr(queries[:B])
        for segs in (0, 5, 10, 21, 31):
            eng.set_option("bm25_segs", segs)
            for what, fn in (("bm25 top-192", lambda: eng.bm25_topk(qi, qt, 192, device_out=True)),
                             ("hybrid 288+192->10", lambda: eng.hybrid_topk(q, qi, qt, k_dense=288, k_sparse=192, K=60, topk=10, device_out=True))):
                for _ in range(3):
                    fn()
                torch.cuda.synchronize()
                eng.set_profiling(True)
                eng.reset_kernel_time()
                reps = 50
                t0 = time.perf_counter()
                for _ in range(reps):
                    fn()
                    torch.cuda.synchronize()
                wall = (time.perf_counter() - t0) / reps * 1e3
                eng.set_profiling(False)
                ks = {nm: eng.kernel_time(c)["ms"] / reps for nm, c in (("scan", ERH_K_BM25_SCAN), ("merge", ERH_K_BM25_MERGE), ("fuse", ERH_K_FUSE))}
                print(f"B={B:2d} bm25_segs={segs:2d} {what:20s}: bm25 scan {ks['scan']:.3f} merge {ks['merge']:.3f} fuse {ks['fuse']:.3f} ms  wall {wall:.3f} ms per call")
    eng.set_option("bm25_segs", 0)
    # the two routes are independent until the fusion: sparse route on a side stream (hybrid_overlap 1) or forked behind
    # the dense scan (2) -- at these batch sizes neither scan fills the chip
    for B in (1, 4, 16, 64, 128, 256, 384, 512, 768, 1024):
        if only_overlap and B < 256:
            continue
        q = synth.dense_queries_torch(x, B, seed=7)
        qi, qt = queries_to_csr(synth.token_queries(flat, lens, vocab, 1024, seed=9)[:B])
        for ov in (0, 1, 2):
            eng.set_option("hybrid_overlap", ov)
            fn = lambda: eng.hybrid_topk(q, qi, qt, k_dense=288, k_sparse=192, K=60, topk=10, device_out=True)
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            reps = 100
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
                torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / reps * 1e3
            print(f"B={B:2d} hybrid 288+192->10 hybrid_overlap={ov}: wall {wall:.3f} ms per call")
    eng.set_option("hybrid_overlap", -1)
    eng.close()


if __name__ == "__main__":
    main()
