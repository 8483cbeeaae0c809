#!/usr/bin/env python
"""One query per call: how much of a call is HOST time.  Times (a) the enqueue loop without any synchronisation (host cost per call, the
device running behind), (b) the same with erh_dense_check per call (what a caller sees), for the dense and the fused call, with and
without a dir filter (four contiguous dirs), and the host cost of the small pageable / pinned host-to-device copies a call is made of.
  python scripts/b1_host_cost.py [calls]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from easyrag_amd import synth  # noqa: E402
from easyrag_amd.engine import RetrievalEngine, queries_to_csr  # noqa: E402
from easyrag_amd.index import BM25S, build_bm25_index_from_postings  # noqa: E402


def main():
    calls = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    dev = torch.device("cuda", 0)
    n, d, vocab = 1_000_000, 1024, 262_144
    eng = RetrievalEngine(0)
    x = synth.dense_corpus_torch(n, d, seed=2, device=dev)
    eng.set_dense(x)
    q = [synth.dense_queries_torch(x, 1, seed=7 + i) for i in range(4)]
    qh = [t.cpu().numpy() for t in q]
    indptr, doc, tf, lens, flat = synth.token_csr_torch(n, vocab, seed=3, device=dev)
    eng.set_bm25(build_bm25_index_from_postings(indptr, doc, tf, lens, BM25S, compute_payload=False), payload_on_device=True)
    csr = [queries_to_csr(synth.token_queries(flat, lens, vocab, 1, seed=9 + i)) for i in range(4)]
    eng.set_doc_meta(n, None, (np.arange(n) * 4 // n).astype(np.int16))
    filt = [np.array([i % 4], np.int16) for i in range(4)]

    def run(name, fn, check):
        for i in range(8):
            fn(i)
        eng.dense_check()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(calls):
            fn(i)
            if check:
                eng.dense_check()
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        eng.dense_check()
        print(f"{name:44s} {'check per call' if check else 'no sync       '}: host loop {t_host / calls * 1e3:.4f} ms per call, "
              f"until the device is idle {t_all / calls * 1e3:.4f}")

    for check in (False, True):
        run("dense top-288, device query", lambda i: eng.dense_topk(q[i % 4], 288, device_out=True), check)
        run("dense top-288, host query", lambda i: eng.dense_topk(qh[i % 4], 288, device_out=True), check)
        run("dense top-288, device query, dir filter", lambda i: eng.dense_topk(q[i % 4], 288, device_out=True, filter_dir=filt[i % 4]), check)
        run("fused, device query", lambda i: eng.hybrid_topk(q[i % 4], *csr[i % 4], k_dense=288, k_sparse=192, K=60, topk=10, device_out=True), check)
        run("fused, device query, dir filter", lambda i: eng.hybrid_topk(q[i % 4], *csr[i % 4], k_dense=288, k_sparse=192, K=60, topk=10,
                                                                         device_out=True, filter_dir=filt[i % 4]), check)
        run("bm25 top-192", lambda i: eng.bm25_topk(*csr[i % 4], 192, device_out=True), check)
    # the small copies
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
    dbuf = torch.empty(4096, dtype=torch.uint8, device=dev)
    page = np.zeros(4096, np.uint8)
    pin = C.c_void_p()
    assert hip.hipHostMalloc(C.byref(pin), 4096, 0) == 0
    for name, src in (("pageable", page.ctypes.data), ("pinned", pin.value)):
        for nbytes in (8, 2048):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(2000):
                hip.hipMemcpyAsync(dbuf.data_ptr(), src, nbytes, 1, None)
            t_host = time.perf_counter() - t0
            torch.cuda.synchronize()
            t_all = time.perf_counter() - t0
            print(f"hipMemcpyAsync H2D {nbytes:5d} B from {name:8s} memory: host {t_host / 2000 * 1e6:.2f} us per copy, until idle {t_all / 2000 * 1e6:.2f}")
    eng.close()


if __name__ == "__main__":
    main()
