#!/usr/bin/env python
"""Section clocks of the fixed-point BM25 scan (thread 0 of every workgroup, summed) with and without a dir filter: where a filtered
query's time goes when it walks only its dir's posting tiles (needs the measurement build: ERH_MEASURE=1).  NOTE: the sweep path of a
tile (no threshold yet, or too many crossings) has no marks of its own -- its time is booked under the NEXT mark, "describe" (or "final
shrink" behind the last tile); profiles/r05y_bm25_filtered_clocks.log splits it out with temporary marks.
  ERH_MEASURE=1 python scripts/bm25_filter_clocks.py [dirs ...]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from easyrag_amd import synth  # noqa: E402
from easyrag_amd.engine import RetrievalEngine, queries_to_csr  # noqa: E402
from easyrag_amd.index import BM25S, build_bm25_index_from_postings  # noqa: E402

NAMES = ["describe", "publish+fill (+extra rounds)", "apply_barrier", "list+clear | sweep", "clear_barrier", "list shrinks", "load wait", "adds",
         "final shrink", "re-score+rank+output"]
TAIL = ["token table", "search set-up", "searches", "payloads", "sums", "rank+output"]


def main():
    dirs_list = [int(a) for a in sys.argv[1:]] or [0, 4, 32]
    dev = torch.device("cuda", 0)
    n, vocab, B, k = 1_000_000, 262_144, 1024, 192
    eng = RetrievalEngine(0)
    indptr, doc, tf, lens, flat = synth.token_csr_torch(n, vocab, seed=3, device=dev)
    eng.set_bm25(build_bm25_index_from_postings(indptr, doc, tf, lens, BM25S, compute_payload=False), payload_on_device=True)
    qi, qt = queries_to_csr(synth.token_queries(flat, lens, vocab, B, seed=9))
    for dirs in dirs_list:
        filt = None
        if dirs:
            eng.set_doc_meta(n, None, (np.arange(n) * dirs // n).astype(np.int16))
            filt = (np.arange(B) % dirs).astype(np.int16)
        else:
            eng.set_doc_meta(n, None, None)
        for _ in range(3):
            eng.bm25_topk(qi, qt, k, device_out=True, filter_dir=filt)
        torch.cuda.synchronize()
        eng.set_profiling(True)
        eng.reset_kernel_time()
        for _ in range(20):
            eng.bm25_topk(qi, qt, k, device_out=True, filter_dir=filt)
        torch.cuda.synchronize()
        eng.set_profiling(False)
        from easyrag_amd._lib import ERH_K_BM25_SCAN
        ms = eng.kernel_time(ERH_K_BM25_SCAN)["ms"] / 20
        eng.set_option("debug_counters", 1)
        eng.bm25_topk(qi, qt, k, device_out=True, filter_dir=filt)
        torch.cuda.synchronize()
        c = eng.debug_counters().astype(np.float64)
        eng.set_option("debug_counters", 0)
        tot = c[:10].sum()
        print(f"dirs={dirs}: scan class {ms:.4f} ms per 1024 queries; thread-0 cycles per query {tot / B:.0f}")
        print("   sections (cycles per query): " + json.dumps({n_: round(v / B) for n_, v in zip(NAMES, c[:10])}))
        print("   tail (cycles per query):     " + json.dumps({n_: round(v / B) for n_, v in zip(TAIL, c[10:16])}))
    eng.close()


if __name__ == "__main__":
    main()
