#!/usr/bin/env python
"""Probe: do the dense scan (power-limited MFMA work) and the BM25 scan (latency-bound LDS work) of a hybrid step gain from
running side by side on disjoint CU sets (hipExtStreamCreateWithCUMask) instead of one after the other on the whole chip?
Prints HIP-event class times of each route alone on its CU set and the wall time of both together.  Measurement only."""
import ctypes
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from easyrag_amd import synth  # noqa: E402
from easyrag_amd._lib import ERH_K_BM25_SCAN, ERH_K_DENSE_SCAN, ERH_K_DENSE_SELECT  # noqa: E402
from easyrag_amd.engine import RetrievalEngine, queries_to_csr  # noqa: E402
from easyrag_amd.index import BM25S, build_bm25_index_from_postings  # noqa: E402

hip = ctypes.CDLL("libamdhip64.so")


def masked_stream(words):
    st = ctypes.c_void_p()
    arr = (ctypes.c_uint32 * len(words))(*words)
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), ctypes.c_uint32(len(words)), arr)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask -> {rc}")
    return st.value


def class_ms(eng, fn, reps=10):
    fn()
    torch.cuda.synchronize()
    eng.set_profiling(True)
    eng.reset_kernel_time()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    eng.set_profiling(False)
    out = {}
    for name, cls in (("dense_scan", ERH_K_DENSE_SCAN), ("dense_select", ERH_K_DENSE_SELECT), ("bm25_scan", ERH_K_BM25_SCAN)):
        kt = eng.kernel_time(cls)
        if kt["launches"]:
            out[name] = round(kt["ms"] / reps, 4)
    return out


def wall_ms(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return round((time.perf_counter() - t0) * 1e3 / reps, 4)


def main():
    dev = torch.device("cuda", 0)
    n, d, vocab, B = 1_000_000, 1024, 262_144, 1024
    eng_d = RetrievalEngine(0)                       # one handle per stream (handles are thread-compatible, not re-entrant)
    eng_s = RetrievalEngine(0)
    x = synth.dense_corpus_torch(n, d, seed=2, device=dev)
    eng_d.set_dense(x)
    q = synth.dense_queries_torch(x, B, seed=7)
    indptr, doc, tf, lens, flat = synth.token_csr_torch(n, vocab, seed=3, device=dev)
    idx = build_bm25_index_from_postings(indptr, doc, tf, lens, BM25S, compute_payload=False)
    eng_s.set_bm25(idx, payload_on_device=True)
    qi, qt = queries_to_csr(synth.token_queries(flat, lens, vocab, B, seed=9))
    full = 0xFFFFFFFF
    print("whole chip, default stream:", json.dumps({**class_ms(eng_d, lambda: eng_d.dense_topk(q, 288, device_out=True)),
                                                     **class_ms(eng_s, lambda: eng_s.bm25_topk(qi, qt, 192, device_out=True))}), flush=True)
    splits = {
        "dense = CUs 0..191, sparse = CUs 192..255": ([full] * 6 + [0, 0], [0] * 6 + [full, full], 192),
        "dense = three of every four CUs, sparse = the fourth": ([0x77777777] * 8, [0x88888888] * 8, 192),
        "dense = CUs 0..223, sparse = CUs 224..255": ([full] * 7 + [0], [0] * 7 + [full], 224),
        "dense = seven of every eight CUs, sparse = the eighth": ([0x7F7F7F7F] * 8, [0x80808080] * 8, 224),
    }
    for label, (md, ms, ncd) in splits.items():
        sd, ss = masked_stream(md), masked_stream(ms)
        eng_d.set_option("n_cus", ncd)
        r = {"dense alone": class_ms(eng_d, lambda: eng_d.dense_topk(q, 288, device_out=True, stream=sd)),
             "sparse alone": class_ms(eng_s, lambda: eng_s.bm25_topk(qi, qt, 192, device_out=True, stream=ss))}
        r["dense alone wall"] = wall_ms(lambda: eng_d.dense_topk(q, 288, device_out=True, stream=sd))
        r["sparse alone wall"] = wall_ms(lambda: eng_s.bm25_topk(qi, qt, 192, device_out=True, stream=ss))

        def both():
            eng_d.dense_topk(q, 288, device_out=True, stream=sd)
            eng_s.bm25_topk(qi, qt, 192, device_out=True, stream=ss)
        r["both wall"] = wall_ms(both)
        print(label + ":", json.dumps(r), flush=True)
        eng_d.set_option("n_cus", 0)
    eng_d.set_option("n_cus", 0)
    r = {"dense wall": wall_ms(lambda: eng_d.dense_topk(q, 288, device_out=True)),
         "sparse wall": wall_ms(lambda: eng_s.bm25_topk(qi, qt, 192, device_out=True))}

    def seq():
        eng_d.dense_topk(q, 288, device_out=True)
        eng_s.bm25_topk(qi, qt, 192, device_out=True)
    r["one after the other wall"] = wall_ms(seq)
    print("whole chip, default stream:", json.dumps(r), flush=True)
    eng_d.close()
    eng_s.close()


if __name__ == "__main__":
    main()
