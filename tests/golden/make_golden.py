#!/usr/bin/env python
"""Generate tests/golden/golden_small.npz from the oracle (the reference ships no golden vectors and its
libraries cannot be imported here -- SURVEY.md 8(c) -- so these pin the oracle against regressions and give
the GPU tests a fixture that does not depend on re-running the oracle).

    python tests/golden/make_golden.py        # rewrites golden_small.npz deterministically
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import (BM25Okapi, BM25SLucene, bm25_filter, dense_exact_topk, reciprocal_rank_fusion,  # noqa: E402
                    fusion, to_f16_unit)
from oracle.retrievers import Item  # noqa: E402

N, V, D, B = 600, 150, 128, 12
K_DENSE, K_SPARSE, TOPK = 288, 192, 10


def build():
    rng = np.random.default_rng(20240924)
    lens = rng.integers(4, 30, size=N)
    p = 1.0 / np.arange(1, V + 1) ** 1.07
    p /= p.sum()
    flat = rng.choice(V, size=int(lens.sum()), p=p).astype(np.int32)
    off = np.concatenate([[0], np.cumsum(lens)])
    docs = [list(map(int, flat[off[i]:off[i + 1]])) for i in range(N)]
    docs[77] = list(docs[5])                       # exact duplicate documents -> exact score ties
    docs[300] = list(docs[5])
    flat = np.asarray([t for d in docs for t in d], np.int32)
    lens = np.asarray([len(d) for d in docs], np.int32)
    q_lens = rng.integers(1, 9, size=B)
    queries = [list(map(int, rng.choice(V + 4, size=n))) for n in q_lens]   # a few out-of-vocabulary ids (>= V)
    queries[3] = queries[3] + queries[3]           # repeats
    x16 = to_f16_unit(rng.standard_normal((N, D)))
    x16[300] = x16[5]
    q16 = to_f16_unit(x16[rng.integers(0, N, size=B)].astype(np.float32) + 0.08 * rng.standard_normal((B, D)))
    cid = np.arange(N, dtype=np.int32)
    cid[77] = 5
    cid[300] = 5
    dir_id = (np.arange(N) % 4).astype(np.int16)

    out = dict(flat=flat, lens=lens, x16=x16, q16=q16, content_id=cid, dir_id=dir_id,
               q_flat=np.asarray([t for q in queries for t in q], np.int32),
               q_lens=np.asarray([len(q) for q in queries], np.int32))
    okapi = BM25Okapi(docs, 1.5, 0.75, 0.25)
    bm25s = BM25SLucene(1.5, 0.75).index(docs)
    out["okapi_scores"] = np.stack([okapi.get_scores(q) for q in queries])
    out["bm25s_scores"] = np.stack([bm25s.get_scores(q) for q in queries])

    def pad(rows, k, fill, dtype):
        a = np.full((len(rows), k), fill, dtype)
        for i, r in enumerate(rows):
            a[i, :len(r)] = r
        return a

    for name, sc in (("okapi", out["okapi_scores"]), ("bm25s", out["bm25s_scores"])):
        tops = [bm25_filter(s, K_SPARSE) for s in sc]
        out[f"{name}_top_ids"] = pad([[i for i, _ in t] for t in tops], K_SPARSE, -1, np.int32)
        out[f"{name}_top_sc"] = pad([[s for _, s in t] for t in tops], K_SPARSE, 0.0, np.float64)
        out[f"{name}_top_len"] = np.asarray([len(t) for t in tops], np.int32)
        ftops = [bm25_filter(s, 20, dir_id == 2) for s in sc]
        out[f"{name}_filt_ids"] = pad([[i for i, _ in t] for t in ftops], 20, -1, np.int32)
    dense = [dense_exact_topk(x16, q16[b], K_DENSE) for b in range(B)]
    out["dense_ids"] = np.stack([d[0] for d in dense]).astype(np.int32)
    out["dense_sc"] = np.stack([d[1] for d in dense])
    rrf_ids, rrf_sc, fus_ids = [], [], []
    for b in range(B):
        A = [Item(int(i), int(cid[i]), float(s)) for i, s in bm25_filter(out["okapi_scores"][b], K_SPARSE)]
        Bl = [Item(int(i), int(cid[i]), float(s)) for i, s in zip(*dense[b])]
        r = reciprocal_rank_fusion([A, Bl], K=60, topk=TOPK)
        rrf_ids.append([w.idx for w in r])
        rrf_sc.append([w.score for w in r])
        fus_ids.append([w.idx for w in fusion([A, Bl], topk=TOPK)])
    out["rrf_ids"] = pad(rrf_ids, TOPK, -1, np.int32)
    out["rrf_sc"] = pad(rrf_sc, TOPK, 0.0, np.float64)
    out["fusion_ids"] = pad(fus_ids, TOPK, -1, np.int32)
    return out


if __name__ == "__main__":
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_small.npz")
    np.savez_compressed(path, **build())
    print("wrote", path, os.path.getsize(path), "bytes")
