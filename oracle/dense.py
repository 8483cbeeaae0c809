"""Oracle: the dense route (QdrantRetriever -> Qdrant COSINE search).  TEST INFRASTRUCTURE ONLY.

Reference call sites: src/easyrag/custom/retrievers.py:37-52 (embed query, VectorStoreQuery,
``vector_store.aquery(..., qdrant_filters=self.filters)``, zip nodes/similarities) over a
collection created with ``Distance.COSINE`` (src/easyrag/pipeline/ingestion.py:178-183).
The arithmetic lives in qdrant-client==1.8.2 (requirements.txt:75), not vendored; restated
from its local mode:

  qdrant_cosine_search   stored vectors L2-normalised at insert (float32), query normalised,
                         scores = np.dot(vectors, query), order = np.argsort(scores)[::-1],
                         walk skipping points that fail the payload filter, stop at `limit`.

The build stores the chunk matrix as fp16 and hands the kernel an fp16 query.  Because two
fp32 accumulation orders (MFMA vs BLAS) can swap near-equal scores, the *ranking* contract
is defined on an order-fixed fp64 evaluation of the very same fp16 values:

  dense_exact_scores     s[i] = sum_k X16[i,k] * q16[k] in float64 with the summation order
                         below (products of two fp16 values are exact in float64, so the only
                         rounding is in the adds, and their order is pinned):
                           lane j in 0..63, round t: elements 512*t + 8*j + e, e = 0..7
                           acc_j += x*q   sequentially over (t, e)
                           then tree: for off in 32,16,8,4,2,1: acc[:off] += acc[off:2*off]
  dense_exact_topk       rank by (score desc, index asc), optional boolean mask, first k.

The GPU re-scores its fp32 MFMA candidates with exactly this order
(easyrag_amd/csrc/select.hip: dense_finalize_kernel), so ids and fp64 scores are compared
bit-for-bit; the fp32 MFMA scores themselves are compared to both oracles within 1e-3.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np

EPS = 1e-12


def to_f16_unit(v: np.ndarray) -> np.ndarray:
    """L2-normalise rows in float64 and round once to float16 (what the caller stores / sends)."""
    v = np.asarray(v, dtype=np.float32)
    v64 = v.astype(np.float64)
    n = np.sqrt(np.sum(v64 * v64, axis=-1, keepdims=True))
    n = np.where(n != 0.0, n, EPS)
    return (v64 / n).astype(np.float16)


def qdrant_cosine_search(vectors: np.ndarray, query: np.ndarray, limit: int,
                         mask: Optional[np.ndarray] = None, prenormalized: bool = False
                         ) -> Tuple[np.ndarray, np.ndarray]:
    """qdrant-client local-mode COSINE search restated.  `vectors` is whatever was upserted
    (any float dtype; stored as float32), `query` the embedding list.  Returns (ids, scores)
    in the literal ``np.argsort(scores)[::-1]`` order (ties implementation-defined).
    prenormalized=True = the insert-time normalisation already happened (the per-query work is
    then only the query normalisation, the GEMV, the argsort and the walk -- what a timed
    baseline should count)."""
    vec = np.asarray(vectors, dtype=np.float32)
    if not prenormalized:
        norm = np.linalg.norm(vec, axis=-1)[:, np.newaxis]
        vec = vec / np.where(norm != 0.0, norm, EPS).astype(np.float32)
    q = np.asarray(query, dtype=np.float32)
    qn = np.linalg.norm(q)
    q = q / np.float32(qn if qn != 0.0 else EPS)
    scores = np.dot(vec, q)
    order = np.argsort(scores)[::-1]
    ids, out = [], []
    for idx in order:
        if len(ids) >= limit:
            break
        if mask is not None and not mask[idx]:
            continue
        ids.append(int(idx))
        out.append(float(scores[idx]))
    return np.asarray(ids, dtype=np.int64), np.asarray(out, dtype=np.float64)


def dense_exact_scores(x16: np.ndarray, q16: np.ndarray, rows: Optional[np.ndarray] = None) -> np.ndarray:
    """Order-pinned float64 inner products of fp16 rows with one fp16 query (see module doc)."""
    x16 = np.asarray(x16)
    q16 = np.asarray(q16)
    assert x16.dtype == np.float16 and q16.dtype == np.float16
    if rows is not None:
        x16 = x16[rows]
    n, d = x16.shape
    T = (d + 511) // 512
    dp = T * 512
    out = np.empty(n, np.float64)
    qp = np.zeros(dp, np.float64)
    qp[:d] = q16.astype(np.float64)
    q3 = qp.reshape(T, 64, 8)
    CH = 8192
    for s in range(0, n, CH):
        xb = x16[s:s + CH].astype(np.float64)
        m = xb.shape[0]
        xp = np.zeros((m, dp), np.float64)
        xp[:, :d] = xb
        p = xp.reshape(m, T, 64, 8) * q3[None]          # exact products
        acc = np.zeros((m, 64), np.float64)
        for t in range(T):
            for e in range(8):
                acc = acc + p[:, t, :, e]                # sequential per lane
        off = 32
        while off >= 1:
            acc = acc[:, :off] + acc[:, off:2 * off]     # xor-butterfly == fold halves
            off //= 2
        out[s:s + m] = acc[:, 0]
    return out


def dense_exact_topk(x16: np.ndarray, q16: np.ndarray, k: int,
                     mask: Optional[np.ndarray] = None) -> Tuple[np.ndarray, np.ndarray]:
    """Canonical dense ranking: (fp64 score desc, index asc), mask applied, first k."""
    s = dense_exact_scores(x16, q16)
    idx = np.arange(s.shape[0])
    if mask is not None:
        keep = np.asarray(mask, dtype=bool)
        idx = idx[keep]
        s = s[keep]
    order = np.lexsort((idx, -s))[:k]
    return idx[order].astype(np.int64), s[order]
