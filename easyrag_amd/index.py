"""Host-side BM25 index build: tokenised corpus -> CSR inverted postings with per-posting payloads.

This is the build-side counterpart of ``BM25Retriever.__init__``
(/root/reference/src/easyrag/custom/retrievers.py:94-118), which hands the tokenised corpus to
``rank_bm25.BM25Okapi(corpus, k1, b, epsilon)`` (bm25_type 0) or ``bm25s.BM25(k1, b).index(corpus)``
(bm25_type 1).  The output layout is what libeasyrag_hip.so scans (include/easyrag_hip.h:
erh_set_bm25_csr / erh_set_bm25_tf):

  indptr  int64[V+1]   postings of term t are [indptr[t], indptr[t+1])
  doc_ids int32[nnz]   strictly ascending inside a term
  tf      int32[nnz]   term frequency
  payload float64[nnz] (Okapi)  idf[t] * (tf*(k1+1) / (tf + k1*((1-b) + b*dl/avgdl)))
          float32[nnz] (bm25s)  idf[t] * (tf / (f32(k1*((1-b) + b*dl/avgdl)) + tf))

The floating-point operation order follows the two libraries so that a
per-document sum of payloads in query-token order reproduces their score vectors bit for bit.
Everything is vectorised numpy over the posting arrays; the GPU can also evaluate the payload
itself from (tf, doc_len, idf) -- see RetrievalEngine.set_bm25(..., payload_on_device=True).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, Hashable, List, Optional, Sequence

import numpy as np

OKAPI = 0   # == bm25_type 0 in the reference config (src/configs/easyrag.yaml:21)
BM25S = 1   # == bm25_type 1


@dataclass
class BM25Index:
    variant: int
    n_docs: int
    n_vocab: int
    indptr: np.ndarray
    doc_ids: np.ndarray
    tf: np.ndarray
    doc_len: np.ndarray
    idf: np.ndarray            # float64[V] (Okapi) / float32[V] (bm25s)
    avgdl: float
    k1: float
    b: float
    epsilon: float
    payload: np.ndarray        # float64[nnz] (Okapi) / float32[nnz] (bm25s)
    vocab: Dict[Hashable, int] = field(default_factory=dict)
    average_idf: float = 0.0
    n_postings: Optional[int] = None     # set when the arrays live only on the device (RetrievalEngine.build_bm25(fetch=False))

    @property
    def nnz(self) -> int:
        return int(self.doc_ids.shape[0]) if self.n_postings is None else int(self.n_postings)

    def tokens_to_ids(self, tokens: Sequence[Hashable]) -> np.ndarray:
        """Query tokens -> term ids, order and repeats kept, out-of-vocabulary tokens dropped (they add
        nothing in either library: ``idf.get(q) or 0`` / ``if token in vocab``)."""
        v = self.vocab
        if hasattr(v, "ids_of"):                              # NativeVocab: one library call
            return v.ids_of(tokens)
        return np.fromiter((v[t] for t in tokens if t in v), dtype=np.int32)


def _flatten(corpus_ids: Sequence[Sequence[int]]):
    lens = np.fromiter((len(d) for d in corpus_ids), dtype=np.int64, count=len(corpus_ids))
    total = int(lens.sum())
    flat = np.empty(total, np.int64)
    pos = 0
    for d in corpus_ids:
        n = len(d)
        flat[pos:pos + n] = d
        pos += n
    return flat, lens


def _finish(variant: int, n_docs: int, n_vocab: int, indptr: np.ndarray, doc: np.ndarray, tf: np.ndarray,
            doc_lens: np.ndarray, first_seen_order: Optional[np.ndarray], k1: float, b: float, epsilon: float,
            compute_payload: bool) -> BM25Index:
    """idf (+ epsilon floor) and per-posting payload from finished postings."""
    df = np.diff(indptr)
    doc_len32 = doc_lens.astype(np.int32)
    present = np.nonzero(df)[0]
    term = np.repeat(np.arange(n_vocab, dtype=np.int64), df) if compute_payload else None

    if variant == OKAPI:
        # rank_bm25: avgdl = num_doc / corpus_size; idf = log(N - n + 0.5) - log(n + 0.5) with math.log
        avgdl = int(doc_lens.sum()) / n_docs
        idf = np.zeros(n_vocab, np.float64)
        lut = {int(n): math.log(n_docs - int(n) + 0.5) - math.log(int(n) + 0.5) for n in np.unique(df[present])}
        idf[present] = np.fromiter((lut[int(n)] for n in df[present]), dtype=np.float64, count=present.size)
        # average_idf: sequential sum in the order rank_bm25's `nd` dict was filled = first appearance of each
        # term in the token stream (ascending term id when the caller has no stream)
        order = present if first_seen_order is None else first_seen_order
        idf_sum = 0
        for v in idf[order].tolist():
            idf_sum += v
        average_idf = idf_sum / max(len(order), 1)
        eps = epsilon * average_idf
        idf[idf < 0] = eps
        payload = np.zeros(0, np.float64)
        if compute_payload:
            q_freq = tf.astype(np.int64)
            dl = doc_lens[doc]
            payload = idf[term] * (q_freq * (k1 + 1) / (q_freq + k1 * (1 - b + b * dl / avgdl)))
            payload = np.ascontiguousarray(payload, dtype=np.float64)
        return BM25Index(OKAPI, n_docs, n_vocab, indptr, doc, tf, doc_len32, idf, float(avgdl), k1, b, epsilon,
                         payload, {}, float(average_idf))

    if variant == BM25S:
        # bm25s lucene: l_avg = mean(len(doc)); idf = log(1 + (N - df + 0.5)/(df + 0.5)) stored float32
        l_avg = np.array(doc_lens).mean()
        idf = np.zeros(n_vocab, np.float32)
        lut = {int(n): math.log(1 + (n_docs - int(n) + 0.5) / (int(n) + 0.5)) for n in np.unique(df[present])}
        idf[present] = np.fromiter((lut[int(n)] for n in df[present]), dtype=np.float64,
                                   count=present.size).astype(np.float32)
        payload = np.zeros(0, np.float32)
        if compute_payload:
            bracket = (k1 * ((1 - b) + b * doc_lens / l_avg)).astype(np.float32)   # float64 scalar per doc -> f32 once
            tf32 = tf.astype(np.float32)
            tfc = tf32 / (bracket[doc] + tf32)                                      # float32 add, float32 divide
            payload = np.ascontiguousarray(idf[term] * tfc, dtype=np.float32)       # float32 multiply
        return BM25Index(BM25S, n_docs, n_vocab, indptr, doc, tf, doc_len32, idf, float(l_avg), k1, b, epsilon,
                         payload, {}, 0.0)

    raise ValueError("variant must be 0 (Okapi) or 1 (bm25s)")


def build_bm25_index_from_postings(indptr: np.ndarray, doc_ids: np.ndarray, tf: np.ndarray, doc_lens: np.ndarray,
                                   variant: int = OKAPI, k1: float = 1.5, b: float = 0.75, epsilon: float = 0.25,
                                   first_seen_order: Optional[np.ndarray] = None,
                                   compute_payload: bool = True) -> BM25Index:
    """Finish an index whose CSR postings (term-major, docs ascending inside a term) already exist."""
    indptr = np.ascontiguousarray(indptr, dtype=np.int64)
    doc_ids = np.ascontiguousarray(doc_ids, dtype=np.int32)
    tf = np.ascontiguousarray(tf, dtype=np.int32)
    doc_lens = np.asarray(doc_lens, dtype=np.int64)
    return _finish(variant, int(doc_lens.shape[0]), int(indptr.shape[0] - 1), indptr, doc_ids, tf, doc_lens,
                   first_seen_order, k1, b, epsilon, compute_payload)


def build_bm25_index_from_ids(corpus_ids: Optional[Sequence[Sequence[int]]] = None, n_vocab: Optional[int] = None,
                              variant: int = OKAPI, k1: float = 1.5, b: float = 0.75, epsilon: float = 0.25,
                              flat: Optional[np.ndarray] = None, doc_lens: Optional[np.ndarray] = None,
                              compute_payload: bool = True) -> BM25Index:
    """Index a corpus given as token-id lists (or as one flat id array + per-document lengths)."""
    if flat is None:
        flat, doc_lens = _flatten(corpus_ids)
    flat = np.asarray(flat, dtype=np.int64)
    doc_lens = np.asarray(doc_lens, dtype=np.int64)
    n_docs = int(doc_lens.shape[0])
    if n_docs == 0:
        raise ValueError("empty corpus")
    if n_vocab is None:
        n_vocab = int(flat.max()) + 1 if flat.size else 1
    if flat.size and (flat.min() < 0 or flat.max() >= n_vocab):
        raise ValueError("token id out of range")
    doc_of = np.repeat(np.arange(n_docs, dtype=np.int64), doc_lens)
    # (term, doc) pairs -> unique with counts; the key order is (term asc, doc asc) = CSR by term
    key = flat * n_docs + doc_of
    ukey, tf = np.unique(key, return_counts=True)
    term = ukey // n_docs
    doc = (ukey - term * n_docs).astype(np.int32)
    tf = tf.astype(np.int32)
    df = np.bincount(term, minlength=n_vocab).astype(np.int64)
    indptr = np.zeros(n_vocab + 1, np.int64)
    np.cumsum(df, out=indptr[1:])
    if flat.size:
        uterm, first = np.unique(flat, return_index=True)
        order = uterm[np.argsort(first, kind="stable")]
    else:
        order = np.zeros(0, np.int64)
    return _finish(variant, n_docs, n_vocab, indptr, doc, tf, doc_lens, order, k1, b, epsilon, compute_payload)


def vocab_ids_python(corpus: Sequence[Sequence[Hashable]]):
    """Tokenised corpus -> (vocab {token: id by first appearance}, flat int32 id stream, int32 tokens per document) with a
    Python dict: the general path (any hashable token) and the checker of the native one."""
    vocab: Dict[Hashable, int] = {}
    lens = np.fromiter((len(d) for d in corpus), dtype=np.int32, count=len(corpus))
    flat = np.empty(int(lens.sum()), np.int32)
    p = 0
    for docu in corpus:
        for tok in docu:
            j = vocab.get(tok)
            if j is None:
                j = len(vocab)
                vocab[tok] = j
            flat[p] = j
            p += 1
    return vocab, flat, lens


def vocab_ids(corpus: Sequence[Sequence[Hashable]], native: bool = False):
    """Tokenised corpus -> (vocab, flat int32 id stream, int32 tokens per document); ids follow first appearance.
    Default: the Python dict loop -- for token lists that already exist as Python str objects it is the faster one
    (their hashes are cached; profiles/r03_text_native.log: 1.8 s against 3.2 s per 11M tokens, the native call spends
    its time re-encoding the strings).  native=True sends string tokens through the library's hash table
    (erh_vocab_encode; `vocab` is then a ``NativeVocab``) -- same ids.  The path that removes the per-token Python work
    altogether is text -> ids inside the library (``NativeCutter.encode_texts``, what BM25Retriever uses when its
    tokenizer is a NativeCutter)."""
    if native:
        from .text import NativeVocab
        if all(NativeVocab.representable(d) for d in corpus):
            vocab = NativeVocab()
            flat, lens = vocab.encode(corpus, add=True)
            return vocab, flat, lens
    return vocab_ids_python(corpus)


def build_bm25_index(corpus: Sequence[Sequence[Hashable]], variant: int = OKAPI, k1: float = 1.5, b: float = 0.75,
                     epsilon: float = 0.25, compute_payload: bool = True) -> BM25Index:
    """Index a tokenised corpus (list of token lists) on the host; vocabulary ids follow first appearance."""
    vocab, flat, lens = vocab_ids(corpus)
    idx = build_bm25_index_from_ids(None, max(len(vocab), 1), variant, k1, b, epsilon, flat=flat, doc_lens=lens,
                                    compute_payload=compute_payload)
    idx.vocab = vocab
    return idx
