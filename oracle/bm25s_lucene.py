"""Oracle: bm25s.BM25 (v0.1.7, method="lucene", float32) restated.  TEST INFRASTRUCTURE ONLY.

The reference builds it at src/easyrag/custom/retrievers.py:107-111
(``bm25s.BM25(k1=1.5, b=0.75); bm25.index(corpus)``) and queries it at retrievers.py:150.
bm25s is a pip dependency (requirements.txt:113), not vendored; this file follows its
published algorithm:

  index   vocab = {token: id}; l_avg = mean(len(doc)) (np.float64)
          idf[t] = log(1 + (N - df + 0.5) / (df + 0.5))  -> stored float32
          per doc, per unique term:  tfc = tf / (k1 * ((1 - b) + b * l_d / l_avg) + tf)
          with tf a float32 array and the bracket a float64 *scalar*: under the pinned
          numpy==1.26.4 (requirements.txt:61) value-based casting keeps float32, i.e. the
          scalar is rounded to float32 once and the add and the divide are float32 ops.
          numpy >= 2 would promote to float64, so the casts below are explicit.
          score = idf[t] * tfc (float32); assembled as CSC (N x V): data/indices/indptr,
          doc indices ascending inside each term ("eager scoring").
  query   ids = [vocab[t] for t in tokens if t in vocab]   (OOV dropped, repeats kept)
          scores = zeros(N, float32); for id in ids (in order):
              np.add.at(scores, indices[s:e], data[s:e])   (float32 adds)
"""
from __future__ import annotations

import math
from collections import Counter
from typing import Dict, Hashable, List, Sequence

import numpy as np


class BM25SLucene:
    def __init__(self, k1: float = 1.5, b: float = 0.75):
        self.k1 = k1
        self.b = b
        self.vocab_dict: Dict[Hashable, int] = {}
        self.data = np.zeros(0, np.float32)
        self.indices = np.zeros(0, np.int32)
        self.indptr = np.zeros(1, np.int64)
        self.num_docs = 0

    def index(self, corpus: Sequence[Sequence[Hashable]]):
        # vocabulary: ids follow first appearance (bm25s iterates a Python set; the id
        # assignment is irrelevant to the scores, only to the column order)
        vocab: Dict[Hashable, int] = {}
        for doc in corpus:
            for tok in doc:
                if tok not in vocab:
                    vocab[tok] = len(vocab)
        self.vocab_dict = vocab
        n_docs = len(corpus)
        n_vocab = len(vocab)
        self.num_docs = n_docs
        corpus_ids = [[vocab[t] for t in doc] for doc in corpus]
        l_avg = np.array([len(d) for d in corpus_ids]).mean() if n_docs else np.float64(0.0)

        df = np.zeros(n_vocab, np.int64)
        for ids in corpus_ids:
            for t in set(ids):
                df[t] += 1
        idf = np.zeros(n_vocab, np.float32)
        for t in range(n_vocab):
            idf[t] = math.log(1 + (n_docs - int(df[t]) + 0.5) / (int(df[t]) + 0.5))
        self.idf = idf
        self.df = df

        nnz = int(df.sum())
        scores = np.empty(nnz, np.float32)
        doc_idx = np.empty(nnz, np.int32)
        voc_idx = np.empty(nnz, np.int32)
        i = 0
        k1, b = self.k1, self.b
        for d, ids in enumerate(corpus_ids):
            cnt = Counter(ids)
            voc = np.array(list(cnt.keys()), dtype=np.int32)
            tf = np.array(list(cnt.values()), dtype=np.float32)
            l_d = len(ids)
            bracket = np.float32(k1 * ((1 - b) + b * l_d / l_avg))  # float64 scalar -> f32 once
            tfc = tf / (bracket + tf)                                   # float32 add, float32 div
            sc = idf[voc] * tfc                                         # float32 mul
            n = len(voc)
            scores[i:i + n] = sc
            doc_idx[i:i + n] = d
            voc_idx[i:i + n] = voc
            i += n
        # CSC assembly: sort by (term, doc); docs were appended in ascending order so a stable
        # sort on term keeps them ascending inside each term (what scipy's coo->csc yields).
        order = np.argsort(voc_idx, kind="stable")
        self.data = scores[order]
        self.indices = doc_idx[order]
        self.indptr = np.zeros(n_vocab + 1, np.int64)
        np.cumsum(np.bincount(voc_idx, minlength=n_vocab), out=self.indptr[1:])
        return self

    def get_tokens_ids(self, tokens: Sequence[Hashable]) -> List[int]:
        return [self.vocab_dict[t] for t in tokens if t in self.vocab_dict]

    def get_scores_from_ids(self, ids: Sequence[int]) -> np.ndarray:
        scores = np.zeros(self.num_docs, dtype=np.float32)
        for t in ids:
            s, e = self.indptr[t], self.indptr[t + 1]
            np.add.at(scores, self.indices[s:e], self.data[s:e])
        return scores

    def get_scores(self, tokens: Sequence[Hashable]) -> np.ndarray:
        # bm25s 0.1.7 indexes tokens[0] to sniff the type and raises IndexError on an empty
        # list (SURVEY.md A.2 "(?)"); the reference never guards it (retrievers.py:148-150).
        if len(tokens) == 0:
            raise IndexError("list index out of range")
        return self.get_scores_from_ids(self.get_tokens_ids(tokens))
