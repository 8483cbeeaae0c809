"""Oracle: rank_bm25.BM25Okapi (v0.2.2) restated.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference instantiates it at src/easyrag/custom/retrievers.py:113-118
(``BM25Okapi(corpus, k1=1.5, b=0.75, epsilon=0.25)``) and queries it at retrievers.py:150
(``bm25.get_scores(tokenized_query)``).  rank-bm25 is a pip dependency
(requirements.txt:102) that is not vendored; this file follows its published algorithm
(SURVEY.md Appendix A.1), keeping the exact floating-point operation order:

  build   doc_len[i] = len(doc_i); doc_freqs[i] = {term: tf}; nd[term] = #docs with term
          (dict insertion order = first appearance while scanning docs in order, terms in
          first-appearance order inside each doc); avgdl = sum(doc_len) / N
  idf     idf[t] = log(N - nd + 0.5) - log(nd + 0.5)   (math.log, Python floats)
          idf_sum accumulated sequentially in nd order; average_idf = idf_sum / |vocab|
          every t with idf < 0 gets idf = epsilon * average_idf
  score   for q in query (in order, repeats included):
            q_freq = [(doc.get(q) or 0) for doc in doc_freqs]
            score += (idf.get(q) or 0) * (q_freq * (k1 + 1)
                                          / (q_freq + k1 * (1 - b + b * doc_len / avgdl)))
"""
from __future__ import annotations

import math
from typing import Dict, Hashable, List, Sequence

import numpy as np


class BM25Okapi:
    def __init__(self, corpus: Sequence[Sequence[Hashable]], k1: float = 1.5, b: float = 0.75,
                 epsilon: float = 0.25):
        self.k1 = k1
        self.b = b
        self.epsilon = epsilon
        self.corpus_size = 0
        self.avgdl = 0.0
        self.doc_freqs: List[Dict[Hashable, int]] = []
        self.idf: Dict[Hashable, float] = {}
        self.doc_len: List[int] = []
        nd = self._initialize(corpus)
        self._calc_idf(nd)

    def _initialize(self, corpus):
        nd: Dict[Hashable, int] = {}
        num_doc = 0
        for document in corpus:
            self.doc_len.append(len(document))
            num_doc += len(document)
            frequencies: Dict[Hashable, int] = {}
            for word in document:
                if word not in frequencies:
                    frequencies[word] = 0
                frequencies[word] += 1
            self.doc_freqs.append(frequencies)
            for word in frequencies:
                if word in nd:
                    nd[word] += 1
                else:
                    nd[word] = 1
            self.corpus_size += 1
        self.avgdl = num_doc / self.corpus_size
        return nd

    def _calc_idf(self, nd):
        idf_sum = 0
        negative_idfs = []
        for word, freq in nd.items():
            idf = math.log(self.corpus_size - freq + 0.5) - math.log(freq + 0.5)
            self.idf[word] = idf
            idf_sum += idf
            if idf < 0:
                negative_idfs.append(word)
        self.average_idf = idf_sum / len(self.idf)
        eps = self.epsilon * self.average_idf
        for word in negative_idfs:
            self.idf[word] = eps

    def get_scores(self, query: Sequence[Hashable]) -> np.ndarray:
        """Literal restatement (one Python pass over all N doc dicts per query token)."""
        score = np.zeros(self.corpus_size)
        doc_len = np.array(self.doc_len)
        for q in query:
            q_freq = np.array([(doc.get(q) or 0) for doc in self.doc_freqs])
            score += (self.idf.get(q) or 0) * (q_freq * (self.k1 + 1) /
                                               (q_freq + self.k1 * (1 - self.b + self.b * doc_len / self.avgdl)))
        return score

    # ---- test helper: the same arithmetic evaluated over postings only -------------------
    def build_postings(self):
        """term -> (doc index array ascending, tf array).  Used by the oracle's own
        consistency test (dense dict loop == sparse postings) and by nothing else."""
        post: Dict[Hashable, List] = {}
        for i, freqs in enumerate(self.doc_freqs):
            for w, tf in freqs.items():
                post.setdefault(w, [[], []])
                post[w][0].append(i)
                post[w][1].append(tf)
        return {w: (np.asarray(v[0], dtype=np.int64), np.asarray(v[1], dtype=np.int64))
                for w, v in post.items()}

    def get_scores_sparse(self, query, postings=None) -> np.ndarray:
        """Same float64 operation order as get_scores, touching only posting entries.
        Non-posting docs receive ``idf * 0.0`` in the literal form, which never changes
        the accumulated bits (x + (+-0.0) == x, and the accumulator starts at +0.0)."""
        if postings is None:
            postings = self.build_postings()
        score = np.zeros(self.corpus_size)
        doc_len = np.array(self.doc_len)
        for q in query:
            if q not in postings:
                continue
            docs, tf = postings[q]
            contrib = (self.idf.get(q) or 0) * (tf * (self.k1 + 1) /
                                                (tf + self.k1 * (1 - self.b + self.b * doc_len[docs] / self.avgdl)))
            score[docs] = score[docs] + contrib
        return score
