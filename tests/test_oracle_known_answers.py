"""Hand-computed known-answer cases that pin the oracle (the reference ships no tests)."""
import math

import numpy as np

from oracle import BM25Okapi, BM25SLucene, bm25_filter, reciprocal_rank_fusion, fusion, hybrid_retrieve
from oracle import dense_exact_scores, dense_exact_topk, qdrant_cosine_search, to_f16_unit, canonical_order
from oracle.retrievers import Item, tokenize_and_remove_stopwords

CORPUS = [["a", "b", "a"], ["b", "c"], ["c", "c", "c", "d"]]


def test_okapi_micro_case():
    bm = BM25Okapi(CORPUS, k1=1.5, b=0.75, epsilon=0.25)
    n, avgdl = 3, 9 / 3
    assert bm.corpus_size == n and bm.avgdl == avgdl
    # df: a 1, b 2, c 2, d 1
    idf_a = math.log(3 - 1 + 0.5) - math.log(1 + 0.5)
    idf_b = math.log(3 - 2 + 0.5) - math.log(2 + 0.5)     # negative -> epsilon floor
    assert idf_b < 0
    # insertion order of nd: a, b, c, d
    avg = (((0 + idf_a) + idf_b) + idf_b) + idf_a
    avg = avg / 4
    assert bm.average_idf == avg
    assert bm.idf["a"] == idf_a and bm.idf["d"] == idf_a
    assert bm.idf["b"] == 0.25 * avg and bm.idf["c"] == 0.25 * avg
    s = bm.get_scores(["a", "c", "zzz"])
    # doc0: tf(a)=2, dl=3
    exp0 = idf_a * (2 * 2.5 / (2 + 1.5 * (1 - 0.75 + 0.75 * 3 / avgdl)))
    # doc1: tf(c)=1, dl=2 ; doc2: tf(c)=3, dl=4
    exp1 = (0.25 * avg) * (1 * 2.5 / (1 + 1.5 * (1 - 0.75 + 0.75 * 2 / avgdl)))
    exp2 = (0.25 * avg) * (3 * 2.5 / (3 + 1.5 * (1 - 0.75 + 0.75 * 4 / avgdl)))
    assert s[0] == exp0 and s[1] == exp1 and s[2] == exp2
    # repeats count twice, in order
    s2 = bm.get_scores(["a", "a"])
    assert s2[0] == exp0 + exp0 and s2[1] == 0.0
    # empty query -> zeros -> filter returns []
    assert bm25_filter(bm.get_scores([]), 5) == []


def test_okapi_sparse_equals_dense_loop():
    rng = np.random.default_rng(0)
    corpus = [list(rng.integers(0, 30, size=rng.integers(1, 25))) for _ in range(200)]
    bm = BM25Okapi(corpus)
    post = bm.build_postings()
    for _ in range(20):
        q = list(rng.integers(0, 35, size=rng.integers(1, 9)))
        a, b = bm.get_scores(q), bm.get_scores_sparse(q, post)
        assert np.array_equal(a.view(np.uint64), b.view(np.uint64))


def test_bm25s_micro_case():
    bm = BM25SLucene(k1=1.5, b=0.75).index(CORPUS)
    l_avg = np.array([3, 2, 4]).mean()
    idf_c = np.float32(math.log(1 + (3 - 2 + 0.5) / (2 + 0.5)))
    # doc2: tf(c) = 3, l_d = 4
    br = np.float32(1.5 * ((1 - 0.75) + 0.75 * 4 / l_avg))
    tfc = np.float32(3) / (br + np.float32(3))
    exp2 = idf_c * tfc
    s = bm.get_scores(["c"])
    assert s.dtype == np.float32 and s[2] == exp2 and s[0] == 0
    # accumulation is float32, in query order, OOV dropped
    s3 = bm.get_scores(["c", "nope", "c", "d"])
    idf_d = np.float32(math.log(1 + (3 - 1 + 0.5) / (1 + 0.5)))
    exp_d = idf_d * (np.float32(1) / (br + np.float32(1)))
    assert s3[2] == np.float32(np.float32(exp2 + exp2) + exp_d)
    # docs ascending inside each term, indptr consistent
    for t in range(len(bm.vocab_dict)):
        seg = bm.indices[bm.indptr[t]:bm.indptr[t + 1]]
        assert np.all(np.diff(seg) > 0)
    try:
        bm.get_scores([])
        assert False, "bm25s raises on an empty token list"
    except IndexError:
        pass


def test_filter_semantics():
    scores = np.array([0.0, 2.0, -1.0, 2.0, 3.0, 1.0])
    assert bm25_filter(scores, 10) == [(4, 3.0), (1, 2.0), (3, 2.0), (5, 1.0)]      # <= 0 cut, ties by index
    assert bm25_filter(scores, 2) == [(4, 3.0), (1, 2.0)]
    mask = np.array([1, 0, 1, 1, 0, 1], bool)
    assert bm25_filter(scores, 2, mask) == [(3, 2.0), (5, 1.0)]
    lit = bm25_filter(scores, 10, tie="literal")
    assert sorted(lit) == sorted(bm25_filter(scores, 10)) and [s for _, s in lit] == [3.0, 2.0, 2.0, 1.0]
    assert list(canonical_order(scores)[:3]) == [4, 1, 3]


def test_rrf_known_answer():
    sparse = [Item(10, "x", 5.0), Item(11, "y", 4.0), Item(12, "z", 3.0)]
    dense = [Item(21, "y", 0.9), Item(20, "x", 0.8), Item(22, "w", 0.7), Item(23, "x", 0.1)]
    out = reciprocal_rank_fusion([sparse, dense], K=60, topk=10)
    x = 0.0 + 1 / 61
    x = x + 1 / 62
    x = x + 1 / 64           # "x" occurs twice in the dense list: both occurrences count
    y = (0.0 + 1 / 62) + 1 / 61
    z = 1 / 63
    w = 1 / 63
    got = [(o.content, o.score, o.idx) for o in out]
    # x > y > z == w ; z first (seen first: sparse list comes first); returned node = LAST seen for the content
    assert got == [("x", x, 23), ("y", y, 21), ("z", z, 12), ("w", w, 22)]
    assert [o.content for o in reciprocal_rank_fusion([sparse, dense], topk=2)] == ["x", "y"]
    assert reciprocal_rank_fusion([[], []]) == []


def test_fusion_known_answer():
    a = [Item(1, "p", 3.0), Item(2, "q", 1.0)]
    b = [Item(3, "q", 9.0), Item(4, "r", 3.0), Item(5, "s", 0.5)]
    out = fusion([a, b], topk=10)
    # q keeps its FIRST occurrence (score 1.0, idx 2); p before r on the 3.0 tie (first seen)
    assert [(o.content, o.idx, o.score) for o in out] == [("p", 1, 3.0), ("r", 4, 3.0), ("q", 2, 1.0), ("s", 5, 0.5)]
    assert len(fusion([a, b], topk=2)) == 2


def test_hybrid_routes():
    s = lambda: [Item(1, "a", 1.0)]
    d = lambda: [Item(2, "b", 1.0)]
    assert [i.idx for i in hybrid_retrieve(1, s, d)] == [2]
    assert [i.idx for i in hybrid_retrieve(2, s, d)] == [1]
    assert [i.idx for i in hybrid_retrieve(3, s, d)] == [1, 2]       # sparse list first on the tie


def test_tokenize_and_remove_stopwords():
    class T:
        def cut(self, t):
            return list(t)
    assert tokenize_and_remove_stopwords(T(), "a b,c", {","}) == ["a", "b", "c"]


def test_dense_exact_and_qdrant():
    rng = np.random.default_rng(5)
    for d in (64, 768, 1024):
        x = to_f16_unit(rng.standard_normal((300, d)))
        q = to_f16_unit(rng.standard_normal(d))
        s = dense_exact_scores(x, q)
        ref = x.astype(np.float64) @ q.astype(np.float64)
        assert np.max(np.abs(s - ref)) < 1e-14
        ids, sc = dense_exact_topk(x, q, 10)
        assert list(ids) == list(np.lexsort((np.arange(300), -s))[:10]) and np.array_equal(sc, s[ids])
        qi, qs = qdrant_cosine_search(x.astype(np.float32), q.astype(np.float32), 10)
        assert np.max(np.abs(qs - s[qi])) < 1e-3 and set(qi) == set(ids)
    # duplicates tie exactly and are ordered by index; mask is honoured
    x = np.repeat(to_f16_unit(rng.standard_normal((4, 64))), 3, axis=0)
    q = x[4]
    ids, sc = dense_exact_topk(x, q, 3)
    assert list(ids) == [3, 4, 5] and sc[0] == sc[1] == sc[2]
    m = np.ones(12, bool)
    m[3] = False
    assert list(dense_exact_topk(x, q, 2, m)[0]) == [4, 5]
