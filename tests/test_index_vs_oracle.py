"""Host logic: the product's vectorised index build (easyrag_amd/index.py) against the oracle's literal
restatements of rank_bm25 / bm25s.  Summing the product's per-posting payloads per document in query-token
order must reproduce the libraries' score vectors bit for bit -- that sum is what the HIP scan kernel does."""
import numpy as np
import pytest

from easyrag_amd.index import BM25S, OKAPI, build_bm25_index, build_bm25_index_from_ids, build_bm25_index_from_postings
from oracle import BM25Okapi, BM25SLucene


def host_sum(idx, q_ids):
    acc = np.zeros(idx.n_docs, idx.payload.dtype)
    for t in q_ids:
        s, e = idx.indptr[t], idx.indptr[t + 1]
        acc[idx.doc_ids[s:e]] = acc[idx.doc_ids[s:e]] + idx.payload[s:e]
    return acc


def random_corpus(rng, n_docs, vocab, lo=1, hi=30):
    return [[f"t{int(x)}" for x in rng.integers(0, vocab, size=rng.integers(lo, hi))] for _ in range(n_docs)]


@pytest.mark.parametrize("seed,n_docs,vocab", [(0, 50, 12), (1, 400, 300), (2, 1500, 80), (3, 7, 3)])
def test_okapi_payload_sums_match_rank_bm25(seed, n_docs, vocab):
    rng = np.random.default_rng(seed)
    corpus = random_corpus(rng, n_docs, vocab)
    ora = BM25Okapi(corpus, k1=1.5, b=0.75, epsilon=0.25)
    idx = build_bm25_index(corpus, OKAPI)
    assert idx.avgdl == ora.avgdl and idx.average_idf == ora.average_idf
    for tok, j in idx.vocab.items():
        assert idx.idf[j] == ora.idf[tok]
    assert np.all(np.diff(idx.indptr) >= 0) and idx.indptr[-1] == idx.nnz
    for _ in range(15):
        q = [f"t{int(x)}" for x in rng.integers(0, vocab + 3, size=rng.integers(0, 10))]
        want = ora.get_scores(q)
        got = host_sum(idx, idx.tokens_to_ids(q))
        assert np.array_equal(got.view(np.uint64), want.view(np.uint64))


def test_okapi_negative_idf_epsilon_floor():
    # a term in > half of the documents gets idf = epsilon * average_idf (possibly negative for tiny corpora)
    corpus = [["x", "y"], ["x"], ["x", "z"], ["x", "y", "y"]]
    ora = BM25Okapi(corpus)
    idx = build_bm25_index(corpus, OKAPI)
    assert ora.idf["x"] == 0.25 * ora.average_idf
    for q in (["x"], ["x", "y", "x"], ["z", "x"]):
        assert np.array_equal(host_sum(idx, idx.tokens_to_ids(q)).view(np.uint64), ora.get_scores(q).view(np.uint64))


@pytest.mark.parametrize("seed,n_docs,vocab", [(0, 50, 12), (1, 400, 300), (2, 1500, 80)])
def test_bm25s_payload_equals_library_matrix(seed, n_docs, vocab):
    rng = np.random.default_rng(seed)
    corpus = random_corpus(rng, n_docs, vocab)
    ora = BM25SLucene().index(corpus)
    idx = build_bm25_index(corpus, BM25S)
    assert idx.vocab == ora.vocab_dict
    assert np.array_equal(idx.indptr, ora.indptr) and np.array_equal(idx.doc_ids, ora.indices)
    assert idx.payload.dtype == np.float32 and np.array_equal(idx.payload.view(np.uint32), ora.data.view(np.uint32))
    for _ in range(15):
        q = [f"t{int(x)}" for x in rng.integers(0, vocab + 3, size=rng.integers(1, 10))]
        want = ora.get_scores(q)
        got = host_sum(idx, idx.tokens_to_ids(q))
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_from_ids_and_from_postings_agree():
    rng = np.random.default_rng(9)
    ids = [rng.integers(0, 40, size=rng.integers(1, 20)) for _ in range(300)]
    for variant in (OKAPI, BM25S):
        a = build_bm25_index_from_ids(ids, 40, variant)
        order = None
        if variant == OKAPI:
            flat = np.concatenate(ids)
            u, first = np.unique(flat, return_index=True)
            order = u[np.argsort(first, kind="stable")]
        b = build_bm25_index_from_postings(a.indptr, a.doc_ids, a.tf, a.doc_len, variant, first_seen_order=order)
        assert np.array_equal(a.payload, b.payload) and np.array_equal(a.idf, b.idf)
        c = build_bm25_index_from_ids(ids, 40, variant, compute_payload=False)
        assert c.payload.shape[0] == 0 and np.array_equal(c.tf, a.tf)
    with pytest.raises(ValueError):
        build_bm25_index_from_ids([[0, 99]], 10, OKAPI)
