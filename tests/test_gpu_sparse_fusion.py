"""GPU parity, sparse route + fusion + the fused dual route: HIP path (C ABI) vs the oracle.
Bar: ids and scores bit-identical (float64 for Okapi, float32 for bm25s) under canonical ties."""
import numpy as np
import pytest

from easyrag_amd import synth
from easyrag_amd.engine import queries_to_csr
from easyrag_amd.index import BM25S, OKAPI, build_bm25_index, build_bm25_index_from_ids
from oracle import (BM25Okapi, BM25SLucene, bm25_filter, dense_exact_topk, fusion, reciprocal_rank_fusion,
                    to_f16_unit)
from oracle.retrievers import Item

pytestmark = pytest.mark.gpu


@pytest.fixture(params=[(1, 0, 1, 2, 1), (1, 0, 1, 2, 0), (1, 0, 1, 1, 1), (1, 0, 1, 0, 1), (0, 1, 2, 2, 1), (0, 1, 0, 2, 1),
                        (0, 0, 0, 2, 1), (1, 0, 1, 2, 1, 1), (1, 0, 1, 1, 1, 1)],
                ids=["approx-scan-packed-4byte", "approx-scan-packed-8byte", "approx-scan-small", "approx-scan-big",
                     "wave-owned-crossings", "wave-owned-sweep", "block-scan", "approx-scan-packed-4byte-split-finish",
                     "approx-scan-small-split-finish"])
def bm25_kernel(request, engine):
    """Every BM25 scan kernel / survivor-selection path must satisfy every parity test (bm25_ascan / bm25_wscan take
    effect at the next set_bm25)."""
    engine.set_option("bm25_ascan", request.param[0])
    engine.set_option("bm25_wscan", request.param[1])
    engine.set_option("bm25_crossing", request.param[2])
    engine.set_option("bm25_small", request.param[3])       # shape of the approximate scan for batches of >= 8 queries
    engine.set_option("bm25_post16", request.param[4])      # packed shape: 4-byte or 8-byte postings
    engine.set_option("bm25_split_finish", request.param[5] if len(request.param) > 5 else 0)   # exact re-score + rank in a kernel of its own
    yield tuple(request.param) + ((0,) if len(request.param) == 5 else ())
    engine.set_option("bm25_split_finish", 0)
    engine.set_option("bm25_ascan", 1)
    engine.set_option("bm25_wscan", 0)
    engine.set_option("bm25_crossing", 1)
    engine.set_option("bm25_small", 2)
    engine.set_option("bm25_post16", 1)


def _oracle_for(variant, docs):
    if variant == OKAPI:
        return BM25Okapi(docs, k1=1.5, b=0.75, epsilon=0.25)
    return BM25SLucene(k1=1.5, b=0.75).index(docs)


def _oracle_scores(ora, variant, q):
    if variant == BM25S and len(q) == 0:
        return np.zeros(ora.num_docs, np.float32)
    return ora.get_scores(q)


_HOST_CACHE = {}


def _cached(key, fn):
    """Host-side work that is a pure function of a test's seeds (the token lists, the oracle object, the host-built index, the oracle's
    answer for one query): evaluated once and shared by the nine kernel arms of a parametrized test -- the arms differ on the device only."""
    if key not in _HOST_CACHE:
        _HOST_CACHE[key] = fn()
    return _HOST_CACHE[key]


@pytest.mark.parametrize("variant", [OKAPI, BM25S])
@pytest.mark.parametrize("n_docs,vocab,seed", [(60, 10, 0), (3000, 500, 1), (40000, 2000, 2)])
def test_bm25_scores_and_topk_match_oracle(engine, bm25_kernel, variant, n_docs, vocab, seed):
    flat, lens = synth.token_corpus(n_docs, vocab, seed=seed, mean_len=20)
    docs = [list(map(int, d)) for d in synth.split_docs(flat, lens)]
    ora = _oracle_for(variant, docs)
    idx = build_bm25_index(docs, variant)
    engine.set_bm25(idx)
    rng = np.random.default_rng(seed + 100)
    queries = [list(map(int, q)) for q in synth.token_queries(flat, lens, vocab, 24, seed=seed + 7)]
    queries[3] = []                                                   # empty query
    queries[4] = [vocab + 5, vocab + 6]                                # all out of vocabulary
    queries[5] = queries[6] + queries[6]                               # repeats count twice
    oscores = [_oracle_scores(ora, variant, q) for q in queries]
    # get_scores parity (dense vector)
    for b in list(range(6)) + [21, 22, 23]:
        got = engine.bm25_scores(idx.tokens_to_ids(queries[b]))
        assert np.array_equal(got, oscores[b].astype(np.float64)), "score vector differs"
    # fused scan + top-k parity
    qi, qt = queries_to_csr([idx.tokens_to_ids(q) for q in queries])
    for k in (1, 10, 192):
        ids, sc, ln = engine.bm25_topk(qi, qt, k)
        for b, q in enumerate(queries):
            want = bm25_filter(oscores[b], k)
            assert ln[b] == len(want)
            assert list(ids[b, :ln[b]]) == [w[0] for w in want], f"k={k} query {b}: ids differ"
            assert list(sc[b, :ln[b]]) == [w[1] for w in want], f"k={k} query {b}: scores differ"
            assert np.all(ids[b, ln[b]:] == -1)


@pytest.mark.parametrize("variant", [OKAPI, BM25S])
def test_bm25_ties_and_filter(engine, bm25_kernel, variant):
    # 16 distinct documents repeated 40 times each: exact score ties across the whole corpus, ordered by index
    rng = np.random.default_rng(4)
    base = [list(map(int, rng.integers(0, 12, size=rng.integers(3, 9)))) for _ in range(16)]
    docs = [base[i % 16] for i in range(640)]
    ora = _oracle_for(variant, docs)
    idx = build_bm25_index(docs, variant)
    engine.set_bm25(idx)
    dir_id = (np.arange(640) // 100).astype(np.int16)
    engine.set_doc_meta(640, None, dir_id)
    # (every query twice: batches of >= 8 queries take the two-workgroups-per-CU shapes of the approximate scan)
    queries = [[0, 1, 2], [5], [3, 3, 7, 11], [2, 9]] * 2
    qi, qt = queries_to_csr([idx.tokens_to_ids(q) for q in queries])
    filt = np.tile(np.array([-1, 2, 6, 0], np.int16), 2)
    for k in (7, 100):
        ids, sc, ln = engine.bm25_topk(qi, qt, k, filter_dir=filt)
        for b, q in enumerate(queries):
            mask = None if filt[b] < 0 else dir_id == filt[b]
            want = bm25_filter(_oracle_scores(ora, variant, q), k, mask)
            assert list(ids[b, :ln[b]]) == [w[0] for w in want] and list(sc[b, :ln[b]]) == [w[1] for w in want]


@pytest.mark.parametrize("variant", [OKAPI, BM25S])
def test_bm25_near_tie_flood(engine, bm25_kernel, variant):
    """Five distinct documents, 1800 copies each: thousands of exact ties around any k-th score.  The approximate-order
    scan cannot shrink its list below capacity (every tied document lies within the margin) and must hand the query to
    the exact block scan; one- and two-token queries (order-independent sums) and longer ones, with and without filter."""
    rng = np.random.default_rng(12)
    base = [list(map(int, rng.integers(0, 9, size=rng.integers(3, 8)))) for _ in range(5)]
    n = 9000
    docs = [base[i % 5] for i in range(n)]
    ora = _oracle_for(variant, docs)
    idx = build_bm25_index(docs, variant)
    engine.set_bm25(idx)
    dir_id = (np.arange(n) % 3).astype(np.int16)
    engine.set_doc_meta(n, None, dir_id)
    present = sorted({t for d in base for t in d})
    queries = [[present[0]], present[:2], present[:4] + [present[1]], list(present), [present[-1]] * 3 + present[:3]] * 2
    qi, qt = queries_to_csr([idx.tokens_to_ids(q) for q in queries])
    for filt in (None, np.tile(np.array([0, -1, 2, 1, 0], np.int16), 2)):
        for k in (3, 192, 1000):
            ids, sc, ln = engine.bm25_topk(qi, qt, k, filter_dir=filt)
            for b, q in enumerate(queries):
                mask = None if (filt is None or filt[b] < 0) else dir_id == filt[b]
                want = bm25_filter(_oracle_scores(ora, variant, q), k, mask)
                assert ln[b] == len(want)
                assert list(ids[b, :ln[b]]) == [w[0] for w in want], f"k={k} query {b}: ids differ"
                assert list(sc[b, :ln[b]]) == [w[1] for w in want], f"k={k} query {b}: scores differ"


@pytest.mark.parametrize("variant", [OKAPI, BM25S])
def test_bm25_mixed_launch_cuts_long_queries(engine, variant):
    """>= 512 queries (one workgroup per query), a few of them longer than bm25_long_tokens: ONE launch whose workgroups pick their body by the
    length of their query, and a long query's documents cut into bm25_long_segs ranges -- items query | segment << 24 built on the host, the
    long queries' partial lists merged into the caller's rows.  Half of the long queries ask for a document that exists 2000 times (spread over
    the whole corpus): every one of their segments floods its list and goes through the exact block scan before the merge.  Every arm --
    cut / not cut / the whole batch on the 32-bit shape / always packed -- gives the same lists; the long queries and a sample of the others
    against the oracle, ids and scores bit for bit; with and without a dir filter."""
    rng = np.random.default_rng(77)
    n, vocab, b, k = 70000, 5000, 520, 60
    flat, lens = synth.token_corpus(n, vocab, seed=41, mean_len=20)
    docs = [list(map(int, d)) for d in synth.split_docs(flat, lens)]
    base = [list(map(int, rng.integers(0, vocab, size=int(rng.integers(12, 20))))) for _ in range(5)]
    for i in range(0, n, 7):                                          # 10000 copies of five documents, everywhere in the corpus
        docs[i] = base[(i // 7) % 5]
    idx = build_bm25_index(docs, variant)
    engine.set_bm25(idx)
    dir_id = np.repeat(np.arange(3), [30000, 25000, 15000]).astype(np.int16)
    engine.set_doc_meta(n, None, dir_id)
    ora = _oracle_for(variant, docs)
    short_lengths = [L for L in synth.REF_QUESTION_LENGTHS if L <= 28]
    qs = [list(map(int, q)) for q in synth.token_queries(flat, lens, vocab, b, seed=42, lengths=short_lengths)]
    long_at = [int(i) for i in rng.choice(b, size=20, replace=False)]
    for j, i in enumerate(long_at):
        L = int(rng.integers(29, 46))
        src = base[j % 5] if j < 10 else docs[int(rng.integers(0, n))]
        qs[i] = [src[int(t)] for t in rng.integers(0, len(src), L)]
    csr = queries_to_csr([idx.tokens_to_ids(q) for q in qs])
    filt = (np.arange(b) % 4 - 1).astype(np.int16)                   # none, 0, 1, 2
    try:
        for f in (None, filt):
            got = {}
            for name, opts in (("cut", (28, 1, 2)), ("cut3", (28, 1, 3)), ("mixed", (28, 1, 1)), ("wide", (28, 0, 2)), ("packed", (0, 1, 2))):
                engine.set_option("bm25_long_tokens", opts[0])
                engine.set_option("bm25_mixed", opts[1])
                engine.set_option("bm25_long_segs", opts[2])
                engine.reset_stats()
                got[name] = engine.bm25_topk(*csr, k, filter_dir=f)
                assert engine.stat("bm25_mixed_launches") == (1 if name in ("cut", "cut3", "mixed") else 0), name
                if name == "cut" and f is None:
                    assert engine.stat("bm25_redo_segments") >= 20, "the flooding long queries were meant to reach the exact scan in both ranges"
            ids, sc, ln = got["cut"]
            for other in ("cut3", "mixed", "wide", "packed"):
                assert np.array_equal(ln, got[other][2]) and np.array_equal(ids, got[other][0]), other
                assert np.array_equal(sc.view(np.uint64), got[other][1].view(np.uint64)), other
            for i in sorted(set(long_at + list(range(0, b, 29)))):
                mask = None if f is None or f[i] < 0 else dir_id == f[i]
                want = bm25_filter(_oracle_scores(ora, variant, qs[i]), k, mask)
                assert list(ids[i, :ln[i]]) == [w[0] for w in want], (i, len(qs[i]))
                assert list(sc[i, :ln[i]]) == [w[1] for w in want], (i, len(qs[i]))
        # every query of the batch long: the item list is B x bm25_long_segs, nothing takes the packed body, every row comes from the merge
        all_long = [[docs[int(t)][int(u) % len(docs[int(t)])] for u in rng.integers(0, 1000, int(rng.integers(29, 46)))] for t in rng.integers(0, n, 512)]
        csr_l = queries_to_csr([idx.tokens_to_ids(q) for q in all_long])
        engine.set_option("bm25_long_tokens", 28)
        engine.set_option("bm25_mixed", 1)
        engine.set_option("bm25_long_segs", 2)
        engine.reset_stats()
        a = engine.bm25_topk(*csr_l, k)
        assert engine.stat("bm25_mixed_launches") == 1
        engine.set_option("bm25_mixed", 0)
        b_ = engine.bm25_topk(*csr_l, k)
        assert np.array_equal(a[2], b_[2]) and np.array_equal(a[0], b_[0]) and np.array_equal(a[1].view(np.uint64), b_[1].view(np.uint64))
        for i in (0, 255, 511):
            want = bm25_filter(_oracle_scores(ora, variant, all_long[i]), k, None)
            assert list(a[0][i, :a[2][i]]) == [w[0] for w in want] and list(a[1][i, :a[2][i]]) == [w[1] for w in want], i
    finally:
        engine.set_option("bm25_long_tokens", 28)
        engine.set_option("bm25_mixed", 1)
        engine.set_option("bm25_long_segs", 4)
        engine.set_doc_meta(n, None, None)


def test_bm25_okapi_negative_idf_floor(engine, bm25_kernel):
    """rank-bm25 replaces negative idf values by epsilon * average_idf -- which is itself negative when most terms sit
    in most documents.  Sums then do not grow monotonically, so the threshold-crossing path of the wave-owned scan
    must stand down (the library checks the payload signs when an index is set); scores and the `score > 0` walk still
    match the reference arithmetic."""
    rng = np.random.default_rng(8)
    docs = []
    for _ in range(3000):                                    # six terms in ~80 % of the documents, two rare ones
        doc = [t for t in range(6) if rng.random() < 0.8] or [0]
        doc += [int(t) for t in rng.integers(0, 6, size=rng.integers(0, 4))]
        doc += [6] * (rng.random() < 0.1) + [7] * (rng.random() < 0.1)
        docs.append(doc)
    ora = _oracle_for(OKAPI, docs)
    idx = build_bm25_index(docs, OKAPI)
    assert float(idx.payload.min()) < 0.0 < float(idx.payload.max())
    engine.set_bm25(idx)
    queries = [[0, 1, 7], [6, 7, 2, 2], [7], [3], [6, 0, 1, 2, 3, 4, 5]]
    qi, qt = queries_to_csr([idx.tokens_to_ids(q) for q in queries])
    for k in (5, 150):
        ids, sc, ln = engine.bm25_topk(qi, qt, k)
        for b, q in enumerate(queries):
            want = bm25_filter(_oracle_scores(ora, OKAPI, q), k)
            assert list(ids[b, :ln[b]]) == [w[0] for w in want] and list(sc[b, :ln[b]]) == [w[1] for w in want]


@pytest.mark.parametrize("variant", [OKAPI, BM25S])
def test_bm25_payload_evaluated_on_gpu_is_bit_identical(engine, variant):
    flat, lens = synth.token_corpus(5000, 800, seed=21, mean_len=30)
    idx = build_bm25_index_from_ids(flat=flat, doc_lens=lens, n_vocab=800, variant=variant)
    engine.set_bm25(idx, payload_on_device=True)
    got = engine.get_bm25_payload()
    assert got.dtype == idx.payload.dtype and np.array_equal(got, idx.payload)


def test_bm25_single_query_uses_segments_and_merge(engine, bm25_kernel):
    # B = 1 splits the document range over several workgroups and merges the partial lists
    flat, lens = synth.token_corpus(150000, 3000, seed=5, mean_len=16)
    idx = build_bm25_index_from_ids(flat=flat, doc_lens=lens, n_vocab=3000, variant=BM25S)
    engine.set_bm25(idx)
    q = synth.token_queries(flat, lens, 3000, 1, seed=1)[0]
    qi, qt = queries_to_csr([q])
    ids, sc, ln = engine.bm25_topk(qi, qt, 50)
    scores = engine.bm25_scores(q)
    want = bm25_filter(scores, 50)
    assert list(ids[0, :ln[0]]) == [w[0] for w in want] and list(sc[0, :ln[0]]) == [w[1] for w in want]
    # and the score vector itself against a numpy scatter-add in token order
    acc = np.zeros(idx.n_docs, np.float32)
    for t in q:
        s, e = idx.indptr[t], idx.indptr[t + 1]
        np.add.at(acc, idx.doc_ids[s:e], idx.payload[s:e])
    assert np.array_equal(scores, acc.astype(np.float64))


@pytest.mark.parametrize("variant", [OKAPI, BM25S])
def test_bm25_long_queries_frequent_terms_and_list_flood(engine, bm25_kernel, variant):
    """Cases the wave-owned scan treats specially: a term with far more than 64 postings per 2048-document
    sub-range (several pieces per step, several steps per tile), queries longer than 64 tokens (routed to the block
    scan), a batch mixing both, and an unseeded first tile (dir filter active) whose touched documents overflow the
    2048-entry candidate list, so the list-full / re-sweep path runs."""
    rng = np.random.default_rng(77)
    n_docs, vocab = 70000, 400
    lens = rng.integers(6, 30, size=n_docs)
    # ids 0..3 are very frequent (each in ~half of the documents), the rest Zipfian
    flat = rng.integers(4, vocab, size=int(lens.sum()))
    common = rng.random(flat.shape[0]) < 0.25
    flat[common] = rng.integers(0, 4, size=int(common.sum()))
    docs = [list(map(int, d)) for d in synth.split_docs(flat.astype(np.int64), lens.astype(np.int64))]
    ora = _oracle_for(variant, docs)
    idx = build_bm25_index(docs, variant)
    engine.set_bm25(idx)
    dir_id = (np.arange(n_docs) % 2).astype(np.int16)
    engine.set_doc_meta(n_docs, None, dir_id)
    queries = [[0, 1, 2, 3, 7], [3, 3, 0, 9, 11, 12, 200], list(range(4, 80)) + [1, 1],        # 78 tokens
               [int(t) for t in rng.integers(0, vocab, size=40)], [5], [0] * 20 + list(range(30, 70))]
    osc = [_oracle_scores(ora, variant, q) for q in queries] * 2
    queries = queries * 2                                              # (>= 8 queries per batch: the packed shape runs them too)
    qi, qt = queries_to_csr([idx.tokens_to_ids(q) for q in queries])
    short = [b for b, q in enumerate(queries) if len(q) <= 64]
    for filt in (None, np.tile(np.array([1, 0, 1, -1, 0, 1], np.int16), 2)):
        for k in (5, 192, 1024):
            for sel in (list(range(len(queries))), short):                    # mixed batch / all-short batch
                qi2, qt2 = queries_to_csr([idx.tokens_to_ids(queries[b]) for b in sel])
                f2 = None if filt is None else filt[sel]
                ids, sc, ln = engine.bm25_topk(qi2, qt2, k, filter_dir=f2)
                for r, b in enumerate(sel):
                    mask = None if (filt is None or filt[b] < 0) else dir_id == filt[b]
                    want = bm25_filter(osc[b], k, mask)
                    assert ln[r] == len(want)
                    assert list(ids[r, :ln[r]]) == [w[0] for w in want], f"k={k} query {b}: ids differ"
                    assert list(sc[r, :ln[r]]) == [w[1] for w in want], f"k={k} query {b}: scores differ"


def test_bm25_two_indices_on_one_handle(engine):
    """Index slots: two BM25 indices over the same documents live side by side on one handle and answer
    independently (content route + know_path route of the reference pipeline, pipeline.py:187-210)."""
    flat_a, lens_a = synth.token_corpus(5000, 300, seed=31, mean_len=20)
    flat_b, lens_b = synth.token_corpus(5000, 40, seed=32, mean_len=3)
    docs_a = [list(map(int, d)) for d in synth.split_docs(flat_a, lens_a)]
    docs_b = [list(map(int, d)) for d in synth.split_docs(flat_b, lens_b)]
    ia, ib = build_bm25_index(docs_a, OKAPI), build_bm25_index(docs_b, BM25S)
    oa, ob = _oracle_for(OKAPI, docs_a), _oracle_for(BM25S, docs_b)
    engine.set_bm25(ia, slot=1)
    engine.set_bm25(ib, slot=2)
    try:
        qa = [list(map(int, q)) for q in synth.token_queries(flat_a, lens_a, 300, 6, seed=1)]
        qb = [list(map(int, q)) for q in synth.token_queries(flat_b, lens_b, 40, 6, seed=2, from_doc=2, random_extra=1)]
        for rep in range(2):                                                  # alternate between the slots
            for slot, idx, ora, qs, var, k in ((1, ia, oa, qa, OKAPI, 192), (2, ib, ob, qb, BM25S, 6)):
                qi, qt = queries_to_csr([idx.tokens_to_ids(q) for q in qs])
                ids, sc, ln = engine.bm25_topk(qi, qt, k, slot=slot)
                for b, q in enumerate(qs):
                    want = bm25_filter(_oracle_scores(ora, var, q), k)
                    assert list(ids[b, :ln[b]]) == [w[0] for w in want] and list(sc[b, :ln[b]]) == [w[1] for w in want]
                assert np.array_equal(engine.bm25_scores(idx.tokens_to_ids(qs[0]), slot=slot),
                                      _oracle_scores(ora, var, qs[0]).astype(np.float64))
    finally:
        engine._select(0)


@pytest.mark.parametrize("variant", [OKAPI, BM25S])
@pytest.mark.parametrize("n_docs,vocab,seed,mean_len", [(50, 12, 0, 6), (4000, 700, 1, 25), (60000, 5000, 2, 40)])
def test_device_index_build_is_bit_identical_to_host_builder(engine, variant, n_docs, vocab, seed, mean_len):
    """erh_build_bm25_index (sort / run-length / df on the GPU, idf on the host with libm's log) against the numpy
    builder that is itself checked against the rank_bm25 / bm25s restatements (tests/test_index_vs_oracle.py): CSR,
    tf, idf, avgdl, average_idf and the per-posting payload, bit for bit; and the built index answers queries."""
    flat, lens = synth.token_corpus(n_docs, vocab, seed=seed, mean_len=mean_len)
    if n_docs == 4000:
        lens = lens.copy()
        empty = lens[7]
        flat = np.concatenate([flat[:lens[:7].sum()], flat[lens[:8].sum():]])     # document 7 becomes empty
        lens[7] = 0
        assert empty > 0
    want = build_bm25_index_from_ids(flat=flat, doc_lens=lens, n_vocab=vocab + 3, variant=variant)   # ids vocab..vocab+2 never occur
    got = engine.build_bm25(flat, lens, vocab + 3, variant=variant)
    assert got.nnz == want.nnz
    assert np.array_equal(got.indptr, want.indptr) and np.array_equal(got.doc_ids, want.doc_ids)
    assert np.array_equal(got.tf, want.tf)
    assert got.idf.dtype == want.idf.dtype and np.array_equal(got.idf, want.idf)
    assert got.avgdl == want.avgdl and got.average_idf == want.average_idf
    assert got.payload.dtype == want.payload.dtype and np.array_equal(got.payload, want.payload)
    queries = [list(map(int, q)) for q in synth.token_queries(flat, lens, vocab, 6, seed=seed + 3)]
    qi, qt = queries_to_csr(queries)
    ids, sc, ln = engine.bm25_topk(qi, qt, 20)
    for b, q in enumerate(queries):
        acc = np.zeros(n_docs, want.payload.dtype)
        for t in q:
            s, e = want.indptr[t], want.indptr[t + 1]
            np.add.at(acc, want.doc_ids[s:e], want.payload[s:e])
        w = bm25_filter(acc, 20)
        assert list(ids[b, :ln[b]]) == [x[0] for x in w] and list(sc[b, :ln[b]]) == [x[1] for x in w]
    with pytest.raises(Exception):
        engine.build_bm25(np.array([0, vocab + 99], np.int32), np.array([2], np.int32), vocab, variant=variant)


def _random_lists(rng, n_items, la, lb, dup_rate):
    ids_a = rng.choice(n_items, size=la, replace=False)
    ids_b = rng.choice(n_items, size=lb, replace=False)
    cid = np.arange(n_items)
    dup = rng.random(n_items) < dup_rate
    cid[dup] = rng.integers(0, n_items, size=int(dup.sum()))
    cid = np.minimum(cid, np.arange(n_items))             # content id = some earlier-or-equal index
    return ids_a, ids_b, cid.astype(np.int32)


def test_rrf_and_fusion_match_oracle(engine):
    rng = np.random.default_rng(12)
    n_items, B = 5000, 40
    depth_a, depth_b = 192, 288
    _, _, cid = _random_lists(rng, n_items, 1, 1, 0.05)
    engine.set_doc_meta(n_items, cid, None)
    ids_a = np.full((B, depth_a), -1, np.int32)
    ids_b = np.full((B, depth_b), -1, np.int32)
    sc_a = np.zeros((B, depth_a))
    sc_b = np.zeros((B, depth_b))
    la = rng.integers(0, depth_a + 1, size=B).astype(np.int32)
    lb = rng.integers(0, depth_b + 1, size=B).astype(np.int32)
    la[0], lb[0] = 0, 0
    la[1], lb[1] = 0, 7
    la[2], lb[2] = depth_a, depth_b
    for b in range(B):
        ids_a[b, :la[b]] = rng.choice(n_items, size=la[b], replace=False)
        # overlap the two routes heavily, as real retrieval does
        pool = np.concatenate([ids_a[b, :la[b]], rng.choice(n_items, size=depth_b, replace=False)])
        ids_b[b, :lb[b]] = rng.permutation(np.unique(pool))[:lb[b]]
        sc_a[b, :la[b]] = np.sort(rng.integers(1, 30, size=la[b]))[::-1]          # integer scores -> many ties
        sc_b[b, :lb[b]] = np.sort(rng.random(lb[b]))[::-1]
    for topk in (10, 256):
        rid, rsc, rln = engine.rrf(ids_a, la, ids_b, lb, K=60, topk=topk)
        fid, fsc, fln = engine.fusion(ids_a, sc_a, la, ids_b, sc_b, lb, topk=topk)
        for b in range(B):
            A = [Item(int(i), int(cid[i]), float(s)) for i, s in zip(ids_a[b, :la[b]], sc_a[b, :la[b]])]
            Bl = [Item(int(i), int(cid[i]), float(s)) for i, s in zip(ids_b[b, :lb[b]], sc_b[b, :lb[b]])]
            want = reciprocal_rank_fusion([A, Bl], K=60, topk=topk)
            assert rln[b] == len(want)
            assert list(rid[b, :rln[b]]) == [w.idx for w in want]
            assert list(rsc[b, :rln[b]]) == [w.score for w in want]
            wantf = fusion([A, Bl], topk=topk)
            assert fln[b] == len(wantf)
            assert list(fid[b, :fln[b]]) == [w.idx for w in wantf]
            assert list(fsc[b, :fln[b]]) == [w.score for w in wantf]


@pytest.mark.parametrize("variant", [OKAPI, BM25S])
def test_hybrid_dual_route_matches_oracle_composition(engine, variant):
    """BASELINE.json configs[0] at its own size: 10k chunks x 768-d, 100 queries, dense(288) + BM25(192) + RRF(60) -> top-10
    (and top-256, the yaml default), every query against the oracle composition; 1 % duplicated contents."""
    n, d, vocab, B = 10000, 768, 4096, 100
    x = synth.dense_corpus(n, d, seed=1)
    q16 = to_f16_unit(synth.dense_queries(x, B, seed=2))
    flat, lens = synth.token_corpus(n, vocab, seed=3)
    docs = [list(map(int, t)) for t in synth.split_docs(flat, lens)]
    idx = build_bm25_index(docs, variant)
    ora = _oracle_for(variant, docs)
    queries = [list(map(int, q)) for q in synth.token_queries(flat, lens, vocab, B, seed=4)]
    # 1 % duplicated contents (SURVEY.md 8(d) config 4 variant)
    rng = np.random.default_rng(6)
    cid = np.arange(n, dtype=np.int32)
    dup = rng.choice(n, size=n // 100, replace=False)
    cid[dup] = np.minimum(cid[dup], rng.integers(0, n, size=dup.shape[0]))
    engine.set_dense(x)
    engine.set_bm25(idx)
    engine.set_doc_meta(n, cid, None)
    qi, qt = queries_to_csr([idx.tokens_to_ids(q) for q in queries])
    o_sparse = [bm25_filter(_oracle_scores(ora, variant, queries[b]), 192) for b in range(B)]
    o_dense = [dense_exact_topk(x, q16[b], 288) for b in range(B)]
    for topk, overlap in ((10, 0), (256, 0), (10, 1), (10, 2)):      # (hybrid_overlap: the sparse route on a side stream)
        engine.set_option("hybrid_overlap", overlap)
        try:
            ids, sc, ln = engine.hybrid_topk(q16, qi, qt, k_dense=288, k_sparse=192, K=60, topk=topk)
        finally:
            engine.set_option("hybrid_overlap", -1)
        for b in range(B):
            sp = o_sparse[b]
            did, dsc = o_dense[b]
            A = [Item(i, int(cid[i]), s) for i, s in sp]
            Bl = [Item(int(i), int(cid[i]), float(s)) for i, s in zip(did, dsc)]
            want = reciprocal_rank_fusion([A, Bl], K=60, topk=topk)
            assert ln[b] == len(want)
            assert list(ids[b, :ln[b]]) == [w.idx for w in want], f"query {b}: fused ids differ"
            assert list(sc[b, :ln[b]]) == [w.score for w in want]


def test_bm25_full_size_properties(engine):
    """Config 3 shape: 1M documents, ~50M postings, B = 256, bm25s payloads.  Exact comparison against a numpy
    scatter-add for a sample of queries plus size-independent properties for all of them."""
    import torch
    dev = torch.device("cuda", 0)
    n, vocab, B, k = 1_000_000, 262_144, 256, 100
    indptr, doc, tf, lens, flat = synth.token_csr_torch(n, vocab, seed=3, device=dev)
    from easyrag_amd.index import build_bm25_index_from_postings
    idx = build_bm25_index_from_postings(indptr, doc, tf, lens, BM25S)
    assert 30_000_000 < idx.nnz < 70_000_000
    engine.set_bm25(idx)
    queries = synth.token_queries(flat, lens, vocab, B, seed=9)
    qi, qt = queries_to_csr(queries)
    ids, sc, ln = engine.bm25_topk(qi, qt, k)
    assert np.all(ln <= k) and np.all(ln > 0)
    for b in range(B):
        m = ln[b]
        assert np.all(sc[b, :m] > 0) and np.all(np.diff(sc[b, :m]) <= 0)
        assert len(set(ids[b, :m])) == m
        tie = np.diff(sc[b, :m]) == 0
        assert np.all(np.diff(ids[b, :m])[tie] > 0)                  # equal scores ordered by index
    for b in (0, 1, 77, 255):
        acc = np.zeros(n, np.float32)
        for t in queries[b]:
            s, e = idx.indptr[t], idx.indptr[t + 1]
            np.add.at(acc, idx.doc_ids[s:e], idx.payload[s:e])
        want = bm25_filter(acc, k)
        assert list(ids[b, :ln[b]]) == [w[0] for w in want] and list(sc[b, :ln[b]]) == [w[1] for w in want]


@pytest.mark.parametrize("variant", [OKAPI, BM25S])
def test_bm25_dir_filter_as_tile_range(engine, bm25_kernel, variant):
    """Round 5: a dir filter becomes a tile range.  The reference's `dir` is the first path component and its loader walks the
    directories one after the other (ref transformation.py:70, ingestion.py:79-87), so a dir is one block of consecutive documents:
    the fixed-point scan walks only the tiles of the filtered class.  Blocks of unequal size that cut through tiles, a class with a
    single document, one that is split in two (its range then spans the block between), a class nobody carries and one beyond the
    table, queries without a filter in the same batch; few queries (document-range segments + merge) and many.  Same lists as the
    oracle's masked walk, and as the scan over all tiles (bm25_dir_range = 0)."""
    n_docs, vocab = 90000, 3000
    flat, lens = synth.token_corpus(n_docs, vocab, seed=17, mean_len=14)
    docs = _cached(("tile_range", "docs"), lambda: [list(map(int, d)) for d in synth.split_docs(flat, lens)])
    ora = _cached(("tile_range", "ora", variant), lambda: _oracle_for(variant, docs))
    idx = _cached(("tile_range", "idx", variant), lambda: build_bm25_index(docs, variant))
    engine.set_bm25(idx)
    dir_id = np.zeros(n_docs, np.int16)
    dir_id[20000:52000] = 1                       # ends inside a tile
    dir_id[52000:52001] = 2                       # one document
    dir_id[52001:70000] = 3
    dir_id[70000:] = 5                            # (class 4: nobody)
    dir_id[100:200] = 3                           # class 3 is split: its range spans classes 0 .. 2
    engine.set_doc_meta(n_docs, None, dir_id)
    try:
        for B in (5, 40):
            queries = [list(map(int, q)) for q in synth.token_queries(flat, lens, vocab, B, seed=31 + B)]
            filt = np.array([(-1, 0, 1, 2, 3, 5, 4, 9)[b % 8] for b in range(B)], np.int16)
            qi, qt = queries_to_csr([idx.tokens_to_ids(q) for q in queries])
            got = {}
            for rng_on in (1, 0):
                engine.set_option("bm25_dir_range", rng_on)
                got[rng_on] = engine.bm25_topk(qi, qt, 50, filter_dir=filt)
            engine.set_option("bm25_dir_range", 1)
            for a, b in zip(got[1], got[0]):
                assert np.array_equal(a, b)
            ids, sc, ln = got[1]
            for b, q in enumerate(queries):
                mask = None if filt[b] < 0 else dir_id == filt[b]
                want = _cached(("tile_range", "want", variant, B, b), lambda: bm25_filter(_oracle_scores(ora, variant, q), 50, mask))
                assert ln[b] == len(want), (B, b)
                assert list(ids[b, :ln[b]]) == [w[0] for w in want] and list(sc[b, :ln[b]]) == [w[1] for w in want], (B, b)
                if filt[b] in (4, 9):
                    assert ln[b] == 0
    finally:
        engine.set_option("bm25_dir_range", 1)
        engine.set_doc_meta(n_docs, None, None)


@pytest.mark.parametrize("variant", [BM25S, OKAPI], ids=["bm25s", "okapi"])
def test_bm25_reference_question_lengths(engine, bm25_kernel, variant):
    """Queries as long as the reference's real questions (synth.REF_QUESTION_LENGTHS: 4 ... 45 tokens, tokens repeated when the target
    document is short) in ONE batch: in a batch whose longest query exceeds bm25_long_tokens the long queries' workgroups run the 32-bit
    body and the others the packed one in ONE launch (bm25_mixed; with it off the whole batch leaves the packed 16-bit shape for the 32-bit
    one), and with bm25_long_tokens at 0 -- always packed -- or 8 the lists are the same; every arm, with and without a dir filter, against
    the oracle -- ids and scores bit for bit."""
    rng = np.random.default_rng(88)
    n, vocab, b, k = 30000, 3000, 103, 60
    flat, lens = synth.token_corpus(n, vocab, seed=31, mean_len=30)
    docs = _cached(("ref_lengths", "docs"), lambda: [list(map(int, t)) for t in synth.split_docs(flat, lens)])
    idx = _cached(("ref_lengths", "idx", variant), lambda: build_bm25_index(docs, variant))
    engine.set_bm25(idx)
    dir_id = np.repeat(np.arange(3), [12000, 10000, 8000]).astype(np.int16)
    engine.set_doc_meta(n, None, dir_id)
    ora = _cached(("ref_lengths", "ora", variant), lambda: _oracle_for(variant, docs))
    qs = synth.token_queries(flat, lens, vocab, b, seed=32, lengths=sorted(synth.REF_QUESTION_LENGTHS))
    for i, L in enumerate(sorted(synth.REF_QUESTION_LENGTHS)):       # every length of the reference's questions, the 45-token one included
        tgt = docs[int(rng.integers(0, n))]
        qs[i] = np.asarray([tgt[int(j)] for j in rng.integers(0, len(tgt), L)], np.int32)
    filt = (np.arange(b) % 4 - 1).astype(np.int16)                   # none, 0, 1, 2
    try:
        for f in (None, filt):
            got = {}
            # the one-launch-two-bodies path exists for the packed shape on 4-byte postings with the tail inside the scan kernel
            can_mix = bm25_kernel[0] == 1 and bm25_kernel[3] == 2 and bm25_kernel[4] == 1 and bm25_kernel[5] == 0
            for long_tokens, mixed in ((28, 1), (28, 0), (0, 1), (8, 1), (8, 0)):
                engine.set_option("bm25_long_tokens", long_tokens)
                engine.set_option("bm25_mixed", mixed)
                engine.reset_stats()
                got[long_tokens, mixed] = engine.bm25_topk(*queries_to_csr([idx.tokens_to_ids(list(map(int, q))) for q in qs]), k, filter_dir=f)
                assert engine.stat("bm25_mixed_launches") == (1 if can_mix and mixed and long_tokens else 0), (long_tokens, mixed)
            ids, sc, ln = got[28, 1]
            for other in ((28, 0), (0, 1), (8, 1), (8, 0)):
                assert np.array_equal(ln, got[other][2]) and np.array_equal(ids, got[other][0]), other
                assert np.array_equal(sc.view(np.uint64), got[other][1].view(np.uint64)), other
            for i in range(b):
                mask = None if f is None or f[i] < 0 else dir_id == f[i]
                want = _cached(("ref_lengths", "want", variant, f is None, i), lambda: bm25_filter(ora.get_scores(list(map(int, qs[i]))), k, mask))
                assert list(ids[i, :ln[i]]) == [w[0] for w in want], (i, len(qs[i]))
                assert list(sc[i, :ln[i]]) == [w[1] for w in want], (i, len(qs[i]))
    finally:
        engine.set_option("bm25_long_tokens", 28)
        engine.set_option("bm25_mixed", 1)
        engine.set_doc_meta(n, None, None)
