"""GPU parity, dense route: HIP path (through the C ABI) vs the oracle on the same seeded inputs.

Bars (BASELINE.json north_star): fp32 MFMA cosine within 1e-3 of the reference-style fp32 scores; ranking
ids and fp64 scores bit-identical to oracle.dense_exact_topk (pinned summation order, canonical ties).
"""
import numpy as np
import pytest

from easyrag_amd import _lib, synth
from oracle import dense_exact_scores, dense_exact_topk, qdrant_cosine_search, to_f16_unit

pytestmark = pytest.mark.gpu


# (dense_cfg, dense_persist, dense_pp, dense_gemv, dense_speculate, dense_tiled)
@pytest.fixture(params=[(0, 1, 3, 0, 1, 0), (0, 1, 3, 0, 1, 1), (0, 1, 3, 0, 0, 0), (0, 0, 0, 0, 1, 0), (0, 0, 0, 0, 0, 0),
                        (1, 0, 0, 0, 1, 0), (2, 0, 0, 0, 0, 0), (0, 1, 3, 1, 1, 1)],
                ids=["pingpong-strict-rowmajor-256x256x32", "pingpong-strict-tiled-256x256x32", "pingpong-strict-rowmajor-guaranteed-bounds",
                     "cfg0-256x256x64-per-tile", "cfg0-per-tile-guaranteed-bounds", "cfg1-128x256x32-per-tile",
                     "cfg2-256x256x32-per-tile-guaranteed-bounds", "gemv-16x16x32-up-to-64-queries"])
def scan_cfg(request, engine):
    """Every dense-scan kernel of the product library (the strict ping-pong scan, the per-tile fallbacks in their three tile
    configurations, the skinny-GEMM stream) / chunk layout must satisfy every parity test, with
    the speculative (verified) first threshold and with guaranteed bounds refined in stages.  The last arm lets batches of
    at most 64 queries take the skinny-GEMM stream in 1 / 2 / 4 column groups of 16 (larger batches use the ping-pong scan); the other arms pin the padded
    256-query scans for every batch size.  (dense_tiled takes effect at the next set_dense: every test sets its own.)"""
    engine.set_option("dense_cfg", request.param[0])
    engine.set_option("dense_persist", request.param[1])
    engine.set_option("dense_pp", request.param[2])
    engine.set_option("dense_gemv", request.param[3])
    engine.set_option("dense_speculate", request.param[4])
    engine.set_option("dense_tiled", request.param[5])
    yield request.param
    engine.set_option("dense_cfg", 0)
    engine.set_option("dense_persist", 1)
    engine.set_option("dense_pp", 3)
    engine.set_option("dense_gemv", 1)
    engine.set_option("dense_speculate", 1)
    engine.set_option("dense_tiled", 0)


def test_mfma_scores_match_plain_gpu_and_numpy(engine, scan_cfg):
    # asymmetric operands: a swapped C/D layout or a wrong swizzle cannot pass
    rng = np.random.default_rng(11)
    n, d, b = 1000, 128, 70
    x = to_f16_unit(rng.standard_normal((n, d)) * np.linspace(0.2, 3.0, d))
    q = to_f16_unit(rng.standard_normal((b, d)) + 0.5)
    engine.set_dense(x)
    ref = x.astype(np.float32) @ q.astype(np.float32).T          # [n, b]
    for row0, rows in ((0, 1000), (37, 300), (999, 1)):
        naive = engine.debug_dense_scores(q, row0, rows, use_mfma=False)
        mfma = engine.debug_dense_scores(q, row0, rows, use_mfma=True)
        want = ref[row0:row0 + rows].T
        assert np.max(np.abs(naive - want)) < 1e-4
        assert np.max(np.abs(mfma - want)) < 1e-4
        assert np.max(np.abs(mfma - naive)) < 1e-4


CASES = [
    # n, d, B, k, n0, n1
    (300, 64, 3, 10, 32768, 262144),        # everything inside the seed prefix, N < k*...
    (5000, 256, 17, 100, 1024, 3072),       # seed -> append -> refine -> append
    (5000, 768, 40, 288, 512, 0),           # no refine stage
    (20000, 1024, 300, 10, 2048, 8192),     # B > one query tile
    (20000, 256, 33, 50, 256, 1024),        # three append stages: boundaries 1024 and 4096 (x4 while 8x fits)
    (60000, 256, 9, 100, 256, 32768),       # loose seed: ~12k candidates per query reach the refinement (full-capacity launch)
    (257, 64, 1, 288, 32768, 0),            # k > N
    (30000, 768, 16, 100, 1024, 4096),      # small batches: the skinny-GEMM stream in the gemv arm (16 = one column group)
    (30000, 1024, 64, 100, 1024, 4096),     # ... four column groups, 128 KiB of query fragments (its widest)
    (30000, 1024, 48, 288, 2048, 0),        # ... three of four groups populated
    (30000, 512, 32, 100, 1024, 0),         # ... two column groups
    (30000, 1024, 100, 100, 1024, 4096),    # 65 ... 128 queries: the ping-pong scan computes half of its query tile (every arm but
    (20000, 512, 128, 288, 2048, 0),        #   the per-tile ones; with dense_gemv = 0 every batch of <= 128 queries takes that mode)
    (20000, 256, 200, 50, 512, 0),          # 129 ... 256 queries: one full query tile
    (12000, 1024, 5, 288, 512, 0),
    (9000, 192, 2, 20, 256, 0),             # d = 6 steps of 32
    (40000, 1280, 1, 50, 2048, 8192),       # one query, d > 1024: two blocks of K
    # d = 3584: the reference's own vector_size (ref:src/configs/easyrag.yaml:16, gte-Qwen2-7B) -- 112 K-stages, the finalize
    # kernel's d > 2048 tail straight from memory, and in the gemv arm the fallback of 17 ... 64 queries to the padded scan
    # (2 / 4 column groups x 3584 x 32 bytes of query fragments exceed the 128 KiB the skinny-GEMM stream keeps in LDS)
    (4000, 3584, 1, 288, 1024, 0),          # one query: the skinny-GEMM stream at 112 KiB of query fragments
    (4000, 3584, 16, 100, 1024, 0),         # one full column group
    (4000, 3584, 17, 100, 1024, 0),         # gemv arm: falls back to the half-tile ping-pong scan
    (4000, 3584, 64, 100, 512, 2048),       # ... likewise, with a refinement boundary
    (4000, 3584, 100, 288, 1024, 0),        # half-tile mode of the ping-pong scan
    (4000, 3584, 256, 100, 1024, 0),        # one full query tile
    (4000, 3584, 600, 10, 768, 0),          # three query tiles; seed prefix = 2 x 384 rows: the 384 x 256 kernel in the ping-pong arms
]


_ORACLE_CACHE = {}


def _oracle_cached(kind, case, i, fn):
    """The oracle's answer for query i of a CASES entry: the inputs are a pure function of the entry (seeded generators), so
    the eight kernel arms share one oracle evaluation (at d = 3584 it is 0.4 s per query on the host)."""
    key = (kind, case, i)
    if key not in _ORACLE_CACHE:
        _ORACLE_CACHE[key] = fn()
    return _ORACLE_CACHE[key]


@pytest.mark.parametrize("n,d,b,k,n0,n1", CASES)
def test_dense_topk_exact_matches_oracle(engine, scan_cfg, n, d, b, k, n0, n1):
    case = (n, d, b, k)
    x = synth.dense_corpus(n, d, seed=n + d)
    q32 = synth.dense_queries(x, b, seed=b + k)
    q16 = to_f16_unit(q32)
    engine.set_dense(x)
    engine.set_option("dense_n0", n0)
    engine.set_option("dense_n1", n1)
    try:
        ids, sc, ln = engine.dense_topk(q16, k, mode=_lib.ERH_DENSE_EXACT)
        fids, fsc, fln = engine.dense_topk(q16, k, mode=_lib.ERH_DENSE_FAST)
    finally:
        engine.set_option("dense_n0", 32768)
        engine.set_option("dense_n1", 131072)
    diag = engine.dense_diag()
    assert diag["uncertified"] == 0 and diag["max_abs_err"] <= diag["margin"]
    kk = min(k, n)
    assert np.all(ln == kk) and np.all(fln == kk)
    check = sorted(set(list(range(min(b, 6))) + list(range(0, b, max(1, b // 10))) + [b - 1, min(b - 1, 256)]))
    for i in check:                                   # the oracle is O(n*d) fp64 per query: sample large batches
        oid, osc = _oracle_cached("exact", case, i, lambda: dense_exact_topk(x, q16[i], k))
        assert ln[i] == kk and fln[i] == kk
        assert np.array_equal(ids[i, :kk], oid), f"query {i}: ids differ"
        assert np.array_equal(sc[i, :kk].view(np.uint64), osc.view(np.uint64)), f"query {i}: fp64 scores differ"
        assert np.all(ids[i, kk:] == -1)
        # fp32 MFMA ranking: scores within 1e-3 (in practice ~1e-6) of the exact ones
        exact_of_fast = dense_exact_scores(x, q16[i], rows=fids[i, :kk])
        assert np.max(np.abs(fsc[i, :kk] - exact_of_fast)) < 1e-3
        assert np.all(np.diff(fsc[i, :kk]) <= 0)
    # against the qdrant-local style fp32 search (reference semantics): cosine within 1e-3, same id set
    for i in range(min(b, 4)):
        qi, qs = _oracle_cached("qdrant", case, i, lambda: qdrant_cosine_search(x.astype(np.float32), q32[i], k))
        assert np.max(np.abs(np.sort(qs)[::-1] - sc[i, :kk])) < 1e-3
        assert len(set(qi) & set(ids[i, :kk])) >= kk - 2            # only near-ties at the cut may differ


def test_dense_duplicates_and_filter(engine, scan_cfg):
    rng = np.random.default_rng(3)
    base = to_f16_unit(rng.standard_normal((50, 128)))
    x = np.repeat(base, 8, axis=0)                        # every chunk 8 times: exact ties everywhere
    n = x.shape[0]
    q16 = base[[5, 17, 33]]
    dir_id = (np.arange(n) % 3).astype(np.int16)
    engine.set_dense(x)
    engine.set_doc_meta(n, None, dir_id)
    ids, sc, ln = engine.dense_topk(q16, 12)
    for i in range(3):
        oid, osc = dense_exact_topk(x, q16[i], 12)
        assert np.array_equal(ids[i], oid) and np.array_equal(sc[i], osc)
    filt = np.array([0, 2, -1], np.int16)
    ids, sc, ln = engine.dense_topk(q16, 12, filter_dir=filt)
    for i in range(3):
        mask = None if filt[i] < 0 else (dir_id == filt[i])
        oid, osc = dense_exact_topk(x, q16[i], 12, mask)
        assert np.array_equal(ids[i, :ln[i]], oid) and np.array_equal(sc[i, :ln[i]], osc)


def test_dense_tie_block_floods_the_survivor_buffers(engine, scan_cfg):
    """600 chunks equal to the (identical) queries sit in three consecutive tiles behind the seed prefix: every
    (chunk, query) pair of those tiles survives the threshold, i.e. thousands of hits per wave and tile -- far
    more than a wave's staging area holds.  The result must still be the 100 lowest indices of the tie block."""
    rng = np.random.default_rng(21)
    n, d, b, k = 6000, 256, 70, 100
    x = to_f16_unit(rng.standard_normal((n, d)))
    hot = to_f16_unit(rng.standard_normal((1, d)))[0]
    x[1500:2100] = hot
    q16 = np.repeat(hot[None, :], b, axis=0)
    engine.set_dense(x)
    engine.set_option("dense_n0", 512)
    engine.set_option("dense_n1", 1024)
    try:
        ids, sc, ln = engine.dense_topk(q16, k)
    finally:
        engine.set_option("dense_n0", 32768)
        engine.set_option("dense_n1", 131072)
    diag = engine.dense_diag()
    assert diag["uncertified"] == 0
    oid, osc = dense_exact_topk(x, hot, k)
    assert np.array_equal(oid, np.arange(1500, 1600))
    for i in (0, 1, 33, b - 1):
        assert ln[i] == k
        assert np.array_equal(ids[i], oid)
        assert np.array_equal(sc[i].view(np.uint64), osc.view(np.uint64))


def test_dense_seed_full_sort_fallback(engine):
    """A filter that leaves valid scores in only 32 of the 1024 per-thread strides of the seed row: the thread-maxima
    pivot has fewer than k entries, so the query goes through the full-sort seed kernel."""
    rng = np.random.default_rng(17)
    n, d, b, k = 6000, 128, 5, 50
    x = to_f16_unit(rng.standard_normal((n, d)))
    q16 = to_f16_unit(rng.standard_normal((b, d)))
    dir_id = np.where(np.arange(n) % 1024 < 32, 1, 0).astype(np.int16)
    engine.set_option("dense_shuffle", 0)            # the stride pattern is about stored positions
    engine.set_option("dense_n0", 4096)
    engine.set_option("dense_n1", 0)
    try:
        engine.set_dense(x)
        engine.set_doc_meta(n, None, dir_id)
        filt = np.array([1, 1, -1, 0, 1], np.int16)
        ids, sc, ln = engine.dense_topk(q16, k, filter_dir=filt)
        for i in range(b):
            mask = None if filt[i] < 0 else dir_id == filt[i]
            oid, osc = dense_exact_topk(x, q16[i], k, mask)
            assert ln[i] == len(oid)
            assert np.array_equal(ids[i, :ln[i]], oid) and np.array_equal(sc[i, :ln[i]].view(np.uint64), osc.view(np.uint64))
    finally:
        engine.set_option("dense_shuffle", 1)
        engine.set_option("dense_n0", 32768)
        engine.set_option("dense_n1", 131072)


@pytest.mark.parametrize("b,k", [(3, 40), (70, 100), (256, 288)])
def test_dense_seed_fallback_on_a_tie_cluster_in_the_prefix(engine, b, k):
    """6000 chunks equal to the query INSIDE the seed prefix: more than 4096 prefix scores lie above the thread-maxima pivot, the
    gather buffer of the seed select overflows and the kernel takes its own fall-back (round 6: a radix select of the rank-th largest
    key over the row, where a second full-sort kernel used to be launched behind every seed select).  Speculative and guaranteed
    first thresholds, the skinny-GEMM stream and the 256-query tile, with and without a filter that halves the cluster: the lowest
    indices of the tie block come back, ids and fp64 scores equal to the oracle."""
    rng = np.random.default_rng(61)
    n, d = 40000, 128
    x = to_f16_unit(rng.standard_normal((n, d)))
    hot = to_f16_unit(rng.standard_normal((1, d)))[0]
    x[1000:7000] = hot
    q16 = to_f16_unit(np.repeat(hot[None, :].astype(np.float32), b, axis=0) + (np.arange(b)[:, None] % 2) * 0.02 * rng.standard_normal((b, d)))
    dir_id = (np.arange(n) % 2).astype(np.int16)
    engine.set_option("dense_shuffle", 0)            # the cluster stays where it was put: inside the first 16384 stored rows
    engine.set_option("dense_n0", 16384)
    try:
        engine.set_dense(x)
        engine.set_doc_meta(n, None, dir_id)
        for spec in (1, 0):
            engine.set_option("dense_speculate", spec)
            for filt in (None, (np.arange(b) % 3 - 1).astype(np.int16)):
                ids, sc, ln = engine.dense_topk(q16, k, filter_dir=filt)
                for i in sorted(set([0, 1, b // 2, b - 1])):
                    mask = None if filt is None or filt[i] < 0 else dir_id == filt[i]
                    oid, osc = dense_exact_topk(x, q16[i], k, mask)
                    assert ln[i] == len(oid), (spec, i)
                    assert np.array_equal(ids[i, :ln[i]], oid), (spec, i)
                    assert np.array_equal(sc[i, :ln[i]].view(np.uint64), osc.view(np.uint64)), (spec, i)
    finally:
        engine.set_option("dense_speculate", 1)
        engine.set_option("dense_shuffle", 1)
        engine.set_option("dense_n0", 32768)
        engine.set_doc_meta(n, None, None)


def test_dense_corpus_sorted_by_topic(engine):
    """A corpus ordered by topic: the queries' topic fills the LAST 60 % of the rows, so a threshold seeded from the
    first rows of the caller's order would admit tens of thousands of candidates per query (more than the candidate
    lists hold).  Rows are stored in a golden-ratio placement, which makes every stored prefix an even sample of
    the caller's order: the result is exact and no list overflows; with dense_shuffle=0 the candidate lists do
    overflow, and every query (12 of them) is then answered by the exhaustive path -- same ids, same scores."""
    rng = np.random.default_rng(31)
    n, d, b, k = 60000, 256, 12, 50
    topic_a = rng.standard_normal(d)
    topic_b = rng.standard_normal(d)
    x32 = rng.standard_normal((n, d)) * 0.35
    x32[: n * 2 // 5] += topic_a
    x32[n * 2 // 5:] += topic_b
    x = to_f16_unit(x32)
    q16 = to_f16_unit(topic_b + 0.35 * rng.standard_normal((b, d)))
    engine.set_option("dense_n0", 512)
    engine.set_option("dense_n1", 2048)
    try:
        engine.set_dense(x)
        ids, sc, ln = engine.dense_topk(q16, k)
        diag = engine.dense_diag()
        assert diag["uncertified"] == 0
        for i in (0, 5, b - 1):
            oid, osc = dense_exact_topk(x, q16[i], k)
            assert np.array_equal(ids[i], oid)
            assert np.array_equal(sc[i].view(np.uint64), osc.view(np.uint64))
        # filter push-down goes through the by-position dir table
        dir_id = (np.arange(n) % 5).astype(np.int16)
        engine.set_doc_meta(n, None, dir_id)
        filt = np.full(b, 3, np.int16)
        ids, sc, ln = engine.dense_topk(q16, k, filter_dir=filt)
        oid, osc = dense_exact_topk(x, q16[2], k, dir_id == 3)
        assert np.array_equal(ids[2], oid) and np.array_equal(sc[2].view(np.uint64), osc.view(np.uint64))
        assert engine.dense_diag()["exhaustive"] == 0
        engine.set_option("dense_shuffle", 0)
        engine.set_dense(x)
        engine.set_doc_meta(n, None, dir_id)
        for f, mask in ((None, None), (filt, dir_id == 3)):
            ids, sc, ln = engine.dense_topk(q16, k, filter_dir=f)
            if f is None:                                          # (the filter keeps a fifth of the candidates: they fit)
                assert engine.dense_diag()["exhaustive"] == b      # the reference always answers; so does this
            for i in (0, 5, b - 1):
                oid, osc = dense_exact_topk(x, q16[i], k, mask)
                assert np.array_equal(ids[i, :ln[i]], oid)
                assert np.array_equal(sc[i, :ln[i]].view(np.uint64), osc.view(np.uint64))
    finally:
        engine.set_option("dense_shuffle", 1)
        engine.set_option("dense_n0", 32768)
        engine.set_option("dense_n1", 131072)


def test_dense_speculation_failure_is_caught(engine):
    """The first pruning threshold is speculative: the rank-r score of the stored prefix (r << k) estimates where the
    corpus' k-th best lies, on the premise that the prefix is an even sample of the corpus.  Here the premise is
    broken on purpose (dense_shuffle=0 and a caller order whose first rows are exactly what the queries ask for): the
    threshold lands far above the true k-th best, fewer than k chunks reach it, dense_finalize_kernel notices, and the
    exhaustive path answers every query -- same ids, same fp64 scores as the oracle.  With the row placement on, the
    same data needs no fallback at all."""
    rng = np.random.default_rng(77)
    n, d, b, k = 40000, 256, 6, 60
    topic = rng.standard_normal(d)
    x32 = rng.standard_normal((n, d))
    x32[:40] = topic + 0.2 * rng.standard_normal((40, d))          # 40 < k rows on topic, all inside the first n0 rows
    x = to_f16_unit(x32)
    q16 = to_f16_unit(topic + 0.2 * rng.standard_normal((b, d)))
    engine.set_option("dense_n0", 512)
    engine.set_option("dense_gemv", 0)
    try:
        for shuffle, want_exhaustive in ((0, b), (1, 0)):
            engine.set_option("dense_shuffle", shuffle)
            engine.set_dense(x)
            ids, sc, ln = engine.dense_topk(q16, k)
            assert engine.dense_diag()["exhaustive"] == want_exhaustive
            for i in range(b):
                oid, osc = dense_exact_topk(x, q16[i], k)
                assert np.array_equal(ids[i], oid)
                assert np.array_equal(sc[i].view(np.uint64), osc.view(np.uint64))
    finally:
        engine.set_option("dense_shuffle", 1)
        engine.set_option("dense_n0", 32768)
        engine.set_option("dense_gemv", 1)


@pytest.mark.parametrize("b,k", [(300, 100), (512, 100), (700, 288), (1024, 10), (1024, 288)])
def test_dense_sample_pass_matches_oracle(engine, b, k):
    """Round 4, batches padded to 512 queries and more: the ping-pong scan draws its own threshold sample -- a pass without thresholds over the first tile(s) of every
    chunk stream whose only output is the two best scores of every 64-row cell, a select over those, then the scan of ALL
    rows (no store kernel, no S0).  Same ids and fp64 scores as the oracle, and as the store-kernel / S0 / seed-select path
    (dense_selfseed = 0) bit for bit."""
    n, d = 200_000, 256
    x = synth.dense_corpus(n, d, seed=401)
    q16 = to_f16_unit(synth.dense_queries(x, b, seed=402 + b))
    engine.set_option("dense_gemv", 0)
    try:
        engine.set_dense(x)
        ids, sc, ln = engine.dense_topk(q16, k)
        diag = engine.dense_diag()
        assert diag["uncertified"] == 0 and diag["max_abs_err"] <= diag["margin"] and diag["exhaustive"] == 0
        engine.set_option("dense_selfseed", 0)
        ids0, sc0, ln0 = engine.dense_topk(q16, k)
    finally:
        engine.set_option("dense_selfseed", 1)
        engine.set_option("dense_gemv", 1)
    assert np.array_equal(ids, ids0) and np.array_equal(sc.view(np.uint64), sc0.view(np.uint64)) and np.array_equal(ln, ln0)
    assert np.all(ln == k)
    for i in sorted(set([0, 1, b // 3, b // 2, b - 2, b - 1, min(b - 1, 127), min(b - 1, 128)])):
        oid, osc = dense_exact_topk(x, q16[i], k)
        assert np.array_equal(ids[i], oid), f"query {i}: ids differ"
        assert np.array_equal(sc[i].view(np.uint64), osc.view(np.uint64)), f"query {i}: fp64 scores differ"


@pytest.mark.parametrize("b,k", [(1, 288), (7, 100), (24, 288), (40, 60), (100, 100), (256, 100)])
def test_dense_cache_policy_and_occupancy_arms(engine, b, k):
    """Round 6 arms that must never change a result: the skinny-GEMM stream's chunk loads with / without the non-temporal hint
    (dense_gemv_nt; default: by call), the 256 x 256 scan's chunk-side LDS-DMA with it (dense_scan_nt; incl. the half-tile variant at
    65 ... 128 queries), two / three workgroups of the final kernel per CU (dense_fin_wgs).  Every arm equals the default bit for bit,
    the default equals the oracle (ids + pinned-order fp64 scores)."""
    n, d = 150_000, 512
    x = synth.dense_corpus(n, d, seed=611)
    q16 = to_f16_unit(synth.dense_queries(x, b, seed=612 + b))
    engine.set_dense(x)
    try:
        ids, sc, ln = engine.dense_topk(q16, k)
        diag = engine.dense_diag()
        assert diag["uncertified"] == 0 and diag["max_abs_err"] <= diag["margin"] and diag["exhaustive"] == 0
        for name, value, reset in (("dense_gemv_nt", 0, -1), ("dense_gemv_nt", 1, -1), ("dense_scan_nt", 1, 0), ("dense_fin_wgs", 2, 4), ("dense_fin_wgs", 3, 4)):
            engine.set_option(name, value)
            try:
                ids1, sc1, ln1 = engine.dense_topk(q16, k)
            finally:
                engine.set_option(name, reset)
            assert np.array_equal(ids, ids1) and np.array_equal(sc.view(np.uint64), sc1.view(np.uint64)) and np.array_equal(ln, ln1), (name, value)
    finally:
        engine.set_option("dense_gemv_nt", -1)
        engine.set_option("dense_scan_nt", 0)
        engine.set_option("dense_fin_wgs", 4)
    assert np.all(ln == k)
    for i in sorted(set([0, b // 2, b - 1])):
        oid, osc = dense_exact_topk(x, q16[i], k)
        assert np.array_equal(ids[i], oid), f"query {i}: ids differ"
        assert np.array_equal(sc[i].view(np.uint64), osc.view(np.uint64)), f"query {i}: fp64 scores differ"


def test_dense_sample_pass_with_a_cell_full_of_copies(engine):
    """Eight copies of the chunk a query asks for, planted in ONE cell of the sampled rows (stored rows 0-3 and 8-11 of tile 0:
    one lane of one wave holds them; dense_shuffle = 0 keeps the caller's order).  The cell shows the sample only two of them
    (a looser threshold, never a wrong one) and the main launch scans the sampled rows like all others: all eight come back,
    in index order, on the pruned path."""
    rng = np.random.default_rng(9)
    n, d, b, k = 200_000, 256, 300, 100                            # (300 queries: padded to 512, the sample pass runs; k large enough for a speculative rank)
    x32 = rng.standard_normal((n, d))
    hot = rng.standard_normal(d)
    for r in (0, 1, 2, 3, 8, 9, 10, 11):
        x32[r] = hot
    x = to_f16_unit(x32)
    q32 = rng.standard_normal((b, d))
    q32[3] = hot + 0.05 * rng.standard_normal(d)
    q16 = to_f16_unit(q32)
    engine.set_option("dense_gemv", 0)
    engine.set_option("dense_shuffle", 0)
    try:
        engine.set_dense(x)
        ids, sc, ln = engine.dense_topk(q16, k)
        assert engine.dense_diag()["exhaustive"] == 0
        for i in (0, 1, 2, 3, 4, 150, 299):
            oid, osc = dense_exact_topk(x, q16[i], k)
            assert np.array_equal(ids[i], oid) and np.array_equal(sc[i].view(np.uint64), osc.view(np.uint64))
        assert list(ids[3, :8]) == [0, 1, 2, 3, 8, 9, 10, 11]
    finally:
        engine.set_option("dense_shuffle", 1)
        engine.set_option("dense_gemv", 1)


def test_dense_budgets_exhausted_still_answers(engine):
    """Corpora that exhaust the pruned pipeline's budgets (boilerplate-heavy manuals are exactly EasyRAG's
    data): 20000 exact copies of the chunk a query asks for (the 16384-entry candidate list cannot hold the tie
    block), 3000 near-copies whose fp32 scores all sit inside the pruning margin (the 1024-row fp64 re-score set is
    too small), and a batch in which 40 queries need the exhaustive path at once (more than one device-side round),
    alone and inside the fused dual route."""
    from easyrag_amd.engine import queries_to_csr
    from easyrag_amd.index import BM25S, build_bm25_index
    from oracle import BM25SLucene, bm25_filter, reciprocal_rank_fusion
    from oracle.retrievers import Item
    rng = np.random.default_rng(41)
    n, d, k = 50000, 128, 60
    x = to_f16_unit(rng.standard_normal((n, d)))
    hot = to_f16_unit(rng.standard_normal((1, d)))[0]
    copies = np.sort(rng.choice(n, size=20000, replace=False))
    x[copies] = hot
    near = to_f16_unit(rng.standard_normal((1, d)))[0]
    near_rows = np.setdiff1d(np.arange(n), copies)[:3000]
    x[near_rows] = near
    # flip the last mantissa bit of one component in half of the near-copies: fp32 scores differ by ~1e-6 (inside the
    # margin), fp64 scores differ for real, so the ranking among them must come from the exact re-score
    flip = near_rows[::2]
    col = int(np.argmax(np.abs(near.astype(np.float32))))
    x[flip, col] = np.nextafter(x[flip, col], np.float16(0))
    engine.set_dense(x)
    q16 = np.stack([hot, near] + [to_f16_unit(rng.standard_normal((1, d)))[0] for _ in range(3)])
    ids, sc, ln = engine.dense_topk(q16, k)
    assert engine.dense_diag()["exhaustive"] >= 2
    assert np.array_equal(ids[0], copies[:k])                          # the tie block: lowest indices first
    for i in range(q16.shape[0]):
        oid, osc = dense_exact_topk(x, q16[i], k)
        assert np.array_equal(ids[i], oid) and np.array_equal(sc[i].view(np.uint64), osc.view(np.uint64)), i
    # 40 flagged queries in one batch, with a dir filter on some
    dir_id = (np.arange(n) % 3).astype(np.int16)
    engine.set_doc_meta(n, None, dir_id)
    qb = np.stack([hot if i % 2 == 0 else near for i in range(40)] + [q16[3]])
    filt = np.full(41, -1, np.int16)
    filt[5], filt[6] = 1, 2
    ids, sc, ln = engine.dense_topk(qb, k, filter_dir=filt)
    assert 38 <= engine.dense_diag()["exhaustive"] <= 40               # (a filtered tie block may fit the budgets again)
    for i in (0, 1, 5, 6, 38, 39, 40):
        mask = None if filt[i] < 0 else dir_id == filt[i]
        oid, osc = dense_exact_topk(x, qb[i], k, mask)
        assert np.array_equal(ids[i], oid) and np.array_equal(sc[i].view(np.uint64), osc.view(np.uint64)), i
    # the fused dual route over the same batch: RRF must see the corrected dense lists
    flat, lens = synth.token_corpus(n, 500, seed=3, mean_len=12)
    docs = [list(map(int, t)) for t in synth.split_docs(flat, lens)]
    idx = build_bm25_index(docs, BM25S)
    engine.set_bm25(idx)
    queries = [list(map(int, t)) for t in synth.token_queries(flat, lens, 500, 41, seed=5)]
    qi, qt = queries_to_csr([idx.tokens_to_ids(t) for t in queries])
    engine.set_doc_meta(n, None, None)
    fid, fsc, fln = engine.hybrid_topk(qb, qi, qt, k_dense=k, k_sparse=50, K=60, topk=10)
    ora = BM25SLucene(1.5, 0.75).index(docs)
    for i in (0, 1, 39, 40):
        sp = bm25_filter(ora.get_scores(queries[i]), 50)
        oid, osc = dense_exact_topk(x, qb[i], k)
        want = reciprocal_rank_fusion([[Item(a, a, s) for a, s in sp],
                                       [Item(int(a), int(a), float(s)) for a, s in zip(oid, osc)]], K=60, topk=10)
        assert list(fid[i, :fln[i]]) == [w.idx for w in want] and list(fsc[i, :fln[i]]) == [w.score for w in want], i


def test_dense_d3584_exhaustive_path_and_rescore_tail(engine):
    """The reference's vector_size (3584) through the budget-exhausted cases: 17000 exact copies of the chunk a query asks for
    (candidate list overflow -> dense_exact_all_kernel with 28 KiB of query rows in LDS, seven rounds of 512 elements per lane),
    1500 near-copies inside the pruning margin (the fp64 re-score of more than 1024 rows: flagged as well), and plain queries
    whose 100 re-scored rows take the finalize kernel's d > 2048 tail.  ids and fp64 scores equal the oracle's."""
    rng = np.random.default_rng(59)
    n, d, k = 20000, 3584, 100
    x = to_f16_unit(rng.standard_normal((n, d), dtype=np.float32))
    hot = to_f16_unit(rng.standard_normal((1, d)))[0]
    copies = np.sort(rng.choice(n, size=17000, replace=False))
    x[copies] = hot
    near = to_f16_unit(rng.standard_normal((1, d)))[0]
    near_rows = np.setdiff1d(np.arange(n), copies)[:1500]
    x[near_rows] = near
    col = int(np.argmax(np.abs(near.astype(np.float32))))
    x[near_rows[::2], col] = np.nextafter(x[near_rows[::2], col], np.float16(0))
    engine.set_dense(x)
    q16 = np.stack([hot, near] + [to_f16_unit(rng.standard_normal((1, d)))[0] for _ in range(3)])
    for qb in (q16, np.concatenate([q16] * 4)):                       # 5 queries (skinny-GEMM stream) and 20 (padded scan)
        engine.reset_stats()
        ids, sc, ln = engine.dense_topk(qb, k)
        diag = engine.dense_diag()
        assert diag["exhaustive"] == 2 * (qb.shape[0] // 5) and engine.stat("dense_exhaustive_queries") == diag["exhaustive"]
        assert np.array_equal(ids[0], copies[:k])
        for i in range(5):
            oid, osc = dense_exact_topk(x, qb[i], k)
            assert np.array_equal(ids[i], oid) and np.array_equal(sc[i].view(np.uint64), osc.view(np.uint64)), i
        assert np.array_equal(ids[-1], ids[4]) and np.array_equal(sc[-2].view(np.uint64), sc[3].view(np.uint64))


def test_dense_fp32_inputs_normalised_on_device(engine):
    rng = np.random.default_rng(8)
    x32 = (rng.standard_normal((3000, 192)) * 3).astype(np.float32)
    q32 = (rng.standard_normal((9, 192)) * 0.1).astype(np.float32)
    engine.set_dense(x32, normalize=True)
    ids, sc, ln = engine.dense_topk(q32, 20, normalize_q=True)
    for i in range(9):
        qi, qs = qdrant_cosine_search(x32, q32[i], 20)             # reference semantics on the fp32 originals
        assert np.max(np.abs(qs - sc[i])) < 1e-3                   # fp16 storage of unit rows: ~1e-4
        assert len(set(qi) & set(ids[i])) >= 18


def test_dense_device_resident_inputs_all_dtypes(engine):
    """erh_set_dense / erh_dense_topk with DEVICE pointers: fp16 rows (placement kernel straight from the caller's
    buffer), fp32 rows (convert + normalise from the caller's buffer) and device queries give what the host paths give."""
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(41)
    x32 = (rng.standard_normal((4000, 128)) * 2).astype(np.float32)
    q32 = rng.standard_normal((7, 128)).astype(np.float32)
    dev = torch.device("cuda", 0)
    engine.set_dense(x32, normalize=True)
    want = engine.dense_topk(q32, 25, normalize_q=True)
    engine.set_dense(torch.from_numpy(x32).to(dev), normalize=True)
    got = engine.dense_topk(torch.from_numpy(q32).to(dev), 25, normalize_q=True)
    for a, b in zip(want, got):
        assert np.array_equal(np.asarray(a), np.asarray(b))
    x16 = to_f16_unit(x32)
    q16 = to_f16_unit(q32)
    engine.set_dense(x16)
    want = engine.dense_topk(q16, 25)
    engine.set_dense(torch.from_numpy(x16).to(dev))
    got = engine.dense_topk(torch.from_numpy(q16).to(dev), 25)
    for a, b in zip(want, got):
        assert np.array_equal(np.asarray(a), np.asarray(b))
    for i in (0, 6):
        oid, osc = dense_exact_topk(x16, q16[i], 25)
        assert np.array_equal(got[0][i], oid) and np.array_equal(got[1][i].view(np.uint64), osc.view(np.uint64))


def test_dense_errors(engine):
    x = synth.dense_corpus(100, 64, seed=1)
    engine.set_dense(x)
    with pytest.raises(_lib.ErhError):
        engine.dense_topk(x[:2], 0)
    with pytest.raises(_lib.ErhError):
        engine.dense_topk(x[:2], 5000)
    with pytest.raises(_lib.ErhError):
        engine.set_dense(np.zeros((10, 100), np.float16))          # d % 64 != 0
    from easyrag_amd.engine import RetrievalEngine
    fresh = RetrievalEngine(0)
    try:
        with pytest.raises(_lib.ErhError):
            fresh.dense_topk(x[:2], 5)                              # before set_dense
        fresh.set_dense(x)
        with pytest.raises(_lib.ErhError):
            fresh.dense_topk(x[:2], 5, filter_dir=np.array([1, 1], np.int16))   # filter without dir metadata
        ids, sc, ln = fresh.dense_topk(x[:2], 5, filter_dir=np.array([-1, -1], np.int16))
        assert np.all(ln == 5) and ids[0, 0] == 0 and ids[1, 0] == 1
    finally:
        fresh.close()


def test_dense_full_size_properties(engine, scan_cfg):
    """Config 2 shape (1M x 1024 fp16, B = 256, k = 100): size-independent properties + sampled exactness."""
    import torch
    dev = torch.device("cuda", 0)
    n, d, b, k = 1_000_000, 1024, 256, 100
    x = synth.dense_corpus_torch(n, d, seed=2, device=dev)
    q = synth.dense_queries_torch(x, b, seed=5)
    planted = torch.randint(0, n, (b,), generator=torch.Generator().manual_seed(9))
    q[:16] = x[planted[:16].to(dev)]                                # exact copies: top-1 must be the row itself
    engine.set_dense(x)
    ids, sc, ln = engine.dense_topk(q, k)
    diag = engine.dense_diag()
    assert diag["uncertified"] == 0 and diag["max_abs_err"] <= diag["margin"]
    assert np.all(ln == k) and np.all(ids >= 0) and np.all(ids < n)
    assert np.all(np.diff(sc, axis=1) <= 0)                        # sorted
    for i in range(b):
        assert len(set(ids[i])) == k                               # no duplicates
    for i in range(16):
        assert ids[i, 0] == int(planted[i]) or sc[i, 0] == sc[i, 1]
        assert abs(sc[i, 0] - 1.0) < 2e-3
    # returned fp64 scores are the pinned-order scores of exactly those rows
    xs = x[torch.from_numpy(ids[:4].reshape(-1).astype(np.int64)).to(dev)].cpu().numpy().reshape(4, k, d)
    qh = q[:4].cpu().numpy()
    for i in range(4):
        assert np.array_equal(dense_exact_scores(xs[i], qh[i]), sc[i])
    # no chunk outside the result beats the k-th score: exact check on a 200k-row slab for 4 queries
    slab = x[300_000:500_000].cpu().numpy()
    for i in range(4):
        s = slab.astype(np.float32) @ qh[i].astype(np.float32)
        top = np.argsort(-s)[:5] + 300_000
        for t in top:
            st = dense_exact_scores(slab[t - 300_000][None], qh[i])[0]
            assert (t in ids[i]) or st < sc[i, k - 1] or (st == sc[i, k - 1] and t > ids[i, k - 1])
    # idempotence
    ids2, sc2, _ = engine.dense_topk(q, k)
    assert np.array_equal(ids, ids2) and np.array_equal(sc, sc2)
