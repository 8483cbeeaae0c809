"""GPU parity of the 384 x 256 dense scan (dense_scan_pp5_kernel: the default for batches padded to >= 512 queries, i.e. for
the headline configuration) as a first-class arm: every adversarial dense case of tests/test_gpu_dense.py at a batch size that
reaches it -- record-buffer floods (the overflow re-run loop), exhausted budgets with the exhaustive hand-over behind a pp5
scan, duplicates + dir filter, a topic-sorted corpus with the row placement off, k at its maximum, N around 2 * 384 -- each
run with dense_tile384 = 1 and = 0, asserted bit-equal to each other AND to the oracle, and each asserting through
erh_get_stat that the kernel under test is the one that ran ("what's weak" 1a).

How a call reaches pp5 (csrc/pipeline_dense.hip: dense_topk_dev / scan_append): Bpad >= 512, d >= 256, N >= 768 and the scan stage starts
on a multiple of 384 rows -- the sample-pass path (c0 = 0: no filter, N >= 2 * sampled rows, speculative rank) or a staged
path whose seed prefix dense_n0 is a multiple of 384.
"""
import numpy as np
import pytest

from easyrag_amd import synth
from oracle import dense_exact_scores, dense_exact_topk, to_f16_unit

pytestmark = pytest.mark.gpu


def _both_tiles(engine, call, want_pp5=True):
    """Runs `call()` with the 384 x 256 tile and with the 256 x 256 tile; asserts which scan kernel answered; returns the two results."""
    out = []
    for tile384 in (1, 0):
        engine.set_option("dense_tile384", tile384)
        engine.reset_stats()
        try:
            res = call()
            diag = engine.dense_diag()
        finally:
            engine.set_option("dense_tile384", 1)
        st = engine.stats()
        if tile384 and want_pp5:
            assert st["dense_scan_pp5_launches"] >= 1, f"the 384 x 256 kernel did not run: {st}"
        else:
            assert st["dense_scan_pp5_launches"] == 0, st
            assert st["dense_scan_pp3_launches"] + st["dense_scan_tile_launches"] >= 1 or not want_pp5, st
        out.append((res, diag, st))
    (a, da, _), (b, db, _) = out
    for x, y in zip(a, b):
        x, y = np.asarray(x), np.asarray(y)
        assert np.array_equal(x.view(np.uint64) if x.dtype == np.float64 else x, y.view(np.uint64) if y.dtype == np.float64 else y), \
            "384 x 256 and 256 x 256 scans disagree"
    assert da["exhaustive"] == db["exhaustive"]
    return out


def _check(x, q16, ids, sc, ln, k, which, mask_of=None):
    for i in which:
        mask = None if mask_of is None else mask_of(i)
        oid, osc = dense_exact_topk(x, q16[i], k, mask)
        assert ln[i] == len(oid), f"query {i}: length {ln[i]} != {len(oid)}"
        assert np.array_equal(ids[i, :ln[i]], oid), f"query {i}: ids differ"
        assert np.array_equal(sc[i, :ln[i]].view(np.uint64), osc.view(np.uint64)), f"query {i}: fp64 scores differ"
        assert np.all(ids[i, ln[i]:] == -1)


def test_tile384_tie_block_floods_the_record_buffers(engine):
    """600 chunks equal to the 600 (identical) queries sit in consecutive 384-row tiles behind the sampled rows (row placement
    off: the caller's order is the stored order): every (chunk, query) pair of those tiles survives the threshold -- 64 queries
    x 192 rows = 12288 hits per wave and tile against a 254-record buffer, so the overflow re-run loop (`over_` / `shift_`) runs
    dozens of passes per tile, with nine-bit row records.  Result: the 100 lowest indices of the tie block, for every query."""
    rng = np.random.default_rng(21)
    n, d, b, k = 70000, 256, 600, 100
    x = to_f16_unit(rng.standard_normal((n, d)))
    hot = to_f16_unit(rng.standard_normal((1, d)))[0]
    x[40010:40610] = hot                                          # (straddles 384-row tile boundaries)
    q16 = np.repeat(hot[None, :], b, axis=0)
    engine.set_option("dense_shuffle", 0)
    engine.set_option("dense_gemv", 0)
    try:
        engine.set_dense(x)
        runs = _both_tiles(engine, lambda: engine.dense_topk(q16, k))
    finally:
        engine.set_option("dense_shuffle", 1)
        engine.set_option("dense_gemv", 1)
    (ids, sc, ln), diag, st = runs[0]
    assert st["dense_sample_passes"] == 1                          # the sample-pass path (c0 = 0)
    assert diag["uncertified"] == 0 and diag["exhaustive"] == 0
    oid, osc = dense_exact_topk(x, hot, k)
    assert np.array_equal(oid, np.arange(40010, 40110))
    for i in (0, 1, 63, 64, 255, 256, 511, 512, b - 1):
        assert ln[i] == k and np.array_equal(ids[i], oid) and np.array_equal(sc[i].view(np.uint64), osc.view(np.uint64)), i
    # the same block under the golden-ratio placement (the tie block is spread over every tile) gives the same answer
    engine.set_dense(x)
    ids2, sc2, ln2 = engine.dense_topk(q16[:520], k)
    assert np.array_equal(ids2[0], oid) and np.array_equal(ids2[519], oid) and np.array_equal(sc2[7].view(np.uint64), osc.view(np.uint64))


def test_tile384_budgets_exhausted_hand_over_to_the_exhaustive_path(engine):
    """20000 exact copies of one chunk (more than the 16384-entry candidate list holds) and 3000 near-copies whose fp32 scores
    all sit inside the pruning margin (more than the 1024-row re-score set), asked for by 48 of 512 queries: the 384 x 256
    scan overflows the lists, dense_finalize_kernel flags the queries, and the exhaustive path answers them -- three
    device-side rounds' worth, so erh_dense_check finishes the job -- while the other 464 queries stay on the pruned path."""
    rng = np.random.default_rng(43)
    n, d, b, k = 70000, 256, 512, 100                              # (k = 100: a speculative rank, i.e. the sample-pass path)
    x = to_f16_unit(rng.standard_normal((n, d)))
    hot = to_f16_unit(rng.standard_normal((1, d)))[0]
    copies = np.sort(rng.choice(n, size=20000, replace=False))
    x[copies] = hot
    near = to_f16_unit(rng.standard_normal((1, d)))[0]
    near_rows = np.setdiff1d(np.arange(n), copies)[:3000]
    x[near_rows] = near
    flip = near_rows[::2]
    col = int(np.argmax(np.abs(near.astype(np.float32))))
    x[flip, col] = np.nextafter(x[flip, col], np.float16(0))      # fp32 scores ~1e-6 apart, fp64 scores really different
    q16 = to_f16_unit(rng.standard_normal((b, d)))
    special = list(range(3, b, 11))[:48]                           # spread over all wave columns and both query tiles
    for j, i in enumerate(special):
        q16[i] = hot if j % 2 == 0 else near
    engine.set_option("dense_gemv", 0)
    try:
        engine.set_dense(x)
        runs = _both_tiles(engine, lambda: engine.dense_topk(q16, k))
    finally:
        engine.set_option("dense_gemv", 1)
    (ids, sc, ln), diag, st = runs[0]
    assert st["dense_sample_passes"] == 1
    # (the 48, plus the odd random query for which the 20000 copies happen to score above its threshold: they flood its list too)
    assert 48 <= diag["exhaustive"] <= 64 and st["dense_exhaustive_queries"] == diag["exhaustive"]
    assert np.array_equal(ids[special[0]], copies[:k])             # the tie block: lowest indices first
    _check(x, q16, ids, sc, ln, k, [0, 1, 2, special[0], special[1], special[2], special[-1], special[-2], 255, 256, b - 1])
    assert np.all(ln == k)


def test_tile384_duplicates_and_dir_filter(engine):
    """Every chunk eight times (exact ties everywhere) under a per-query `dir` filter: the staged path (a filter rules the
    sample pass out) with a seed prefix of 4 x 384 rows, so the append scan is the 384 x 256 kernel with its flush-time filter."""
    rng = np.random.default_rng(3)
    base = to_f16_unit(rng.standard_normal((2000, 256)))
    x = np.repeat(base, 8, axis=0)
    n, b, k = x.shape[0], 520, 12
    q16 = base[rng.integers(0, 2000, size=b)]
    dir_id = (np.arange(n) % 3).astype(np.int16)
    filt = rng.integers(-1, 3, size=b).astype(np.int16)
    filt[:4] = (0, 2, -1, 1)
    engine.set_option("dense_n0", 1536)
    engine.set_option("dense_gemv", 0)
    engine.set_option("dense_dir_blocks", 0)                        # the filter COLUMN on the 384-row tile is what is under test (by default these dirs get block copies)
    try:
        engine.set_dense(x)
        engine.set_doc_meta(n, None, dir_id)
        plain = _both_tiles(engine, lambda: engine.dense_topk(q16, k))
        filtered = _both_tiles(engine, lambda: engine.dense_topk(q16, k, filter_dir=filt))
    finally:
        engine.set_option("dense_dir_blocks", 1)
        engine.set_option("dense_n0", 32768)
        engine.set_option("dense_gemv", 1)
        engine.set_doc_meta(n, None, None)
    (ids, sc, ln), _, st = plain[0]
    assert st["dense_sample_passes"] == 0                          # N < 2 x sampled rows: the staged path, c0 = 1536
    _check(x, q16, ids, sc, ln, k, [0, 1, 2, 3, 100, 255, 256, 300, b - 1])
    (ids, sc, ln), _, _ = filtered[0]
    _check(x, q16, ids, sc, ln, k, [0, 1, 2, 3, 100, 255, 256, 300, b - 1],
           mask_of=lambda i: None if filt[i] < 0 else dir_id == filt[i])


def test_tile384_filter_at_full_seed_prefix(engine):
    """The default-sized seed prefix rounded to a multiple of 384 (32640) with a dir filter on 200000 rows: store kernel + seed
    select + ONE 384 x 256 append stage from row 32640 on, filter applied at flush time."""
    n, d, b, k = 200_000, 256, 512, 100
    x = synth.dense_corpus(n, d, seed=401)
    q16 = to_f16_unit(synth.dense_queries(x, b, seed=17))
    dir_id = (np.arange(n) % 7).astype(np.int16)
    filt = np.full(b, 3, np.int16)
    filt[::5] = -1
    engine.set_option("dense_n0", 32640)
    engine.set_option("dense_gemv", 0)
    engine.set_option("dense_dir_blocks", 0)                        # (the filter column: see above)
    try:
        engine.set_dense(x)
        engine.set_doc_meta(n, None, dir_id)
        runs = _both_tiles(engine, lambda: engine.dense_topk(q16, k, filter_dir=filt))
    finally:
        engine.set_option("dense_dir_blocks", 1)
        engine.set_option("dense_n0", 32768)
        engine.set_option("dense_gemv", 1)
        engine.set_doc_meta(n, None, None)
    (ids, sc, ln), diag, st = runs[0]
    assert diag["exhaustive"] == 0 and diag["uncertified"] == 0
    _check(x, q16, ids, sc, ln, k, [0, 1, 5, 256, b - 1], mask_of=lambda i: None if filt[i] < 0 else dir_id == filt[i])


def test_tile384_topic_sorted_corpus(engine):
    """A corpus ordered by topic, the queries' topic in the LAST 60 % of the rows.  With the golden-ratio row placement every
    sampled tile is an even sample of the caller's order: exact, nothing overflows.  With dense_shuffle = 0 the sample pass sees
    only the other topic, the threshold admits tens of thousands of chunks per query, every list overflows behind the 384 x 256
    scan, and all 512 queries are answered by the exhaustive path (32 rounds) -- same ids, same scores."""
    rng = np.random.default_rng(31)
    n, d, b, k = 150_000, 256, 512, 100
    topic_a, topic_b = rng.standard_normal(d), rng.standard_normal(d)
    x32 = rng.standard_normal((n, d)).astype(np.float32) * 0.35
    x32[: n * 2 // 5] += topic_a.astype(np.float32)
    x32[n * 2 // 5:] += topic_b.astype(np.float32)
    x = to_f16_unit(x32)
    del x32
    q16 = to_f16_unit(topic_b + 0.35 * rng.standard_normal((b, d)))
    which = [0, 1, 255, 256, b - 1]
    engine.set_option("dense_gemv", 0)
    try:
        engine.set_dense(x)
        runs = _both_tiles(engine, lambda: engine.dense_topk(q16, k))
        (ids, sc, ln), diag, st = runs[0]
        assert st["dense_sample_passes"] == 1 and diag["exhaustive"] == 0 and diag["uncertified"] == 0
        _check(x, q16, ids, sc, ln, k, which)
        engine.set_option("dense_shuffle", 0)
        engine.set_dense(x)
        runs0 = _both_tiles(engine, lambda: engine.dense_topk(q16, k))
        (ids0, sc0, ln0), diag0, st0 = runs0[0]
        assert diag0["exhaustive"] == b and st0["dense_exhaustive_queries"] == b
        assert np.array_equal(ids0, ids) and np.array_equal(sc0.view(np.uint64), sc.view(np.uint64)) and np.array_equal(ln0, ln)
    finally:
        engine.set_option("dense_shuffle", 1)
        engine.set_option("dense_gemv", 1)


@pytest.mark.parametrize("n,k,expect_pp5", [(767, 768, False), (768, 768, True), (769, 768, True), (1151, 288, True),
                                            (1153, 10, True)])
def test_tile384_k_at_its_maximum_and_n_around_two_tiles(engine, n, k, expect_pp5):
    """N just below / at / above 2 x 384 (below: the 256 x 256 kernel must take over), tiles that end inside the zero padding,
    k = 768 (the ABI's maximum) >= N: everything comes back, ranked.  Seed prefix = one 384-row tile, so the append stage starts
    on row 384 and -- from N = 768 on -- runs on the 384 x 256 kernel, most of whose workgroups have no tile at all."""
    d, b = 256, 512
    x = synth.dense_corpus(n, d, seed=n)
    q16 = to_f16_unit(synth.dense_queries(x, b, seed=k))
    engine.set_option("dense_n0", 384)
    engine.set_option("dense_gemv", 0)
    try:
        engine.set_dense(x)
        runs = _both_tiles(engine, lambda: engine.dense_topk(q16, k), want_pp5=expect_pp5)
    finally:
        engine.set_option("dense_n0", 32768)
        engine.set_option("dense_gemv", 1)
    (ids, sc, ln), diag, st = runs[0]
    kk = min(k, n)
    assert np.all(ln == kk) and diag["uncertified"] == 0
    _check(x, q16, ids, sc, ln, k, [0, 1, 255, 256, 300, b - 1])
    if k >= n:
        for i in (0, b - 1):
            assert sorted(ids[i, :kk]) == list(range(n))            # every chunk exactly once


def test_tile384_guaranteed_bounds_and_refinement_stages(engine):
    """dense_speculate = 0: guaranteed thresholds refined at stage boundaries (cand_refine_kernel between append launches).  With
    dense_n0 = 768 and dense_n1 = 3072 (dense_n1_auto off) the stage boundaries 768 / 3072 / 12288 are multiples of 384: three
    384 x 256 launches with c0 > 0, each clamped at its own c1."""
    n, d, b, k = 40000, 256, 520, 50
    x = synth.dense_corpus(n, d, seed=77)
    q16 = to_f16_unit(synth.dense_queries(x, b, seed=78))
    for name, v in (("dense_speculate", 0), ("dense_n0", 768), ("dense_n1", 3072), ("dense_n1_auto", 0), ("dense_gemv", 0)):
        engine.set_option(name, v)
    try:
        engine.set_dense(x)
        runs = _both_tiles(engine, lambda: engine.dense_topk(q16, k))
    finally:
        for name, v in (("dense_speculate", 1), ("dense_n0", 32768), ("dense_n1", 131072), ("dense_n1_auto", 1), ("dense_gemv", 1)):
            engine.set_option(name, v)
    (ids, sc, ln), diag, st = runs[0]
    assert st["dense_scan_pp5_launches"] == 3 and st["dense_scan_pp3_launches"] == 0, st
    assert diag["uncertified"] == 0 and diag["exhaustive"] == 0
    _check(x, q16, ids, sc, ln, k, [0, 1, 2, 255, 256, 511, 512, b - 1])


def test_tile384_copy_that_does_not_fit_falls_back_to_the_256_tile(engine):
    """the 384-row tiled copy doubles the chunk matrix; when its allocation fails the call must not -- the
    256 x 256 scan, which needs no copy, answers.  The out-of-memory answer is forced with the dense_tile384_max_mb hook (a
    copy above the limit is refused as hipMalloc would refuse it); the result is the oracle's, the fallback is counted, and
    the next erh_set_dense tries again."""
    n, d, b, k = 70000, 256, 512, 100
    x = synth.dense_corpus(n, d, seed=5)
    q16 = to_f16_unit(synth.dense_queries(x, b, seed=6))
    engine.set_option("dense_gemv", 0)
    try:
        engine.set_option("dense_tile384_max_mb", 1)                 # the copy is 34 MiB
        engine.set_dense(x)
        engine.reset_stats()
        ids, sc, ln = engine.dense_topk(q16, k)
        st = engine.stats()
        assert st["dense_tile384_nomem"] == 1 and st["dense_scan_pp5_launches"] == 0 and st["dense_scan_pp3_launches"] >= 1, st
        engine.dense_topk(q16, k)
        assert engine.stat("dense_tile384_nomem") == 1             # not retried on every call
        _check(x, q16, ids, sc, ln, k, [0, 255, 256, b - 1])
        engine.set_option("dense_tile384_max_mb", -1)
        engine.set_dense(x)                                        # a new matrix: the copy is tried again and fits
        engine.reset_stats()
        ids2, sc2, ln2 = engine.dense_topk(q16, k)
        assert engine.stat("dense_scan_pp5_launches") == 1
        assert np.array_equal(ids, ids2) and np.array_equal(sc.view(np.uint64), sc2.view(np.uint64))
    finally:
        engine.set_option("dense_tile384_max_mb", -1)
        engine.set_option("dense_gemv", 1)


def test_tile384_scores_of_the_returned_rows_are_the_pinned_order_scores(engine):
    """Cross-check that does not go through dense_exact_topk: the fp64 scores returned for 1024 queries are exactly
    dense_exact_scores of the returned rows, sorted, without duplicates, and the FAST (fp32 MFMA) ranking of the same call
    agrees with them to 1e-3 (north_star's dense tolerance)."""
    n, d, b, k = 200_000, 256, 1024, 288                           # (4 x speculative rank <= the 256 cells of a 1024-query sample)
    x = synth.dense_corpus(n, d, seed=9)
    q16 = to_f16_unit(synth.dense_queries(x, b, seed=10))
    engine.set_dense(x)
    engine.reset_stats()
    ids, sc, ln = engine.dense_topk(q16, k)
    fids, fsc, fln = engine.dense_topk(q16, k, mode=1)
    assert engine.stat("dense_scan_pp5_launches") == 2 and engine.stat("dense_sample_passes") == 2
    assert np.all(ln == k) and np.all(fln == k)
    assert np.all(np.diff(sc, axis=1) <= 0)
    for i in (0, 1, 511, 512, 1023):
        assert len(set(ids[i])) == k
        assert np.array_equal(dense_exact_scores(x, q16[i], rows=ids[i]), sc[i])
        assert np.max(np.abs(fsc[i] - dense_exact_scores(x, q16[i], rows=fids[i]))) < 1e-3
        assert len(set(fids[i]) & set(ids[i])) >= k - 2
