"""GPU parity, dense route with the `dir` filter pushed down as a ROW RANGE (round 5).

The reference filters every real query on its document directory (pipeline.py:301-312 builds `filters` -- a qdrant `must` on `dir`,
ingestion.py:207-216 -- and `filter_dict` from the question's "document" field and pushes them into the retrievers, pipeline.py:331-341;
applied at retrievers.py:44-47 and 198-202), and its loader walks the directories one after the other, so each dir is one run of consecutive
chunks.  libeasyrag_hip keeps a copy of every dir's chunks (wherever they lie in the caller's numbering) with the block's own row placement
and lets a filtered query scan its block only; the results must be what the filter column on the whole matrix gives (dense_dir_blocks = 0,
itself checked against the oracle's masked ranking throughout the suite) and what
the oracle gives: ids in the caller's numbering, fp64 scores bit for bit, canonical ties.
"""
import numpy as np
import pytest

from oracle import dense_exact_topk, to_f16_unit

pytestmark = pytest.mark.gpu


def _blocks(sizes):
    return np.repeat(np.arange(len(sizes)), sizes).astype(np.int16)


def _same(a, b):
    (ia, sa, la), (ib, sb, lb) = a, b
    assert np.array_equal(la, lb)
    for i in range(len(la)):
        assert np.array_equal(ia[i, :la[i]], ib[i, :lb[i]]), i
        assert np.array_equal(sa[i, :la[i]].view(np.uint64), sb[i, :lb[i]].view(np.uint64)), i


def _check_oracle(x, q16, k, dir_id, filt, got, rows):
    ids, sc, ln = got
    for i in rows:
        mask = None if filt[i] < 0 else dir_id == filt[i]
        oid, osc = dense_exact_topk(x, q16[i], k, mask)
        assert ln[i] == len(oid), i
        assert np.array_equal(ids[i, :ln[i]], oid), i
        assert np.array_equal(sc[i, :ln[i]].view(np.uint64), osc.view(np.uint64)), i


@pytest.fixture(params=[(1, 1), (1, 0), (0, 0)], ids=["one-launch-per-stage", "one-launch-per-stage-store-kernel-seed", "pipeline-per-group"])
def blocks_opts(engine, request):
    """The ways of running a batch's block groups: ONE launch per stage over all of them through the view table (round 6, the default) -- with
    the thresholds from a sample pass per view where every view qualifies, or always from store kernel + seed select -- and every group as a
    pipeline of its own (round 5; what a group with a flagged query still falls back to)."""
    engine.set_option("dense_dir_blocks", 2)                        # every batch with a block dir takes the route (1 = by the work estimate)
    engine.set_option("dense_group_launch", request.param[0])
    engine.set_option("dense_group_sample", request.param[1])
    engine.group_launch = request.param[0]
    yield engine
    engine.set_option("dense_group_sample", 1)
    engine.set_option("dense_group_launch", 1)
    engine.set_option("dense_dir_blocks", 1)
    engine.set_option("dense_dir_block_min_rows", 4096)
    engine.set_option("dense_shuffle", 1)


def test_blocks_mixed_batch_against_oracle(blocks_opts):
    """Four dirs of uneven size (one below the block minimum), a batch mixing all of them with unfiltered queries: three block
    groups + one ordinary group with a filter column, scattered back to the caller's order."""
    engine = blocks_opts
    rng = np.random.default_rng(501)
    sizes = [5000, 12000, 3000, 20000]
    n, d, b, k = sum(sizes), 256, 48, 50
    x = to_f16_unit(rng.standard_normal((n, d)) + 0.4 * rng.standard_normal(d))
    q16 = to_f16_unit(x[rng.integers(0, n, b)].astype(np.float32) + 0.3 * rng.standard_normal((b, d)))
    dir_id = _blocks(sizes)
    filt = rng.integers(-1, 4, b).astype(np.int16)
    filt[:5] = [-1, 0, 1, 2, 3]
    engine.set_dense(x)
    engine.set_doc_meta(n, None, dir_id)
    engine.set_option("dense_dir_blocks", 0)
    plain = engine.dense_topk(q16, k, filter_dir=filt)
    engine.set_option("dense_dir_blocks", 2)
    engine.reset_stats()
    routed = engine.dense_topk(q16, k, filter_dir=filt)
    assert engine.stat("dense_block_groups") == 3                 # dirs 0, 1, 3; dir 2 (3000 rows) rides with the unfiltered queries
    assert engine.stat("dense_grouped_launches") == engine.group_launch
    assert engine.dense_diag()["uncertified"] == 0
    _same(plain, routed)
    _check_oracle(x, q16, k, dir_id, filt, routed, range(b))
    engine.set_option("dense_scan_nt", 1)                          # the grouped scan's chunk-side loads with the non-temporal hint: an arm, same lists
    try:
        _same(routed, engine.dense_topk(q16, k, filter_dir=filt))
    finally:
        engine.set_option("dense_scan_nt", 0)
    # device outputs: complete after dense_check, identical
    import torch
    out = engine.dense_topk(torch.from_numpy(q16).cuda(), k, device_out=True, filter_dir=filt)
    engine.dense_check()
    dev = tuple(t.cpu().numpy() for t in out)
    _same(routed, dev)
    # a second matrix on the same handle: the blocks follow it
    x2 = to_f16_unit(rng.standard_normal((n, d)))
    engine.set_dense(x2)
    engine.set_doc_meta(n, None, dir_id)
    again = engine.dense_topk(q16, k, filter_dir=filt)
    _check_oracle(x2, q16, k, dir_id, filt, again, (0, 1, 2, 3, 4, b - 1))


def test_blocks_fp32_queries_fast_mode_and_device_inputs(blocks_opts):
    """The other faces of erh_dense_topk through the route: fp32 queries normalised on the device (the gathered batch keeps the caller's
    dtype), device-resident queries, and ERH_DENSE_FAST (ranked by the fp32 MFMA score: within 1e-3 of the exact list, ids inside
    the asked dir)."""
    import torch
    from easyrag_amd import _lib
    engine = blocks_opts
    rng = np.random.default_rng(505)
    sizes = [6000, 9000, 5000]
    n, d, b, k = sum(sizes), 256, 21, 30
    x32 = (rng.standard_normal((n, d)) * 2).astype(np.float32)
    q32 = (x32[rng.integers(0, n, b)] + 0.8 * rng.standard_normal((b, d))).astype(np.float32) * 0.3
    dir_id = _blocks(sizes)
    filt = (np.arange(b) % 4 - 1).astype(np.int16)                  # -1 (no filter), 0, 1, 2
    engine.set_dense(x32, normalize=True)
    engine.set_doc_meta(n, None, dir_id)
    engine.set_option("dense_dir_blocks", 0)
    plain = engine.dense_topk(q32, k, filter_dir=filt, normalize_q=True)
    engine.set_option("dense_dir_blocks", 2)
    engine.reset_stats()
    routed = engine.dense_topk(q32, k, filter_dir=filt, normalize_q=True)
    assert engine.stat("dense_block_groups") == 3
    _same(plain, routed)
    dev_q = engine.dense_topk(torch.from_numpy(q32).cuda(), k, filter_dir=filt, normalize_q=True)
    _same(plain, dev_q)
    x16 = to_f16_unit(x32)
    q16 = to_f16_unit(q32)
    engine.set_dense(x16)                                           # (the host's rounding of the unit rows, as the oracle has them)
    engine.set_doc_meta(n, None, dir_id)
    _check_oracle(x16, q16, k, dir_id, filt, engine.dense_topk(q16, k, filter_dir=filt), range(b))
    fids, fsc, fln = engine.dense_topk(q16, k, filter_dir=filt, mode=_lib.ERH_DENSE_FAST)
    eids, esc, eln = engine.dense_topk(q16, k, filter_dir=filt)
    assert np.array_equal(fln, eln)
    for i in range(b):
        assert np.max(np.abs(fsc[i] - esc[i])) < 1e-3 and len(set(fids[i]) & set(eids[i])) >= k - 2
        assert filt[i] < 0 or np.all(dir_id[fids[i]] == filt[i])


def test_blocks_ties_across_blocks_and_tiny_blocks(blocks_opts):
    """The same 300 chunks repeated in every block (exact ties between blocks and inside them; ids must stay inside the asked
    block, lowest first), blocks smaller than k (the list is the block) and of one row, with the block minimum lowered to 1."""
    engine = blocks_opts
    rng = np.random.default_rng(502)
    d, k = 256, 40
    base = to_f16_unit(rng.standard_normal((300, d)))
    sizes = [1, 30, 300, 900, 2400]
    x = np.concatenate([np.tile(base, (max(1, s // 300), 1))[:s] if s >= 300 else base[:s] for s in sizes])
    n = x.shape[0]
    assert n == sum(sizes)
    dir_id = _blocks(sizes)
    b = 20
    q16 = to_f16_unit(base[rng.integers(0, 300, b)].astype(np.float32) + 0.05 * rng.standard_normal((b, d)))
    filt = (np.arange(b) % 5).astype(np.int16)
    engine.set_option("dense_dir_block_min_rows", 1)
    engine.set_dense(x)
    engine.set_doc_meta(n, None, dir_id)
    engine.reset_stats()
    routed = engine.dense_topk(q16, k, filter_dir=filt)
    assert engine.stat("dense_block_groups") == 5
    ids, sc, ln = routed
    assert list(ln[:5]) == [1, 30, 40, 40, 40]
    _check_oracle(x, q16, k, dir_id, filt, routed, range(b))
    engine.set_option("dense_dir_blocks", 0)
    _same(engine.dense_topk(q16, k, filter_dir=filt), routed)


def test_blocks_exhaustive_path_inside_a_block(blocks_opts):
    """A block that is one chunk 6000 times: every candidate budget overflows on ties, the group's queries take the exhaustive
    path INSIDE the block (its rows, its ids shifted by the block's first document)."""
    engine = blocks_opts
    rng = np.random.default_rng(503)
    d, k = 256, 64
    a = to_f16_unit(rng.standard_normal((5000, d)))
    one = to_f16_unit(rng.standard_normal((1, d)))
    c = to_f16_unit(rng.standard_normal((7000, d)))
    x = np.concatenate([a, np.repeat(one, 6000, axis=0), c])
    n = x.shape[0]
    dir_id = _blocks([5000, 6000, 7000])
    b = 9
    q16 = to_f16_unit(one.astype(np.float32) + 0.2 * rng.standard_normal((b, d)))
    filt = np.array([1, 1, 1, 0, 2, 1, -1, 0, 2], np.int16)
    engine.set_dense(x)
    engine.set_doc_meta(n, None, dir_id)
    routed = engine.dense_topk(q16, k, filter_dir=filt)
    diag = engine.dense_diag()
    assert diag["uncertified"] == 0 and diag["exhaustive"] >= 4    # the four queries of dir 1 at least (the unfiltered one sees the ties too)
    ids, sc, ln = routed
    for i in np.flatnonzero(filt == 1):
        assert np.array_equal(ids[i], np.arange(5000, 5000 + k))   # all ties: lowest ids of the block
    _check_oracle(x, q16, k, dir_id, filt, routed, range(b))
    # device outputs: the flagged group is run again inside erh_dense_check
    import torch
    from easyrag_amd import synth
    from easyrag_amd.engine import queries_to_csr
    from easyrag_amd.index import BM25S, build_bm25_index_from_postings
    out = engine.dense_topk(torch.from_numpy(q16).cuda(), k, device_out=True, filter_dir=filt)
    engine.dense_check()
    assert engine.dense_diag()["exhaustive"] == diag["exhaustive"]
    _same(routed, tuple(t.cpu().numpy() for t in out))
    # the fused call on top: its RRF is redone over the corrected dense lists (host and device outputs, against the filter column)
    dev = torch.device("cuda", 0)
    indptr, doc, tf, lens, flat = synth.token_csr_torch(n, 4096, seed=9, device=dev)
    engine.set_bm25(build_bm25_index_from_postings(indptr, doc, tf, lens, BM25S, compute_payload=False), payload_on_device=True)
    csr = queries_to_csr(synth.token_queries(flat, lens, 4096, b, seed=90))
    engine.set_option("dense_dir_blocks", 0)
    want = engine.hybrid_topk(q16, *csr, k_dense=k, k_sparse=50, K=60, topk=10, filter_dir=filt)
    engine.set_option("dense_dir_blocks", 2)
    got = engine.hybrid_topk(q16, *csr, k_dense=k, k_sparse=50, K=60, topk=10, filter_dir=filt)
    out = engine.hybrid_topk(torch.from_numpy(q16).cuda(), *csr, k_dense=k, k_sparse=50, K=60, topk=10, filter_dir=filt, device_out=True)
    engine.dense_check()
    for a_, b_, c_ in zip(want, got, out):
        a_, b_, c_ = np.asarray(a_), np.asarray(b_), c_.cpu().numpy()
        assert np.array_equal(a_.view(np.uint64) if a_.dtype == np.float64 else a_, b_.view(np.uint64) if b_.dtype == np.float64 else b_)
        assert np.array_equal(a_.view(np.uint64) if a_.dtype == np.float64 else a_, c_.view(np.uint64) if c_.dtype == np.float64 else c_)


def test_blocks_of_scattered_dirs_and_many_groups(blocks_opts):
    """Interleaved dirs (document i in dir i % 4: a corpus that was NOT loaded dir by dir) get block copies too -- the class' documents
    gathered in ascending order, block rows mapped back through the id table; twelve dirs in one batch are twelve groups (round 5
    stopped at eight and fell back to the filter column)."""
    engine = blocks_opts
    rng = np.random.default_rng(504)
    n, d, b, k = 30000, 256, 36, 25
    x = to_f16_unit(rng.standard_normal((n, d)))
    q16 = to_f16_unit(rng.standard_normal((b, d)))
    engine.set_option("dense_dir_block_min_rows", 1)
    engine.set_dense(x)
    dir_id = (np.arange(n) % 4).astype(np.int16)
    engine.set_doc_meta(n, None, dir_id)
    filt = (np.arange(b) % 4).astype(np.int16)
    engine.reset_stats()
    got = engine.dense_topk(q16, k, filter_dir=filt)
    assert engine.stat("dense_block_groups") == 4
    _check_oracle(x, q16, k, dir_id, filt, got, range(b))
    engine.set_option("dense_dir_blocks", 0)
    _same(engine.dense_topk(q16, k, filter_dir=filt), got)
    engine.set_option("dense_dir_blocks", 2)
    engine.reset_stats()
    dir_id = _blocks([2500] * 12)
    engine.set_doc_meta(n, None, dir_id)
    filt = (np.arange(b) % 12).astype(np.int16)
    got = engine.dense_topk(q16, k, filter_dir=filt)
    assert engine.stat("dense_block_groups") == 12 and engine.stat("dense_grouped_launches") == engine.group_launch
    _check_oracle(x, q16, k, dir_id, filt, got, range(b))
    engine.set_option("dense_dir_blocks", 0)
    _same(engine.dense_topk(q16, k, filter_dir=filt), got)
    engine.set_option("dense_dir_blocks", 2)
    engine.reset_stats()
    filt = (np.arange(b) % 3 + 4).astype(np.int16)                 # three of the twelve
    got = engine.dense_topk(q16, k, filter_dir=filt)
    assert engine.stat("dense_block_groups") == 3
    _check_oracle(x, q16, k, dir_id, filt, got, (0, 1, 2, b - 1))
    # no room for the block copies (the test hook that also bounds the 384-row copy): the filter column answers
    engine.set_option("dense_tile384_max_mb", 0)
    try:
        engine.set_doc_meta(n, None, dir_id)
        engine.reset_stats()
        got = engine.dense_topk(q16, k, filter_dir=filt)
        assert engine.stat("dense_block_groups") == 0
        _check_oracle(x, q16, k, dir_id, filt, got, (0, 1, 2, b - 1))
    finally:
        engine.set_option("dense_tile384_max_mb", -1)
        engine.set_doc_meta(n, None, dir_id)


def test_blocks_large_batch_and_hybrid(blocks_opts):
    """640 queries over four blocks of 50000 chunks (groups of 160 queries on the 256 x 256 scan over 50000 rows each) against the
    filter column on the 384 x 256 scan over all 200000 rows; the hybrid call on top of both gives the same fused lists."""
    import torch
    from easyrag_amd import synth
    from easyrag_amd.engine import queries_to_csr
    from easyrag_amd.index import BM25S, build_bm25_index_from_postings
    engine = blocks_opts
    n, d, b, k = 200_000, 256, 640, 288
    dev = torch.device("cuda", 0)
    xt = synth.dense_corpus_torch(n, d, seed=7, device=dev)
    qt = synth.dense_queries_torch(xt, b, seed=70)
    x = xt.cpu().numpy()
    q16 = qt.cpu().numpy()
    dir_id = _blocks([50000] * 4)
    filt = ((np.arange(b) * 7) % 4).astype(np.int16)
    engine.set_dense(xt)
    engine.set_doc_meta(n, None, dir_id)
    engine.set_option("dense_dir_blocks", 0)
    plain = engine.dense_topk(q16, k, filter_dir=filt)
    engine.set_option("dense_dir_blocks", 2)
    engine.reset_stats()
    routed = engine.dense_topk(q16, k, filter_dir=filt)
    assert engine.stat("dense_block_groups") == 4
    assert engine.stat("dense_grouped_launches") == engine.group_launch
    assert engine.stat("dense_scan_pp3_launches") == (1 if engine.group_launch else 4)      # ONE scan launch over the four blocks
    assert engine.dense_diag()["uncertified"] == 0
    _same(plain, routed)
    _check_oracle(x, q16, k, dir_id, filt, routed, (0, 1, 2, 3, 317, b - 1))
    indptr, doc, tf, lens, flat = synth.token_csr_torch(n, 32768, seed=8, device=dev)
    engine.set_bm25(build_bm25_index_from_postings(indptr, doc, tf, lens, BM25S, compute_payload=False), payload_on_device=True)
    csr = queries_to_csr(synth.token_queries(flat, lens, 32768, b, seed=80))
    # the same filter on both routes, then a different dir per route and no filter on the sparse one (the reference keeps the two apart:
    # filter_dict -> BM25, filters -> Qdrant; retrievers.py:278,283)
    other = ((filt + 1) % 4).astype(np.int16)
    for f_sparse, f_dense in ((filt, "same"), (filt, other), (None, other)):
        outs = []
        for on in (0, 1):
            engine.set_option("dense_dir_blocks", 2 * on)
            engine.reset_stats()
            outs.append(engine.hybrid_topk(q16, *csr, k_dense=288, k_sparse=192, K=60, topk=10, filter_dir=f_sparse, filter_dense=f_dense))
            assert engine.stat("dense_block_groups") == 4 * on
        for a, c in zip(outs[0], outs[1]):
            a, c = np.asarray(a), np.asarray(c)
            assert np.array_equal(a.view(np.uint64) if a.dtype == np.float64 else a, c.view(np.uint64) if c.dtype == np.float64 else c)
    engine.set_option("dense_dir_blocks", 1)


@pytest.mark.parametrize("seed", [601, 602, 603, 604, 605, 606])
def test_blocks_random_shapes_against_filter_column(blocks_opts, seed):
    """Seeded random shapes -- 2 ... 7 dirs of 1 ... 9000 chunks, as runs of consecutive documents or scattered over the corpus (some below the
    block minimum, some empty classes in between, some documents without a dir), 1 ... 300
    queries with k up to 300 and filter values that include none (-1), classes without documents and classes beyond the table: the route and
    the filter column must agree on every list, bit for bit; a sample of the queries is checked against the oracle."""
    engine = blocks_opts
    rng = np.random.default_rng(seed)
    n_dirs = int(rng.integers(2, 8))
    sizes = [int(rng.choice([1, 40, 700, 3000, 9000])) for _ in range(n_dirs)]
    labels = rng.permutation(n_dirs + 2)[:n_dirs]                    # class ids in any order, two ids of the range unused
    dir_id = np.concatenate([np.full(s, l, np.int16) for s, l in zip(sizes, labels)])
    n = int(dir_id.shape[0])
    if seed % 2 == 0:                                                # every other case: the dirs scattered over the corpus, some documents
        dir_id = rng.permutation(dir_id)                             # without any dir (-1)
        dir_id[rng.integers(0, n, max(1, n // 50))] = -1
    d = int(rng.choice([64, 256, 320]))                              # (64: below the persistent scan's 8 K-stages, every group its own pipeline)
    b = int(rng.choice([1, 7, 65, 300]))
    k = int(rng.choice([1, 10, 100, 300]))
    x = to_f16_unit(rng.standard_normal((n, d)) + 0.3 * rng.standard_normal(d))
    q16 = to_f16_unit(x[rng.integers(0, n, b)].astype(np.float32) + 0.4 * rng.standard_normal((b, d)))
    filt = rng.integers(-1, n_dirs + 4, b).astype(np.int16)
    engine.set_option("dense_dir_block_min_rows", int(rng.choice([1, 500, 4096])))
    engine.set_dense(x)
    engine.set_doc_meta(n, None, dir_id)
    engine.set_option("dense_dir_blocks", 0)
    plain = engine.dense_topk(q16, k, filter_dir=filt)
    engine.set_option("dense_dir_blocks", 2)
    routed = engine.dense_topk(q16, k, filter_dir=filt)
    assert engine.dense_diag()["uncertified"] == 0
    _same(plain, routed)
    _check_oracle(x, q16, k, dir_id, filt, routed, sorted(set([0, b // 2, b - 1])))


def test_grouped_launch_tile_shapes(engine):
    """The view table's corner cases in one batch of 700 queries: a dir with 300 queries (two query tiles on one block: 256 + 44), dirs
    with 129 / 128 / 1 queries (whole and half query tiles), a block shorter than the seed prefix (the store kernel scores all of it,
    no scan stage), a block of 40 rows (fewer than k: the list is the block) and of one row, and queries without a filter riding along
    as the ordinary group.  Then the same blocks with at most 128 queries each: the half-query-tile instantiation of the scan."""
    rng = np.random.default_rng(707)
    sizes = [70000, 45000, 36000, 9000, 40, 1, 52000]
    n, d, k = sum(sizes), 256, 60
    x = to_f16_unit(rng.standard_normal((n, d)) + 0.3 * rng.standard_normal(d))
    dir_id = _blocks(sizes)
    counts = [300, 129, 128, 60, 9, 3, 1]
    filt = np.concatenate([np.full(c, i, np.int16) for i, c in enumerate(counts)] + [np.full(70, -1, np.int16)])
    filt = rng.permutation(filt)
    b = filt.shape[0]
    q16 = to_f16_unit(x[rng.integers(0, n, b)].astype(np.float32) + 0.4 * rng.standard_normal((b, d)))
    engine.set_option("dense_dir_block_min_rows", 1)
    engine.set_option("dense_dir_blocks", 2)
    try:
        engine.set_dense(x)
        engine.set_doc_meta(n, None, dir_id)
        engine.reset_stats()
        got = engine.dense_topk(q16, k, filter_dir=filt)
        st, diag = engine.stats(), engine.dense_diag()
        assert st["dense_block_groups"] == 7 and st["dense_grouped_launches"] == 1 and diag["uncertified"] == 0, (st, diag)
        assert diag["exhaustive"] == 0 and diag["max_abs_err"] <= diag["margin"]
        engine.set_option("dense_group_launch", 0)
        _same(engine.dense_topk(q16, k, filter_dir=filt), got)
        engine.set_option("dense_dir_blocks", 0)
        _same(engine.dense_topk(q16, k, filter_dir=filt), got)
        sample = [int(np.flatnonzero(filt == c)[j]) for c in range(7) for j in (0, -1)] + [int(np.flatnonzero(filt < 0)[0])]
        _check_oracle(x, q16, k, dir_id, filt, got, sample)
        # at most 128 queries per dir: every tile is a half tile
        engine.set_option("dense_group_launch", 1)
        engine.set_option("dense_dir_blocks", 2)
        keep = np.concatenate([np.flatnonzero(filt == c)[:128] for c in range(7)])
        engine.reset_stats()
        half = engine.dense_topk(q16[keep], k, filter_dir=filt[keep])
        assert engine.stat("dense_grouped_launches") == 1
        for j, i in enumerate(keep):
            assert half[2][j] == got[2][i] and np.array_equal(half[0][j], got[0][i])
            assert np.array_equal(half[1][j].view(np.uint64), got[1][i].view(np.uint64))
    finally:
        engine.set_option("dense_group_launch", 1)
        engine.set_option("dense_dir_blocks", 1)
        engine.set_option("dense_dir_block_min_rows", 4096)


def test_route_decision_by_work(engine):
    """dense_dir_blocks = 1 (the default) routes where the routed work -- rows x max(query columns, dense_route_ridge) summed over the
    groups -- is smaller than the whole-matrix work; no table of measured milliseconds.  Four blocks of 30000 rows: one filtered query
    (a quarter of the rows), 4 x 64 and 4 x 160 queries (half tiles / whole tiles against one padded 256 / 768-column scan of everything)
    route; 4 x 8 queries do not (32 columns over the matrix once against 4 x a block at the ridge: the same rows)."""
    rng = np.random.default_rng(808)
    n, d, k = 120_000, 256, 40
    x = to_f16_unit(rng.standard_normal((n, d)))
    dir_id = _blocks([30000] * 4)
    engine.set_dense(x)
    engine.set_doc_meta(n, None, dir_id)
    engine.set_option("dense_dir_blocks", 1)
    for per_dir, want_groups, want_grouped in ((64, 4, 1), (160, 4, 1), (8, 0, 0)):
        b = 4 * per_dir
        q16 = to_f16_unit(rng.standard_normal((b, d)))
        filt = (np.arange(b) % 4).astype(np.int16)
        engine.reset_stats()
        got = engine.dense_topk(q16, k, filter_dir=filt)
        assert engine.stat("dense_block_groups") == want_groups and engine.stat("dense_grouped_launches") == want_grouped, per_dir
        _check_oracle(x, q16, k, dir_id, filt, got, (0, 1, b // 2, b - 1))
    q16 = to_f16_unit(rng.standard_normal((1, d)))
    engine.reset_stats()
    got = engine.dense_topk(q16, k, filter_dir=np.array([2], np.int16))
    assert engine.stat("dense_block_groups") == 1 and engine.stat("dense_grouped_launches") == 0
    _check_oracle(x, q16, k, dir_id, np.array([2], np.int16), got, (0,))
    # the ridge is an option: at 1 column every comparison is by columns alone, and 4 x 8 queries (4 x 128 columns of a block = 128 of the
    # matrix, against 32 of the matrix) still do not route; at 4096 nothing but a smaller row count routes
    engine.set_option("dense_route_ridge", 4096)
    try:
        b = 256
        q16 = to_f16_unit(rng.standard_normal((b, d)))
        filt = (np.arange(b) % 4).astype(np.int16)
        engine.reset_stats()
        engine.dense_topk(q16, k, filter_dir=filt)
        assert engine.stat("dense_block_groups") == 0               # the same rows either way
        filt = (np.arange(b) % 2).astype(np.int16)
        engine.dense_topk(q16, k, filter_dir=filt)
        assert engine.stat("dense_block_groups") == 2               # half the rows
    finally:
        engine.set_option("dense_route_ridge", 160)


def test_grouped_launch_at_the_reference_vector_size(engine):
    """d = 3584 (the reference's `vector_size`, gte-Qwen2-7B: ref src/configs/easyrag.yaml:15-16) with the reference's real call pattern: every
    query filtered on its dir, the dirs uneven runs of consecutive chunks (41 / 36 / 15 / 11 % like the 103 questions' documents).  112 K-stages
    per tile through the grouped store kernel, the grouped persistent scan and the final kernel's d > 2048 tail; ids and pinned-order fp64
    scores against the oracle, and the filter column bit for bit."""
    rng = np.random.default_rng(909)
    sizes = [24600, 21600, 9000, 6600]
    n, d, b, k = sum(sizes), 3584, 300, 100
    x = to_f16_unit(rng.standard_normal((n, d), dtype=np.float32))
    dir_id = _blocks(sizes)
    filt = rng.choice(4, size=b, p=[0.41, 0.36, 0.15, 0.08]).astype(np.int16)
    filt[:4] = [0, 1, 2, 3]
    q16 = to_f16_unit(x[rng.integers(0, n, b)].astype(np.float32) + 0.4 * rng.standard_normal((b, d), dtype=np.float32) / np.sqrt(d) * 8)
    engine.set_option("dense_dir_blocks", 2)
    try:
        engine.set_dense(x)
        engine.set_doc_meta(n, None, dir_id)
        engine.reset_stats()
        got = engine.dense_topk(q16, k, filter_dir=filt)
        st, diag = engine.stats(), engine.dense_diag()
        assert st["dense_grouped_launches"] == 1 and st["dense_block_groups"] == 4 and diag["uncertified"] == 0, (st, diag)
        assert diag["max_abs_err"] <= diag["margin"]
        engine.set_option("dense_dir_blocks", 0)
        _same(engine.dense_topk(q16, k, filter_dir=filt), got)
        _check_oracle(x, q16, k, dir_id, filt, got, (0, 1, 2, 3, 150, b - 1))
    finally:
        engine.set_option("dense_dir_blocks", 1)
        engine.set_doc_meta(n, None, None)


def test_grouped_launch_sample_pass(engine):
    """Three blocks of 120 k / 90 k / 70 k chunks, 200 queries each: every view is large enough for a threshold sample of its own, so the
    grouped launch draws it with the scan kernel (one sample pass over the first tiles of every view's chunk streams, cell select with the
    view's rank, the main launch over ALL rows) instead of store kernel + seed scores + seed select.  Same lists as the store-kernel seed, the
    per-group pipelines and the filter column, bit for bit; a sample of them against the oracle."""
    rng = np.random.default_rng(1212)
    sizes = [120_000, 90_000, 70_000]
    n, d, b, k = sum(sizes), 256, 600, 100
    x = to_f16_unit(rng.standard_normal((n, d), dtype=np.float32) + 0.3 * rng.standard_normal(d).astype(np.float32))
    dir_id = _blocks(sizes)
    filt = (np.arange(b) % 3).astype(np.int16)
    q16 = to_f16_unit(x[rng.integers(0, n, b)].astype(np.float32) + 0.4 * rng.standard_normal((b, d), dtype=np.float32))
    engine.set_option("dense_dir_blocks", 2)
    try:
        engine.set_dense(x)
        engine.set_doc_meta(n, None, dir_id)
        engine.reset_stats()
        got = engine.dense_topk(q16, k, filter_dir=filt)
        st, diag = engine.stats(), engine.dense_diag()
        assert st["dense_grouped_launches"] == 1 and st["dense_sample_passes"] == 1 and st["dense_scan_pp3_launches"] == 1, st
        assert diag["uncertified"] == 0 and diag["exhaustive"] == 0 and diag["max_abs_err"] <= diag["margin"], diag
        engine.set_option("dense_group_sample", 0)
        engine.reset_stats()
        _same(engine.dense_topk(q16, k, filter_dir=filt), got)
        assert engine.stat("dense_sample_passes") == 0 and engine.stat("dense_grouped_launches") == 1
        engine.set_option("dense_group_launch", 0)
        _same(engine.dense_topk(q16, k, filter_dir=filt), got)
        engine.set_option("dense_dir_blocks", 0)
        _same(engine.dense_topk(q16, k, filter_dir=filt), got)
        _check_oracle(x, q16, k, dir_id, filt, got, (0, 1, 2, 299, 598, 599))
        # k so deep that a view's speculative rank reaches it: the batch takes the store-kernel seed, same contract
        engine.set_option("dense_dir_blocks", 2)
        engine.set_option("dense_group_launch", 1)
        engine.set_option("dense_group_sample", 1)
        engine.reset_stats()
        deep = engine.dense_topk(q16[:90], 700, filter_dir=filt[:90])
        assert engine.stat("dense_grouped_launches") == 1
        _check_oracle(x, q16, 700, dir_id, filt, deep, (0, 1, 2))
    finally:
        engine.set_option("dense_group_sample", 1)
        engine.set_option("dense_group_launch", 1)
        engine.set_option("dense_dir_blocks", 1)
        engine.set_doc_meta(n, None, None)
