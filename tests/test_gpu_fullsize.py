"""GPU parity at BASELINE.json's full sizes (1M chunks x 1024, ~60M postings): EXACT ids and scores against the oracle
composition on sampled queries, for the configurations round 1 only property-checked:

  configs[3]  dense(288) + BM25(192) + RRF -> top-10, B = 1024, both BM25 variants      (32 sampled queries each)
  configs[2]  BM25 only, B = 256 (two document segments per query + merge), k = 100 / 192 (32 sampled queries)
  Okapi       float64 accumulation, 61 tiles of 16384 documents                          (12 sampled queries)
  configs[3]  bm25s, ALL 1024 queries of the batch (round 4; ~2 minutes of oracle work on the host)
  configs[3]  with a `dir` filter on both routes (one column per route, as the reference keeps them apart) and 1 % of the
              documents sharing their content with another one (SURVEY.md 8(d) config 4 variant), 32 sampled queries

How the oracle stays affordable at 1M documents without weakening it:
  sparse  the per-posting payload comes from the host index (easyrag_amd/index.py, bit-equal to the dict-loop
          rank_bm25 / bm25s restatements on small corpora: tests/test_index_vs_oracle.py); a query's score vector is
          the oracle's scatter-add over the postings in token order (numpy, accumulation type of the variant), then
          oracle.bm25_filter.
  dense   an independent fp32 GEMM (torch, not this library) proposes every row within 2e-3 of the k-th best fp32
          score -- |fp32 - exact| <= d * 2^-23 * |x||q| ~ 1.2e-4 for these unit vectors, so the proposal provably
          contains the exact top-k -- and the oracle's pinned-order float64 scores of those rows decide.
"""
import numpy as np
import pytest

from easyrag_amd import synth
from easyrag_amd.engine import queries_to_csr
from easyrag_amd.index import BM25S, OKAPI, build_bm25_index_from_postings
from oracle import bm25_filter, dense_exact_scores, reciprocal_rank_fusion
from oracle.retrievers import Item

pytestmark = pytest.mark.gpu

N, D, VOCAB = 1_000_000, 1024, 262_144


@pytest.fixture(scope="module")
def dense_data():
    import torch
    dev = torch.device("cuda", 0)
    x = synth.dense_corpus_torch(N, D, seed=2, device=dev)
    q = synth.dense_queries_torch(x, 1024, seed=1000)
    return x, q


@pytest.fixture(scope="module")
def sparse_data():
    import torch
    dev = torch.device("cuda", 0)
    indptr, doc, tf, lens, flat = synth.token_csr_torch(N, VOCAB, seed=3, device=dev)
    queries = synth.token_queries(flat, lens, VOCAB, 1024, seed=2000)
    torch.cuda.empty_cache()
    return indptr, doc, tf, lens, queries, flat


_INDEX = {}


def host_index(sparse_data, variant):
    """Host-built index (payload included) per variant, built once per session."""
    if variant not in _INDEX:
        indptr, doc, tf, lens = sparse_data[:4]
        _INDEX[variant] = build_bm25_index_from_postings(indptr, doc, tf, lens, variant)
    return _INDEX[variant]


def sparse_oracle_scores(idx, q_tokens):
    """The variant's get_scores over the CSR postings: contributions added in query-token order, repeats included,
    in the variant's accumulation type."""
    acc = np.zeros(idx.n_docs, np.float64 if idx.variant == OKAPI else np.float32)
    for t in q_tokens:
        s, e = idx.indptr[t], idx.indptr[t + 1]
        np.add.at(acc, idx.doc_ids[s:e], idx.payload[s:e])
    return acc


def sparse_oracle_topk(idx, q_tokens, k, keep_mask=None):
    """bm25_filter over the documents the query touches at all (the walk stops at the first score <= 0, so the others never
    appear; the index map is monotone, so (score desc, index asc) is the same order): 7 x cheaper than sorting 1M scores."""
    acc = sparse_oracle_scores(idx, q_tokens)
    nz = np.nonzero(acc)[0]
    got = bm25_filter(acc[nz], k, keep_mask=None if keep_mask is None else keep_mask[nz])
    return [(int(nz[i]), s) for i, s in got]


def dense_oracle_topk(x, q16_rows, k, allowed=None):
    """Exact (pinned-order fp64, index-ascending ties) top-k of each query row over the whole matrix (`allowed`: per query
    a boolean row mask or None -- the payload filter of the Qdrant search, applied before the limit)."""
    import torch
    out = []
    qf = q16_rows.float()
    s32 = torch.empty((qf.shape[0], x.shape[0]), dtype=torch.float32, device=x.device)
    for s in range(0, x.shape[0], 131072):
        s32[:, s:s + 131072] = qf @ x[s:s + 131072].float().T
    if allowed is not None:
        for i, m in enumerate(allowed):
            if m is not None:
                s32[i, ~m] = -float("inf")
    kth = torch.topk(s32, k, dim=1).values[:, -1]
    for i in range(qf.shape[0]):
        cand = torch.nonzero(s32[i] >= kth[i] - 2e-3).reshape(-1)
        rows = x[cand].cpu().numpy()
        sc = dense_exact_scores(rows, q16_rows[i].cpu().numpy())
        ids = cand.cpu().numpy()
        order = np.lexsort((ids, -sc))[:k]
        out.append((ids[order].astype(np.int64), sc[order]))
    return out


@pytest.mark.parametrize("variant", [BM25S, OKAPI], ids=["bm25s", "okapi"])
def test_hybrid_configs3_exact(engine, dense_data, sparse_data, variant):
    x, q = dense_data
    queries = sparse_data[4]
    idx = host_index(sparse_data, variant)                                          # host payload = the oracle's data
    engine.set_dense(x)
    engine.set_bm25(idx, payload_on_device=True)                                    # the GPU evaluates its own payload
    assert np.array_equal(engine.get_bm25_payload(), idx.payload)
    engine.set_doc_meta(N, None, None)
    qi, qt = queries_to_csr(queries)
    ids, sc, ln = engine.hybrid_topk(q, qi, qt, k_dense=288, k_sparse=192, K=60, topk=10)
    assert np.all(ln == 10)
    sample = list(range(0, 1024, 33))[:32]
    dense_want = dense_oracle_topk(x, q[sample], 288)
    for (did, dsc), b in zip(dense_want, sample):
        sp = bm25_filter(sparse_oracle_scores(idx, queries[b]), 192)
        want = reciprocal_rank_fusion([[Item(i, i, s) for i, s in sp],
                                       [Item(int(i), int(i), float(s)) for i, s in zip(did, dsc)]], K=60, topk=10)
        assert list(ids[b, :ln[b]]) == [w.idx for w in want], f"query {b}: fused ids differ"
        assert list(sc[b, :ln[b]]) == [w.score for w in want], f"query {b}: fused scores differ"
    # the two routes on their own, same sample (what feeds the fusion)
    d_ids, d_sc, d_ln = engine.dense_topk(q[sample], 288)
    for r, (did, dsc) in enumerate(dense_want):
        assert np.array_equal(d_ids[r], did) and np.array_equal(d_sc[r], dsc)


@pytest.mark.parametrize("speculate", [1, 0], ids=["speculative-threshold", "guaranteed-bounds"])
def test_dense_configs1_exact(engine, dense_data, speculate):
    """BASELINE.json configs[1] ITSELF (the configuration the >= 0.70 HBM target and `sub_benchmarks.dense_b256_top100` are
    defined on): 1M x 1024 fp16, dense only, 256 queries, k = 100, library defaults -- the seed prefix scored by the store
    kernel, `seed_select`, ONE 256-query tile of `dense_scan_pp3_kernel` appending from row `dense_n0` to row 1M (the
    stage-boundary arithmetic at this N), finalize.  40 sampled queries (both ends of the batch, both ends of the query
    tile's wave columns) against the oracle: ids and pinned-order fp64 scores bit for bit; the whole batch against the
    structural properties; and with `dense_speculate = 0` (guaranteed bounds + refinement stages) the same lists."""
    x, q = dense_data
    b, k = 256, 100
    engine.set_dense(x)
    engine.set_doc_meta(N, None, None)
    engine.set_option("dense_speculate", speculate)
    try:
        engine.reset_stats()
        ids, sc, ln = engine.dense_topk(q[:b].contiguous(), k)
        st, diag = engine.stats(), engine.dense_diag()
    finally:
        engine.set_option("dense_speculate", 1)
    assert np.all(ln == k)
    assert diag["uncertified"] == 0 and diag["max_abs_err"] <= diag["margin"] and diag["exhaustive"] == 0, diag
    # the store kernel + pp3 append pipeline answered: no sample pass, no 384-row tile, no skinny-GEMM stream, no lock-step tiles
    assert st["dense_scan_pp3_launches"] >= 1 and st["dense_sample_passes"] == 0 and st["dense_scan_pp5_launches"] == 0, st
    assert st["dense_scan_gemv_launches"] == 0 and st["dense_exhaustive_queries"] == 0, st
    if speculate:
        assert st["dense_scan_pp3_launches"] == 1, st                 # one scan stage covers [n0, N)
    sample = sorted(set(list(range(0, b, 8)) + [1, 63, 64, 127, 128, 191, 254, 255]))
    assert len(sample) >= 32
    want = dense_oracle_topk(x, q[sample], k)
    for (oid, osc), i in zip(want, sample):
        assert np.array_equal(ids[i], oid), f"query {i}: ids differ"
        assert np.array_equal(sc[i].view(np.uint64), osc.view(np.uint64)), f"query {i}: fp64 scores differ"
    for i in range(b):
        assert len(set(ids[i])) == k and ids[i].min() >= 0 and ids[i].max() < N and np.all(np.diff(sc[i]) <= 0)


@pytest.mark.parametrize("variant,k,n_sample", [(BM25S, 100, 32), (BM25S, 192, 32), (OKAPI, 100, 12), (OKAPI, 192, 12)],
                         ids=["bm25s-k100", "bm25s-k192", "okapi-k100", "okapi-k192"])
@pytest.mark.parametrize("ascan,wscan,crossing", [(1, 0, 1), (0, 1, 2), (0, 1, 0), (0, 0, 0)],
                         ids=["approx-scan-rescore", "wave-owned-crossings", "wave-owned-sweep", "block-scan"])
def test_bm25_configs2_exact(engine, sparse_data, variant, k, n_sample, ascan, wscan, crossing):
    """B = 256: the document range of every query is split over two workgroups (segments) whose lists are merged."""
    queries = sparse_data[4]
    idx = host_index(sparse_data, variant)
    engine.set_option("bm25_ascan", ascan)
    engine.set_option("bm25_wscan", wscan)
    engine.set_option("bm25_crossing", crossing)
    try:
        engine.set_bm25(idx)
        qi, qt = queries_to_csr(queries[:256])
        ids, sc, ln = engine.bm25_topk(qi, qt, k)
    finally:
        engine.set_option("bm25_ascan", 1)
        engine.set_option("bm25_wscan", 0)
        engine.set_option("bm25_crossing", 1)
    assert np.all(ln == k)
    for b in list(range(0, 256, 256 // n_sample))[:n_sample]:
        want = bm25_filter(sparse_oracle_scores(idx, queries[b]), k)
        assert list(ids[b, :ln[b]]) == [w[0] for w in want], f"query {b}: ids differ"
        assert list(sc[b, :ln[b]]) == [w[1] for w in want], f"query {b}: scores differ"


@pytest.mark.parametrize("variant", [BM25S, OKAPI], ids=["bm25s", "okapi"])
def test_device_index_build_full_size(engine, sparse_data, variant):
    """f3: 1M documents / ~56M tokens indexed on the device (erh_build_bm25_index): CSR, tf, idf and payload equal the
    host builder's bit for bit (the postings themselves also equal the torch-sorted CSR the other tests use)."""
    import time
    indptr, doc, tf, lens, _, flat = sparse_data
    want = host_index(sparse_data, variant)
    t0 = time.perf_counter()
    got = engine.build_bm25(flat.astype(np.int32), lens.astype(np.int32), VOCAB, variant=variant)
    dt = time.perf_counter() - t0
    print(f"device index build, {flat.shape[0]} tokens -> {got.nnz} postings: {dt:.2f} s (including the copies back)")
    assert np.array_equal(got.indptr, indptr) and np.array_equal(got.doc_ids, doc) and np.array_equal(got.tf, tf)
    assert np.array_equal(got.idf, want.idf) and got.avgdl == want.avgdl
    assert np.array_equal(got.payload, want.payload)


def test_hybrid_configs3_all_queries(engine, dense_data, sparse_data):
    """every one of the 1024 queries of configs[3] (bm25s), not a sample: fused ids and fp64 RRF scores."""
    import torch
    x, q = dense_data
    queries = sparse_data[4]
    idx = host_index(sparse_data, BM25S)
    engine.set_dense(x)
    engine.set_bm25(idx, payload_on_device=True)
    engine.set_doc_meta(N, None, None)
    qi, qt = queries_to_csr(queries)
    ids, sc, ln = engine.hybrid_topk(q, qi, qt, k_dense=288, k_sparse=192, K=60, topk=10)
    assert np.all(ln == 10)
    bad = []
    for lo in range(0, 1024, 128):                                                   # the fp32 proposal matrix in slabs of 128 queries
        dense_want = dense_oracle_topk(x, q[lo:lo + 128], 288)
        torch.cuda.empty_cache()
        for r, (did, dsc) in enumerate(dense_want):
            b = lo + r
            sp = sparse_oracle_topk(idx, queries[b], 192)
            want = reciprocal_rank_fusion([[Item(i, i, s) for i, s in sp],
                                           [Item(int(i), int(i), float(s)) for i, s in zip(did, dsc)]], K=60, topk=10)
            if list(ids[b, :ln[b]]) != [w.idx for w in want] or list(sc[b, :ln[b]]) != [w.score for w in want]:
                bad.append(b)
    assert not bad, f"{len(bad)} of 1024 queries differ from the oracle: {bad[:20]}"


def test_hybrid_configs3_filters_and_duplicate_contents(engine, dense_data, sparse_data):
    """VERDICT r3 4(b) / SURVEY.md 8(d) config 4 variant at 1M: a `dir` equality filter pushed into both routes (per query:
    none / the same class on both / sparse route only / a different class per route -- the reference keeps filter_dict and
    filters apart, retrievers.py:278,283) and 1 % of the documents sharing their content with an earlier one, with the
    shared contents planted INSIDE the top lists of the sampled queries so that the RRF really merges them (key = content,
    += per occurrence, the last-seen node is returned: retrievers.py:256-274)."""
    import torch
    from collections import Counter
    x, q = dense_data
    queries = sparse_data[4]
    idx = host_index(sparse_data, BM25S)
    rng = np.random.default_rng(77)
    dir_id = rng.choice(4, size=N, p=[0.40, 0.35, 0.15, 0.10]).astype(np.int16)     # the reference's four manuals, unevenly sized
    fs = np.full(1024, -1, np.int16)
    fd = np.full(1024, -1, np.int16)
    cls = rng.integers(0, 4, size=1024).astype(np.int16)
    b_idx = np.arange(1024)
    fs[b_idx % 4 == 1] = cls[b_idx % 4 == 1]; fd[b_idx % 4 == 1] = cls[b_idx % 4 == 1]
    fs[b_idx % 4 == 2] = cls[b_idx % 4 == 2]
    fs[b_idx % 4 == 3] = cls[b_idx % 4 == 3]; fd[b_idx % 4 == 3] = (cls[b_idx % 4 == 3] + 1) % 4
    sample = list(range(0, 1024, 33))[:32]
    engine.set_dense(x)
    engine.set_bm25(idx, payload_on_device=True)
    engine.set_doc_meta(N, None, dir_id)
    qi, qt = queries_to_csr(queries)
    # the filtered route lists of the sampled queries (no content ids yet): where the duplicates are planted
    allowed = [None if fd[b] < 0 else torch.from_numpy(dir_id == fd[b]).to(x.device) for b in sample]
    dense_want = dense_oracle_topk(x, q[sample], 288, allowed)
    sparse_want = [sparse_oracle_topk(idx, queries[b], 192, None if fs[b] < 0 else (dir_id == fs[b])) for b in sample]
    content = np.arange(N, dtype=np.int32)
    used = set()

    def tie(a, b2):
        a, b2 = int(a), int(b2)
        if a == b2 or a in used or b2 in used:
            return
        used.update((a, b2))
        content[max(a, b2)] = min(a, b2)                                           # content id = smallest index with that text

    for (did, _), sp in zip(dense_want, sparse_want):
        for j in range(0, 40, 2):
            tie(did[j], did[j + 1])                                                 # two dense hits with one text
        for j in range(min(20, len(sp))):
            tie(sp[j][0], did[100 + j])                                             # a sparse hit and a dense hit with one text
        for j in range(60, min(100, len(sp)) - 1, 2):
            tie(sp[j][0], sp[j + 1][0])                                             # two sparse hits with one text
    planted = len(used) // 2
    while len(used) < 20000:                                                        # up to 1 % of the corpus: 10,000 shared pairs
        a, b2 = rng.integers(0, N, size=2)
        tie(a, b2)
    assert planted > 1500 and int((content != np.arange(N)).sum()) == 10000
    engine.set_doc_meta(N, content, dir_id)
    try:
        ids, sc, ln = engine.hybrid_topk(q, qi, qt, k_dense=288, k_sparse=192, K=60, topk=10, filter_dir=fs, filter_dense=fd)
        merged = 0
        for (did, dsc), sp, b in zip(dense_want, sparse_want, sample):
            lists = [[Item(i, int(content[i]), s) for i, s in sp],
                     [Item(int(i), int(content[i]), float(s)) for i, s in zip(did, dsc)]]
            want = reciprocal_rank_fusion(lists, K=60, topk=10)
            assert list(ids[b, :ln[b]]) == [w.idx for w in want], f"query {b}: fused ids differ"
            assert list(sc[b, :ln[b]]) == [w.score for w in want], f"query {b}: fused scores differ"
            seen = Counter(it.content for lst in lists for it in lst)
            merged += sum(1 for w in want if seen[int(content[w.idx])] > 1)
        assert merged > 0                                                           # shared contents reached the fused top-10
        d_ids, d_sc, d_ln = engine.dense_topk(q[sample], 288, filter_dir=fd[sample])
        for r, (did, dsc) in enumerate(dense_want):
            assert np.array_equal(d_ids[r, :d_ln[r]], did) and np.array_equal(d_sc[r, :d_ln[r]], dsc)
        qi_s, qt_s = queries_to_csr([queries[b] for b in sample])
        s_ids, s_sc, s_ln = engine.bm25_topk(qi_s, qt_s, 192, filter_dir=fs[sample])
        for r, sp in enumerate(sparse_want):
            assert list(s_ids[r, :s_ln[r]]) == [i for i, _ in sp] and list(s_sc[r, :s_ln[r]]) == [s for _, s in sp]
    finally:
        engine.set_doc_meta(N, None, None)


def test_hybrid_configs3_dir_blocks(engine, dense_data, sparse_data):
    """The reference's real call pattern at full size: every query filtered on its document directory, the directories four
    contiguous blocks of uneven size (400k / 350k / 150k / 100k chunks).  With the library's defaults the dense route answers
    each dir's queries from that dir's block copy (dense_dir_blocks: 4 groups of 256 queries) and the BM25 scan walks the dir's
    posting tiles only; the fused lists must be the oracle's for the sampled queries and, for all 1024, bit for bit what the
    filter column over the whole matrix gives (dense_dir_blocks = 0).  One query per call takes the same route."""
    import torch
    x, q = dense_data
    queries = sparse_data[4]
    idx = host_index(sparse_data, BM25S)
    dir_id = np.repeat(np.arange(4), [400_000, 350_000, 150_000, 100_000]).astype(np.int16)
    assert dir_id.shape[0] == N
    filt = (np.arange(1024) % 4).astype(np.int16)
    sample = list(range(0, 1024, 41))[:24]
    engine.set_dense(x)
    engine.set_bm25(idx, payload_on_device=True)
    engine.set_doc_meta(N, None, dir_id)
    qi, qt = queries_to_csr(queries)
    try:
        engine.reset_stats()
        ids, sc, ln = engine.hybrid_topk(q, qi, qt, k_dense=288, k_sparse=192, K=60, topk=10, filter_dir=filt)
        assert engine.stat("dense_block_groups") == 4 and engine.stat("dense_scan_pp5_launches") == 0
        assert engine.stat("dense_grouped_launches") == 1 and engine.stat("dense_sample_passes") == 1 and engine.stat("dense_scan_pp3_launches") == 1
        assert engine.dense_diag()["uncertified"] == 0
        allowed = [torch.from_numpy(dir_id == filt[b]).to(x.device) for b in sample]
        dense_want = dense_oracle_topk(x, q[sample], 288, allowed)
        for (did, dsc), b in zip(dense_want, sample):
            sp = sparse_oracle_topk(idx, queries[b], 192, dir_id == filt[b])
            want = reciprocal_rank_fusion([[Item(i, i, s) for i, s in sp],
                                           [Item(int(i), int(i), float(s)) for i, s in zip(did, dsc)]], K=60, topk=10)
            assert list(ids[b, :ln[b]]) == [w.idx for w in want], f"query {b}: fused ids differ"
            assert list(sc[b, :ln[b]]) == [w.score for w in want], f"query {b}: fused scores differ"
        d_ids, d_sc, d_ln = engine.dense_topk(q, 288, filter_dir=filt)
        for (did, dsc), b in zip(dense_want, sample):
            assert np.array_equal(d_ids[b, :d_ln[b]], did) and np.array_equal(d_sc[b, :d_ln[b]], dsc)
        # one query per call: its dir's block only
        engine.reset_stats()
        for (did, dsc), b in list(zip(dense_want, sample))[:4]:
            i1, s1, l1 = engine.dense_topk(q[b:b + 1], 288, filter_dir=filt[b:b + 1])
            assert np.array_equal(i1[0, :l1[0]], did) and np.array_equal(s1[0, :l1[0]], dsc)
        assert engine.stat("dense_block_groups") == 4
        # the grouped launch with the store-kernel seed instead of the sample pass, and one pipeline per group: the same lists
        for name in ("dense_group_sample", "dense_group_launch"):
            engine.set_option(name, 0)
            engine.reset_stats()
            g_ids, g_sc, g_ln = engine.dense_topk(q, 288, filter_dir=filt)
            assert engine.stat("dense_block_groups") == 4 and engine.stat("dense_sample_passes") == 0
            assert np.array_equal(g_ln, d_ln) and np.array_equal(g_ids, d_ids) and np.array_equal(g_sc.view(np.uint64), d_sc.view(np.uint64))
        engine.set_option("dense_group_sample", 1)
        engine.set_option("dense_group_launch", 1)
        engine.set_option("dense_dir_blocks", 0)
        ids0, sc0, ln0 = engine.hybrid_topk(q, qi, qt, k_dense=288, k_sparse=192, K=60, topk=10, filter_dir=filt)
        p_ids, p_sc, p_ln = engine.dense_topk(q, 288, filter_dir=filt)
        assert np.array_equal(ln0, ln) and np.array_equal(ids0, ids) and np.array_equal(sc0.view(np.uint64), sc.view(np.uint64))
        assert np.array_equal(p_ln, d_ln) and np.array_equal(p_ids, d_ids) and np.array_equal(p_sc.view(np.uint64), d_sc.view(np.uint64))
    finally:
        engine.set_option("dense_group_sample", 1)
        engine.set_option("dense_group_launch", 1)
        engine.set_option("dense_dir_blocks", 1)
        engine.set_doc_meta(N, None, None)


# ---- the reference's own vector size: d = 3584 (ref:src/configs/easyrag.yaml:15-16, gte-Qwen2-7B-instruct) ------------------------
N3584, D3584 = 285_696, 3584          # 744 x 384 rows x 3584 halves = 2.05 GB: the byte volume of the 1M x 1024 configurations


@pytest.fixture(scope="module")
def dense_3584():
    import torch
    dev = torch.device("cuda", 0)
    x = synth.dense_corpus_torch(N3584, D3584, seed=12, device=dev)
    q = synth.dense_queries_torch(x, 1024, seed=1012)
    return x, q


@pytest.mark.parametrize("b,k", [(256, 100), (1024, 288), (1, 288), (40, 100)])
def test_dense_d3584_full_size_exact(engine, dense_3584, b, k):
    """2.05 GB of 3584-d chunks (the reference's vector_size) through every batch regime -- one query (skinny-GEMM stream, 112 KiB of
    query fragments), 40 (two column groups do not fit LDS at this row length: the padded scan), 256 (one query tile, store
    kernel + seed select), 1024 (sample pass + 384 x 256 scan, 112 K-stages per tile) -- against the oracle: an independent
    fp32 GEMM proposes every row within 2e-3 of the k-th best (|fp32 - exact| <= d 2^-23 |x||q| = 4.3e-4 here), the pinned-order
    float64 scores of those rows decide; ids and scores bit for bit."""
    x, q = dense_3584
    engine.set_dense(x)
    engine.reset_stats()
    ids, sc, ln = engine.dense_topk(q[:b].contiguous(), k)
    diag, st = engine.dense_diag(), engine.stats()
    assert diag["uncertified"] == 0 and diag["max_abs_err"] <= diag["margin"] and diag["exhaustive"] == 0
    assert np.all(ln == k)
    if b == 1024:
        assert st["dense_scan_pp5_launches"] == 1 and st["dense_sample_passes"] == 1, st
    elif b == 1:
        assert st["dense_scan_gemv_launches"] == 1, st
    else:
        assert st["dense_scan_pp3_launches"] == 1, st
    sample = sorted(set([0, 1, b // 3, b // 2, b - 2, b - 1, min(b - 1, 255), min(b - 1, 256), min(b - 1, 700)]) & set(range(b)))
    want = dense_oracle_topk(x, q[sample], k)
    for (oid, osc), i in zip(want, sample):
        assert np.array_equal(ids[i], oid), f"query {i}: ids differ"
        assert np.array_equal(sc[i].view(np.uint64), osc.view(np.uint64)), f"query {i}: fp64 scores differ"
    for i in range(b):
        assert len(set(ids[i])) == k and np.all(np.diff(sc[i]) <= 0)


# ---- anisotropic, topic-sorted embeddings with exact duplicates (the numbers must not be an artefact of Gaussian data) -----
@pytest.mark.parametrize("b,k", [(1024, 288), (256, 100), (1, 288)])
def test_dense_clustered_corpus_exact(engine, b, k):
    """1M x 1024 chunks of synth.clustered_corpus_torch -- rows sorted by topic, random-pair cosine ~0.4, intra-topic ~0.7, 2 % exact copies of
    another row of the topic -- with queries that are noisy copies of corpus rows: every member of the query's topic scores far above the
    rest of the corpus, the duplicates tie exactly.  32 sampled queries against the oracle (ids + pinned-order fp64 scores bit for bit),
    through the batch regimes the bench times (1024: sample pass + 384 x 256 scan; 256: store kernel + 256 x 256 scan; 1: skinny-GEMM
    stream), and the whole batch must stay on the pruned pipeline (no query handed to the exhaustive path)."""
    import torch
    dev = torch.device("cuda", 0)
    x = synth.clustered_corpus_torch(N, D, seed=21, device=dev)
    q = synth.dense_queries_torch(x, b, seed=5000)
    # the shape the generator promises (fp32 on a sample of rows)
    xs = x[torch.arange(0, N, 977, device=dev)].float()
    cos = xs @ xs.T
    off = cos[~torch.eye(cos.shape[0], dtype=torch.bool, device=dev)]
    assert 0.33 < float(off.mean()) < 0.47
    same_topic = (x[:400].float() @ x[:400].float().T)[~torch.eye(400, dtype=torch.bool, device=dev)]
    assert 0.62 < float(same_topic.mean()) < 0.78
    engine.set_dense(x)
    engine.set_doc_meta(N, None, None)
    engine.reset_stats()
    ids, sc, ln = engine.dense_topk(q, k)
    diag, st = engine.dense_diag(), engine.stats()
    assert np.all(ln == k)
    assert diag["uncertified"] == 0 and diag["max_abs_err"] <= diag["margin"], diag
    assert diag["exhaustive"] == 0 and st["dense_exhaustive_queries"] == 0, (diag, st)
    if b == 1024:
        assert st["dense_scan_pp5_launches"] == 1 and st["dense_sample_passes"] == 1, st
    sample = sorted(set(range(0, b, max(1, b // 32))))[:32]
    want = dense_oracle_topk(x, q[sample], k)
    for (oid, osc), i in zip(want, sample):
        assert np.array_equal(ids[i], oid), f"query {i}: ids differ"
        assert np.array_equal(sc[i].view(np.uint64), osc.view(np.uint64)), f"query {i}: fp64 scores differ"
    cand = engine.dense_candidates_last_call() / b
    print(f"clustered corpus, {b} queries, k = {k}: {cand:.0f} candidates per query reached the final kernel")
    assert k <= cand < 16384
    del x
    torch.cuda.empty_cache()


def test_bm25_reference_question_lengths_full_size(engine, sparse_data):
    """configs[2]-sized corpus, 1024 queries with the length distribution of the reference's 103 real questions (4 ... 45 tokens).  On the
    packed 16-bit shape alone a handful of the longest queries cannot shrink their candidate lists and fall back to the exact block scan
    (bm25_long_tokens = 0: `bm25_redo_segments` > 0, 3 x the time); by default the long queries' workgroups run the 32-bit body beside the packed
    ones in one launch (bm25_mixed; 0: the whole batch with 32-bit sums): no redo, same lists.  The longest queries and a sample of the others
    against the oracle, ids and scores bit for bit."""
    indptr, doc, tf, lens, _, flat = sparse_data
    idx = host_index(sparse_data, BM25S)
    engine.set_bm25(idx)
    engine.set_doc_meta(N, None, None)
    qs = synth.token_queries(flat, lens, VOCAB, 1024, seed=4000, lengths=synth.REF_QUESTION_LENGTHS)
    ql = np.array([len(q) for q in qs])
    assert ql.max() >= 45 and ql.min() <= 4
    csr = queries_to_csr(qs)
    try:
        engine.reset_stats()
        ids, sc, ln = engine.bm25_topk(*csr, 192)
        assert engine.stat("bm25_redo_segments") == 0 and engine.stat("bm25_mixed_launches") == 1
        engine.set_option("bm25_long_segs", 1)                     # the long queries in one workgroup each
        engine.reset_stats()
        ids2, sc2, ln2 = engine.bm25_topk(*csr, 192)
        assert engine.stat("bm25_redo_segments") == 0 and engine.stat("bm25_mixed_launches") == 1
        assert np.array_equal(ln, ln2) and np.array_equal(ids, ids2) and np.array_equal(sc.view(np.uint64), sc2.view(np.uint64))
        engine.set_option("bm25_long_segs", 4)
        engine.set_option("bm25_mixed", 0)
        engine.reset_stats()
        ids1, sc1, ln1 = engine.bm25_topk(*csr, 192)
        assert engine.stat("bm25_redo_segments") == 0 and engine.stat("bm25_mixed_launches") == 0
        assert np.array_equal(ln, ln1) and np.array_equal(ids, ids1) and np.array_equal(sc.view(np.uint64), sc1.view(np.uint64))
        # fewer queries than workgroup slots: several document-range segments per query, both bodies cutting at the same documents
        few = [int(i) for i in np.argsort(-ql)[:12]] + list(range(5, 1024, 11))[:88]
        csr_few = queries_to_csr([qs[i] for i in few])
        engine.set_option("bm25_mixed", 1)
        engine.reset_stats()
        idf, scf, lnf = engine.bm25_topk(*csr_few, 192)
        assert engine.stat("bm25_mixed_launches") == 1 and engine.stat("bm25_redo_segments") == 0
        assert np.array_equal(lnf, ln[few]) and np.array_equal(idf, ids[few]) and np.array_equal(scf.view(np.uint64), sc[few].view(np.uint64))
        engine.set_option("bm25_long_tokens", 0)
        engine.reset_stats()
        ids0, sc0, ln0 = engine.bm25_topk(*csr, 192)
        print(f"packed shape alone: {engine.stat('bm25_redo_segments')} (query, segment) pairs went to the exact block scan")
        assert np.array_equal(ln, ln0) and np.array_equal(ids, ids0) and np.array_equal(sc.view(np.uint64), sc0.view(np.uint64))
    finally:
        engine.set_option("bm25_long_tokens", 28)
        engine.set_option("bm25_mixed", 1)
        engine.set_option("bm25_long_segs", 4)
    sample = sorted(set(list(np.argsort(-ql)[:10]) + list(range(0, 1024, 73))))
    for b in sample:
        want = sparse_oracle_topk(idx, qs[b], 192)
        assert list(ids[b, :ln[b]]) == [w[0] for w in want], f"query {b} ({ql[b]} tokens): ids differ"
        assert list(sc[b, :ln[b]]) == [w[1] for w in want], f"query {b} ({ql[b]} tokens): scores differ"
