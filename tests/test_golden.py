"""Golden fixture (tests/golden/golden_small.npz, made by tests/golden/make_golden.py from the oracle):
  - CPU: the oracle and the product's host index build still reproduce it bit for bit (regression pin);
  - GPU: the HIP path through the C ABI matches it without re-running the oracle."""
import os

import numpy as np
import pytest

from easyrag_amd.index import BM25S, OKAPI, build_bm25_index

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def gold():
    z = np.load(os.path.join(HERE, "golden", "golden_small.npz"))
    g = {k: z[k] for k in z.files}
    off = np.concatenate([[0], np.cumsum(g["lens"])])
    g["docs"] = [list(map(int, g["flat"][off[i]:off[i + 1]])) for i in range(len(g["lens"]))]
    qo = np.concatenate([[0], np.cumsum(g["q_lens"])])
    g["queries"] = [list(map(int, g["q_flat"][qo[i]:qo[i + 1]])) for i in range(len(g["q_lens"]))]
    return g


def test_oracle_reproduces_golden(gold):
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    fresh = mod.build()
    for k in fresh:
        assert np.array_equal(fresh[k], gold[k]), f"golden field {k} changed"


def test_host_index_payload_sums_match_golden(gold):
    for variant, key in ((OKAPI, "okapi_scores"), (BM25S, "bm25s_scores")):
        idx = build_bm25_index(gold["docs"], variant)
        for b, q in enumerate(gold["queries"]):
            acc = np.zeros(idx.n_docs, idx.payload.dtype)
            for t in idx.tokens_to_ids(q):
                s, e = idx.indptr[t], idx.indptr[t + 1]
                acc[idx.doc_ids[s:e]] = acc[idx.doc_ids[s:e]] + idx.payload[s:e]
            assert np.array_equal(acc, gold[key][b])


@pytest.mark.gpu
def test_gpu_matches_golden(gold, engine):
    from easyrag_amd.engine import queries_to_csr
    n = len(gold["docs"])
    B = len(gold["queries"])
    engine.set_dense(gold["x16"])
    engine.set_doc_meta(n, gold["content_id"], gold["dir_id"])
    ids, sc, ln = engine.dense_topk(gold["q16"], 288)
    assert np.array_equal(ids, gold["dense_ids"]) and np.array_equal(sc, gold["dense_sc"])
    for variant, name in ((OKAPI, "okapi"), (BM25S, "bm25s")):
        idx = build_bm25_index(gold["docs"], variant)
        engine.set_bm25(idx)
        qi, qt = queries_to_csr([idx.tokens_to_ids(q) for q in gold["queries"]])
        for b, q in enumerate(gold["queries"]):
            assert np.array_equal(engine.bm25_scores(idx.tokens_to_ids(q)), gold[f"{name}_scores"][b].astype(np.float64))
        ids, sc, ln = engine.bm25_topk(qi, qt, 192)
        assert np.array_equal(ln, gold[f"{name}_top_len"])
        assert np.array_equal(ids, gold[f"{name}_top_ids"]) and np.array_equal(sc, gold[f"{name}_top_sc"])
        fids, _, _ = engine.bm25_topk(qi, qt, 20, filter_dir=np.full(B, 2, np.int16))
        assert np.array_equal(fids, gold[f"{name}_filt_ids"])
        if variant == OKAPI:
            hid, hsc, hln = engine.hybrid_topk(gold["q16"], qi, qt, k_dense=288, k_sparse=192, K=60, topk=10)
            assert np.array_equal(hid, gold["rrf_ids"]) and np.array_equal(hsc, gold["rrf_sc"])
            sid, ssc, sln = engine.bm25_topk(qi, qt, 192)
            did, dsc, dln = engine.dense_topk(gold["q16"], 288)
            fid, _, _ = engine.fusion(sid, ssc, sln, did, dsc, dln, topk=10)
            assert np.array_equal(fid, gold["fusion_ids"])
