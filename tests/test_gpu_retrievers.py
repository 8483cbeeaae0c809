"""GPU: the reference-named retriever classes (easyrag_amd.retrievers) against the oracle's restatement of
the reference glue, on a small text corpus with metadata filters -- reads like a test of the reference's own
retrievers.py would."""
import asyncio

import numpy as np
import pytest

from conftest import make_text_corpus
from easyrag_amd.retrievers import (BM25Retriever, HipVectorStore, HybridRetriever, QdrantRetriever,
                                    tokenize_and_remove_stopwords)
from easyrag_amd.schema import NodeWithScore, QueryBundle, TextNode
from oracle import (BM25Okapi, BM25SLucene, bm25_filter, dense_exact_topk, fusion, reciprocal_rank_fusion,
                    to_f16_unit)
from oracle.retrievers import Item

pytestmark = pytest.mark.gpu

STOP = {"w0", "w1", ""}


class FakeEmbedder:
    """Deterministic stand-in for GTEEmbedding: unit-norm fp32 list per string (gte_embeddings.py:70-71)."""

    def __init__(self, d=256):
        self.d = d

    def get_query_embedding(self, text):
        rng = np.random.default_rng(abs(hash(text)) % (2 ** 32))
        v = rng.standard_normal(self.d).astype(np.float32)
        return (v / np.linalg.norm(v)).tolist()


@pytest.fixture(scope="module")
def corpus(tokenizer):
    texts = make_text_corpus(1200, 150, seed=7)
    texts[100] = texts[7]                                            # duplicated contents -> shared RRF key
    texts[900] = texts[7]
    dirs = ["umac", "rcp", "director", "emsplus"]
    nodes = [TextNode(text=t, metadata={"dir": dirs[i % 4], "know_path": f"kp {i % 9}"}, id_=f"n{i}")
             for i, t in enumerate(texts)]
    emb = FakeEmbedder()
    vecs = np.asarray([emb.get_query_embedding("doc:" + t + str(i)) for i, t in enumerate(texts)], np.float32)
    return nodes, vecs, emb


@pytest.mark.parametrize("bm25_type", [0, 1])
def test_bm25_retriever_matches_reference_semantics(corpus, tokenizer, bm25_type):
    nodes, _, _ = corpus
    r = BM25Retriever.from_defaults(nodes=nodes, tokenizer=tokenizer, similarity_top_k=20, stopwords=STOP,
                                    embed_type=0, bm25_type=bm25_type)
    toks = [tokenize_and_remove_stopwords(tokenizer, n.get_content(), STOP) for n in nodes]
    ora = BM25Okapi(toks, 1.5, 0.75, 0.25) if bm25_type == 0 else BM25SLucene(1.5, 0.75).index(toks)
    for query in ("w5 w9 w3 w77", "w2 w2 w40", "zzz w1"):
        qt = tokenize_and_remove_stopwords(tokenizer, query, STOP)
        want_scores = ora.get_scores(qt) if qt else np.zeros(len(nodes))
        got_scores = r.get_scores(query) if (qt or bm25_type == 0) else None
        if got_scores is not None:
            assert got_scores.dtype == (np.float64 if bm25_type == 0 else np.float32)
            assert np.array_equal(got_scores, want_scores)
        for fd in (None, {"dir": "rcp"}, {"dir": "nope"}):
            r.filter_dict = fd
            mask = None if fd is None else np.array([n.metadata["dir"] == fd["dir"] for n in nodes])
            want = bm25_filter(want_scores, 20, mask)
            got = r.retrieve(query)
            assert all(isinstance(g, NodeWithScore) for g in got)
            assert [g.node.node_id for g in got] == [nodes[i].node_id for i, _ in want]
            assert [g.score for g in got] == [s for _, s in want]
            if got_scores is not None:
                assert [g.node.node_id for g in r.filter(got_scores)] == [g.node.node_id for g in got]
        r.filter_dict = None
    got = asyncio.run(r.aretrieve(QueryBundle(query_str="w5 w9")))
    assert [g.node.node_id for g in got] == [g.node.node_id for g in r.retrieve("w5 w9")]
    if bm25_type == 1:
        with pytest.raises(IndexError):
            r.get_scores("w0 w1")                                    # all stop-words -> empty token list
    # ad-hoc docs (compressor path, ref retrievers.py:131-147)
    docs = ["w5 w9 w12", "w9 w9", "w3"]
    dt = [tokenize_and_remove_stopwords(tokenizer, d, STOP) for d in docs]
    o2 = BM25Okapi(dt, 1.5, 0.75, 0.25) if bm25_type == 0 else BM25SLucene(1.5, 0.75).index(dt)
    assert np.array_equal(r.get_scores("w9 w3", docs), o2.get_scores(["w9", "w3"]))
    with pytest.raises(ValueError):
        BM25Retriever.from_defaults(tokenizer=tokenizer)


def test_dense_and_hybrid_retrievers(corpus, tokenizer):
    nodes, vecs, emb = corpus
    sparse = BM25Retriever.from_defaults(nodes=nodes, tokenizer=tokenizer, similarity_top_k=192, stopwords=STOP,
                                         bm25_type=0)
    store = HipVectorStore(nodes, vecs, engine=sparse.engine)
    dense = QdrantRetriever(store, emb, similarity_top_k=288)
    x16 = to_f16_unit(vecs)
    toks = [tokenize_and_remove_stopwords(tokenizer, n.get_content(), STOP) for n in nodes]
    ora = BM25Okapi(toks, 1.5, 0.75, 0.25)
    key = {}
    cid = [key.setdefault(n.get_content(), i) for i, n in enumerate(nodes)]
    for query in ("w5 w9 w3", "w14 w2 w2 w60 w8"):
        q16 = to_f16_unit(np.asarray(emb.get_query_embedding(query), np.float32))
        did, dsc = dense_exact_topk(x16, q16, 288)
        got = dense.retrieve(query)
        assert [g.node.node_id for g in got] == [nodes[i].node_id for i in did]
        assert np.max(np.abs(np.array([g.score for g in got]) - dsc)) < 1e-6
        sp = bm25_filter(ora.get_scores(tokenize_and_remove_stopwords(tokenizer, query, STOP)), 192)
        A = [Item(i, cid[i], s) for i, s in sp]
        Bl = [Item(int(i), cid[int(i)], float(s)) for i, s in zip(did, dsc)]
        want = reciprocal_rank_fusion([A, Bl], topk=256)
        for rt, exp in ((1, [nodes[i].node_id for i in did]), (2, [nodes[i].node_id for i, _ in sp]),
                        (3, [nodes[w.idx].node_id for w in want])):
            hy = HybridRetriever(dense, sparse, retrieval_type=rt, topk=256)
            assert [g.node.node_id for g in asyncio.run(hy.aretrieve(query))] == exp   # what the pipeline calls
            assert [g.node.node_id for g in hy.retrieve_batch([query])[0]] == exp
        hy = HybridRetriever(dense, sparse, retrieval_type=3, topk=256)
        got3 = asyncio.run(hy.aretrieve(query))
        assert [g.score for g in got3] == [w.score for w in want]
        # the reference's sync path (retrievers.py:293-305): sparse + dense, de-duplicated by node_id, no fusion
        sync = hy.retrieve(query)
        exp_sync, seen = [], set()
        for i in [i for i, _ in sp] + [int(i) for i in did]:
            if nodes[i].node_id not in seen:
                seen.add(nodes[i].node_id)
                exp_sync.append(nodes[i].node_id)
        assert [g.node.node_id for g in sync] == exp_sync
        # classmethods over already-retrieved lists (pipeline.py:362, 408): on the host objects for inputs of this size
        # (the default), through the library's fusion kernels when forced (fusion_device_min = 0) -- same answer
        for dev_min in (HybridRetriever.fusion_device_min, 0):
            HybridRetriever.fusion_device_min, keep = dev_min, HybridRetriever.fusion_device_min
            try:
                s_nodes, d_nodes = sparse.retrieve(query), dense.retrieve(query)
                rr = HybridRetriever.reciprocal_rank_fusion([s_nodes, d_nodes], topk=10)
                assert [g.node.node_id for g in rr] == [nodes[w.idx].node_id for w in want[:10]]
                assert [g.score for g in rr] == [w.score for w in want[:10]]
                s_nodes, d_nodes = sparse.retrieve(query), dense.retrieve(query)
                wantf = fusion([[Item(i, cid[i], s) for i, s in sp], Bl], topk=256)
                fu = HybridRetriever.fusion([s_nodes, d_nodes], topk=256)
                assert [g.node.node_id for g in fu] == [nodes[w.idx].node_id for w in wantf]
            finally:
                HybridRetriever.fusion_device_min = keep
        # any number of lists, as the reference's loops accept (retrievers.py:245, 261)
        s_nodes, d_nodes = sparse.retrieve(query), dense.retrieve(query)
        three = [s_nodes, d_nodes, s_nodes[:5]]
        want3 = reciprocal_rank_fusion([[Item(i, cid[i], s) for i, s in sp], Bl, [Item(i, cid[i], s) for i, s in sp][:5]], topk=20)
        rr3 = HybridRetriever.reciprocal_rank_fusion(three, topk=20)
        assert [g.node.node_id for g in rr3] == [nodes[w.idx].node_id for w in want3]
        assert [g.score for g in rr3] == [w.score for w in want3]
        assert HybridRetriever.fusion([], topk=5) == [] and HybridRetriever.reciprocal_rank_fusion([[]], topk=5) == []
    # filters pushed down through the hybrid retriever (retrievers.py:278, 283): filter_dict -> sparse route only,
    # filters -> dense route only; every combination against the oracle composition
    query = "w5 w9 w3"
    q16 = to_f16_unit(np.asarray(emb.get_query_embedding(query), np.float32))
    qtok = tokenize_and_remove_stopwords(tokenizer, query, STOP)
    dirs = np.array([n.metadata["dir"] for n in nodes])
    for fdict, filters in (({"dir": "umac"}, {"dir": "umac"}), ({"dir": "umac"}, None), (None, {"dir": "rcp"}),
                           ({"dir": "umac"}, {"dir": "rcp"}), ({"dir": "umac"}, {"know_path": "kp 3"})):
        hy = HybridRetriever(dense, sparse, retrieval_type=3, topk=50)
        hy.filter_dict, hy.filters = fdict, filters
        m_s = None if fdict is None else dirs == fdict["dir"]
        if filters is None:
            m_d = None
        elif "dir" in filters:
            m_d = dirs == filters["dir"]
        else:
            m_d = np.array([n.metadata["know_path"] == filters["know_path"] for n in nodes])
        sp = bm25_filter(ora.get_scores(qtok), 192, m_s)
        did, dsc = dense_exact_topk(x16, q16, 288, m_d)
        want = reciprocal_rank_fusion([[Item(i, cid[i], s) for i, s in sp],
                                       [Item(int(i), cid[int(i)], float(s)) for i, s in zip(did, dsc)]], topk=50)
        out = asyncio.run(hy.aretrieve(query))
        assert [g.node.node_id for g in out] == [nodes[w.idx].node_id for w in want], (fdict, filters)
        assert [g.score for g in out] == [w.score for w in want]
    sparse.filter_dict = None
    dense.filters = None


def test_path_route_composition(corpus, tokenizer):
    """What the shipped yaml executes (retrieval_type 2; pipeline.py:187-210, 357-365): a content BM25 retriever
    (embed_type 2, top 192) and a know_path BM25 retriever (embed_type 5, top 6) over the same nodes, both with the
    query's `dir` filter, merged by HybridRetriever.fusion([content, path]).  The two retrievers share one engine
    (two index slots)."""
    from easyrag_amd.retrievers import get_node_content
    nodes, _, _ = corpus
    for bm25_type in (0, 1):
        content_r = BM25Retriever.from_defaults(nodes=nodes, tokenizer=tokenizer, similarity_top_k=192,
                                                stopwords=STOP, embed_type=2, bm25_type=bm25_type)
        path_r = BM25Retriever.from_defaults(nodes=nodes, tokenizer=tokenizer, similarity_top_k=6, stopwords=STOP,
                                             embed_type=5, bm25_type=bm25_type, engine=content_r.engine)
        assert path_r.engine is content_r.engine and path_r._slot != content_r._slot
        mk = (lambda t: BM25Okapi(t, 1.5, 0.75, 0.25)) if bm25_type == 0 else (lambda t: BM25SLucene(1.5, 0.75).index(t))
        toks_c = [tokenize_and_remove_stopwords(tokenizer, get_node_content(n, 2), STOP) for n in nodes]
        toks_p = [tokenize_and_remove_stopwords(tokenizer, get_node_content(n, 5), STOP) for n in nodes]
        ora_c, ora_p = mk(toks_c), mk(toks_p)
        key = {}
        cid = [key.setdefault(n.get_content(), i) for i, n in enumerate(nodes)]
        dirs = np.array([n.metadata["dir"] for n in nodes])
        for query, d in (("w5 w9 kp 3", "umac"), ("kp 7 w2 w40 w8", "rcp"), ("w77 w3", "emsplus"), ("kp 1", None)):
            fd = None if d is None else {"dir": d}
            content_r.filter_dict = fd
            path_r.filter_dict = fd
            qt = tokenize_and_remove_stopwords(tokenizer, query, STOP)
            mask = None if d is None else dirs == d
            sc_c, sc_p = ora_c.get_scores(qt), ora_p.get_scores(qt)
            want_c, want_p = bm25_filter(sc_c, 192, mask), bm25_filter(sc_p, 6, mask)
            got_c, got_p = content_r.retrieve(query), path_r.retrieve(query)
            assert [(g.node.node_id, g.score) for g in got_c] == [(nodes[i].node_id, s) for i, s in want_c]
            assert [(g.node.node_id, g.score) for g in got_p] == [(nodes[i].node_id, s) for i, s in want_p]
            wantf = fusion([[Item(i, cid[i], s) for i, s in want_c], [Item(i, cid[i], s) for i, s in want_p]], topk=256)
            fu = HybridRetriever.fusion([got_c, got_p], topk=256)
            assert [g.node.node_id for g in fu] == [nodes[w.idx].node_id for w in wantf]
            assert [g.score for g in fu] == [w.score for w in wantf]
            # the content retriever still answers from ITS index after the path retriever was used
            assert [g.node.node_id for g in content_r.retrieve(query)] == [nodes[i].node_id for i, _ in want_c]
        path_r.close()
        content_r.close()


def test_engine_sharing_rules(corpus, tokenizer):
    nodes, vecs, _ = corpus
    r = BM25Retriever.from_defaults(nodes=nodes, tokenizer=tokenizer, similarity_top_k=5, stopwords=STOP)
    with pytest.raises(ValueError):                       # a different node list on the same engine
        BM25Retriever.from_defaults(nodes=nodes[:100], tokenizer=tokenizer, stopwords=STOP, engine=r.engine)
    with pytest.raises(ValueError):                       # wrong embedding dimension at query time
        HipVectorStore(nodes, vecs, engine=r.engine).query_batch(np.zeros((1, 64), np.float32), 3)
    extra = [BM25Retriever.from_defaults(nodes=nodes, tokenizer=tokenizer, stopwords=STOP, embed_type=5, engine=r.engine)
             for _ in range(3)]
    with pytest.raises(RuntimeError):                     # the handle's four index slots are taken
        BM25Retriever.from_defaults(nodes=nodes, tokenizer=tokenizer, stopwords=STOP, engine=r.engine)
    for e in extra:
        e.close()
    assert r.retrieve("w5 w9")
    r.close()


@pytest.mark.parametrize("bm25_type", [0, 1])
def test_bm25_retriever_with_the_native_cutter(bm25_type):
    """f3: a Chinese corpus tokenised, stop-word filtered and numbered inside the library (NativeCutter: erh_text_encode)
    gives the same index and the same results as the same retriever fed by a Python tokenizer object with the same
    cutting rules (the restatement of jieba's HMM=False algorithm) -- and both match the oracle."""
    from easyrag_amd.text import NativeCutter
    from oracle.jieba_cut import DictCutter
    rng = np.random.default_rng(21)
    chars = [chr(c) for c in range(0x4E00, 0x4E00 + 50)]
    words = sorted({"".join(chars[int(i)] for i in rng.integers(0, len(chars), size=int(rng.integers(1, 4)))) for _ in range(300)})
    dict_text = "\n".join(f"{w} {int(rng.integers(1, 900))}" for w in words)
    texts = ["".join(words[int(i)] for i in rng.integers(0, len(words), size=int(rng.integers(5, 40)))) + "。 ok"
             for _ in range(800)]
    nodes = [TextNode(text=t, metadata={"dir": "umac" if i % 2 else "rcp"}, id_=f"c{i}") for i, t in enumerate(texts)]
    stop = {words[0], words[5], "。", ""}
    native, python = NativeCutter(dict_text), DictCutter(dict_text)
    r_nat = BM25Retriever.from_defaults(nodes=nodes, tokenizer=native, similarity_top_k=15, stopwords=stop, bm25_type=bm25_type)
    r_py = BM25Retriever.from_defaults(nodes=nodes, tokenizer=python, similarity_top_k=15, stopwords=stop, bm25_type=bm25_type,
                                       engine=r_nat.engine)
    assert r_nat._corpus is None and r_py._corpus is not None          # the native path kept no Python token lists
    toks = [tokenize_and_remove_stopwords(python, t, stop) for t in texts]
    ora = BM25Okapi(toks, 1.5, 0.75, 0.25) if bm25_type == 0 else BM25SLucene(1.5, 0.75).index(toks)
    for query in (texts[3][:9], words[7] + words[7] + words[20], words[0] + "未登录"):
        qt = tokenize_and_remove_stopwords(python, query, stop)
        if not qt and bm25_type == 1:
            continue
        want = bm25_filter(ora.get_scores(qt) if qt else np.zeros(len(nodes)), 15)
        for r in (r_nat, r_py):
            got = r.retrieve(query)
            assert [g.node.node_id for g in got] == [nodes[i].node_id for i, _ in want]
            assert [g.score for g in got] == [s for _, s in want]
    r_py.close()
    r_nat.close()


def test_vector_store_restart_round_trip(corpus, tmp_path):
    """The restart path on the device (ref pipeline.py:138-141: ingestion is skipped when the collection is populated): a store
    built from embeddings, one rebuilt by scrolling a Qdrant-shaped client (points in another order, other ids: joined on the
    text), and one loaded from the file the first one saved must answer every query with the same nodes and the same scores, bit
    for bit -- the saved rows are the fp16 rows the kernels score, read back through erh_get_dense_rows (the golden-ratio row
    placement undone) and uploaded as they are."""
    import json
    from easyrag_amd.engine import RetrievalEngine
    nodes, vecs, emb = corpus
    first = HipVectorStore(nodes, vecs, engine=RetrievalEngine(0))
    assert np.array_equal(first.engine.get_dense_rows(0, len(nodes)), to_f16_unit(vecs))         # rows come back in the caller's order
    assert np.array_equal(first.engine.get_dense_rows(1000, 200), to_f16_unit(vecs)[1000:])
    path = first.save(tmp_path / "store")

    class Point:
        def __init__(self, id, vector, payload): self.id, self.vector, self.payload = id, vector, payload

    class Client:
        def __init__(self, pts): self.pts = pts
        def scroll(self, collection_name, limit, offset, with_payload, with_vectors):
            s = int(offset or 0)
            return self.pts[s:s + limit], (s + limit if s + limit < len(self.pts) else None)

    order = np.random.default_rng(3).permutation(len(nodes))
    # (texts 7 / 100 / 900 are equal and carry no file_path: equal texts are handed out first come first served, so the scroll
    # keeps those three in node order -- any other order is an equally valid join, with the three vectors permuted)
    where = sorted(int(np.nonzero(order == j)[0][0]) for j in (7, 100, 900))
    order[where] = (7, 100, 900)
    pts = [Point(f"uuid-{j}", vecs[j].tolist(), {"dir": nodes[j].metadata["dir"],
                                                "_node_content": json.dumps({"text": nodes[j].text, "metadata": nodes[j].metadata})})
           for j in order]
    second = HipVectorStore.from_qdrant(Client(pts), "aiops24", nodes, engine=RetrievalEngine(0), batch_size=500)
    third = HipVectorStore.load(path, nodes, engine=RetrievalEngine(0))
    try:
        qs = np.asarray([emb.get_query_embedding(f"question {i}") for i in range(9)], np.float32)
        for filters in (None, {"dir": "rcp"}):
            want = first.query_batch(qs, 50, filters)
            for other in (second, third):
                got = other.query_batch(qs, 50, filters)
                assert np.array_equal(got[0], want[0]) and np.array_equal(got[2], want[2])
                assert np.array_equal(got[1].view(np.uint64), want[1].view(np.uint64))
        x16 = to_f16_unit(vecs)
        oid, osc = dense_exact_topk(x16, to_f16_unit(qs[:1])[0], 50)
        got = third.query(qs[0], 50)
        assert [n.node_id for n in got[0]] == [nodes[i].node_id for i in oid] and got[1] == [float(s) for s in osc]
    finally:
        for s in (first, second, third):
            s.engine.close()


def test_hybrid_retriever_on_a_corpus_loaded_dir_by_dir(tokenizer):
    """The reference's loader walks the document directories one after the other (ingestion.py:79-87), so the nodes of a dir are consecutive;
    its pipeline then sets `filters` / `filter_dict` to the question's dir on every query (pipeline.py:301-312).  On such a corpus the dense
    route answers from the dir's block copy (dense_dir_blocks; block minimum lowered to this corpus' size): same nodes and scores as the
    oracle composition, for the single-query call the pipeline makes and for a batch."""
    texts = make_text_corpus(1500, 150, seed=17)
    sizes = {"umac": 600, "rcp": 500, "director": 250, "emsplus": 150}
    dirs = np.repeat(list(sizes), list(sizes.values()))
    nodes = [TextNode(text=t, metadata={"dir": str(dirs[i])}, id_=f"n{i}") for i, t in enumerate(texts)]
    emb = FakeEmbedder()
    vecs = np.asarray([emb.get_query_embedding("doc:" + t + str(i)) for i, t in enumerate(texts)], np.float32)
    sparse = BM25Retriever.from_defaults(nodes=nodes, tokenizer=tokenizer, similarity_top_k=192, stopwords=STOP, bm25_type=0)
    store = HipVectorStore(nodes, vecs, engine=sparse.engine)
    dense = QdrantRetriever(store, emb, similarity_top_k=288)
    eng = sparse.engine
    eng.set_option("dense_dir_block_min_rows", 1)
    eng.set_option("dense_dir_blocks", 2)
    try:
        x16 = to_f16_unit(vecs)
        toks = [tokenize_and_remove_stopwords(tokenizer, n.get_content(), STOP) for n in nodes]
        ora = BM25Okapi(toks, 1.5, 0.75, 0.25)
        key = {}
        cid = [key.setdefault(n.get_content(), i) for i, n in enumerate(nodes)]
        hy = HybridRetriever(dense, sparse, retrieval_type=3, topk=50)
        queries = ["w5 w9 w3", "w14 w2 w2 w60 w8", "w7 w30"]
        for dname in ("rcp", "emsplus"):
            hy.filter_dict, hy.filters = {"dir": dname}, {"dir": dname}
            mask = dirs == dname
            want_all = []
            for query in queries:
                q16 = to_f16_unit(np.asarray(emb.get_query_embedding(query), np.float32))
                sp = bm25_filter(ora.get_scores(tokenize_and_remove_stopwords(tokenizer, query, STOP)), 192, mask)
                did, dsc = dense_exact_topk(x16, q16, 288, mask)
                want_all.append(reciprocal_rank_fusion([[Item(i, cid[i], s) for i, s in sp],
                                                        [Item(int(i), cid[int(i)], float(s)) for i, s in zip(did, dsc)]], topk=50))
            eng.reset_stats()
            out = asyncio.run(hy.aretrieve(queries[0]))
            assert eng.stat("dense_block_groups") == 1
            assert [g.node.node_id for g in out] == [nodes[w.idx].node_id for w in want_all[0]]
            assert [g.score for g in out] == [w.score for w in want_all[0]]
            assert all(g.node.metadata["dir"] == dname for g in out)
            for got, want in zip(hy.retrieve_batch(queries), want_all):
                assert [g.node.node_id for g in got] == [nodes[w.idx].node_id for w in want]
                assert [g.score for g in got] == [w.score for w in want]
            # the dense retriever on its own, as QdrantRetriever._aretrieve with `filters` set (retrievers.py:37-52)
            dense.filters = {"dir": dname}
            q16 = to_f16_unit(np.asarray(emb.get_query_embedding(queries[1]), np.float32))
            did, dsc = dense_exact_topk(x16, q16, 288, mask)
            got = dense.retrieve(queries[1])
            assert [g.node.node_id for g in got] == [nodes[i].node_id for i in did]
    finally:
        eng.set_option("dense_dir_block_min_rows", 4096)
        eng.set_option("dense_dir_blocks", 1)
        sparse.filter_dict = None
        dense.filters = None
