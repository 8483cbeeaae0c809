#!/usr/bin/env python
"""Per-call wall time of the reference-named retriever classes, one query per call -- the reference's only call pattern
(src/main.py:48-52 -> pipeline.py:357-365) -- at the reference's depths (sparse 192, path route 6, dense 288, fusion
256): ctypes + PCIe staging + kernels + NodeWithScore wrapping.  Corpus: synthetic text nodes (default 100k; the Python
node objects, not the GPU, limit the size here).  Usage: python scripts/shim_latency.py [n_nodes]"""
import asyncio
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import WhitespaceTokenizer, make_text_corpus  # noqa: E402
from easyrag_amd.retrievers import BM25Retriever, HipVectorStore, HybridRetriever, QdrantRetriever  # noqa: E402
from easyrag_amd.schema import TextNode  # noqa: E402


class Emb:
    def __init__(self, d):
        self.d = d

    def get_query_embedding(self, text):
        rng = np.random.default_rng(abs(hash(text)) % (2 ** 32))
        v = rng.standard_normal(self.d).astype(np.float32)
        return (v / np.linalg.norm(v)).tolist()


def timed(fn, reps=200):
    fn()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        t.append(time.perf_counter() - t0)
    t = np.sort(np.asarray(t)) * 1e3
    return f"median {t[len(t) // 2]:.3f} ms  p10 {t[len(t) // 10]:.3f}  p90 {t[(9 * len(t)) // 10]:.3f}"


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
    d = 1024
    tok = WhitespaceTokenizer()
    texts = make_text_corpus(n, 20000, seed=3, min_len=20, max_len=90)
    dirs = ["umac", "rcp", "director", "emsplus"]
    nodes = [TextNode(text=t, metadata={"dir": dirs[i % 4], "know_path": f"kp{i % 500} kq{i % 37}"}, id_=f"n{i}")
             for i, t in enumerate(texts)]
    rng = np.random.default_rng(0)
    vecs = rng.standard_normal((n, d)).astype(np.float32)
    vecs /= np.linalg.norm(vecs, axis=1, keepdims=True)
    emb = Emb(d)
    sparse = BM25Retriever.from_defaults(nodes=nodes, tokenizer=tok, similarity_top_k=192, stopwords={""}, embed_type=0,
                                         bm25_type=0)
    path = BM25Retriever.from_defaults(nodes=nodes, tokenizer=tok, similarity_top_k=6, stopwords={""}, embed_type=5,
                                       bm25_type=0, engine=sparse.engine)
    store = HipVectorStore(nodes, vecs, engine=sparse.engine)
    dense = QdrantRetriever(store, emb, similarity_top_k=288)
    hyb = HybridRetriever(dense, sparse, retrieval_type=3, topk=256)
    q = "w5 w90 w333 w17 w2048 w9 w77 w1200"
    qp = "kp7 kq3"
    print(f"shim latency, one query per call, {n} nodes x {d}-d, Okapi (bm25_type 0), filters off unless stated")
    print("BM25Retriever.retrieve (content, top-192)      ", timed(lambda: sparse.retrieve(q)))
    sparse.filter_dict = {"dir": "rcp"}
    print("BM25Retriever.retrieve (content, dir filter)   ", timed(lambda: sparse.retrieve(q)))
    sparse.filter_dict = None
    print("BM25Retriever.retrieve (know_path, top-6)      ", timed(lambda: path.retrieve(qp)))
    print("QdrantRetriever.retrieve (dense top-288)       ", timed(lambda: dense.retrieve(q)))
    print("HybridRetriever.aretrieve (type 3, RRF top-256)", timed(lambda: asyncio.run(hyb.aretrieve(q))))
    a, b, c = sparse.retrieve(q), path.retrieve(qp), dense.retrieve(q)
    print("HybridRetriever.fusion([192, 6])               ", timed(lambda: HybridRetriever.fusion([a, b], topk=256)))
    print("HybridRetriever.reciprocal_rank_fusion([192,288])", timed(lambda: HybridRetriever.reciprocal_rank_fusion([a, c], topk=256)))
    HybridRetriever.fusion_device_min = 0
    print("  ... the same two through the fusion kernels    ", timed(lambda: HybridRetriever.fusion([a, b], topk=256)), "|",
          timed(lambda: HybridRetriever.reciprocal_rank_fusion([a, c], topk=256)))
    docs = [t for t in texts[:12]]
    print("BM25Retriever.get_scores(query, 12 docs)        ", timed(lambda: sparse.get_scores(q, docs), reps=50))


if __name__ == "__main__":
    main()
