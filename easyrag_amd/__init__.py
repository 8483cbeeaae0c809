"""easyrag_amd -- MI355X-native coarse-ranking hot path of EasyRAG (dense cosine top-k, BM25 over CSR
postings, reciprocal-rank fusion) behind the reference's retriever API.

  easyrag_amd.retrievers   QdrantRetriever / BM25Retriever / HybridRetriever (reference names & semantics)
  easyrag_amd.engine       RetrievalEngine: one libeasyrag_hip handle (one GPU replica of the corpus)
  easyrag_amd.index        host-side BM25 index build (tokens -> CSR postings + payloads)
  easyrag_amd.dist         query sharding across ranks + all-gather of the fused top-k (RCCL via torch.distributed)
  easyrag_amd.synth        seeded synthetic corpora / queries of the BASELINE.json shapes

The compute lives in easyrag_amd/csrc/*.hip (C ABI: include/easyrag_hip.h).  Importing the package does not
need a GPU; creating a RetrievalEngine does, and fails loudly without one.
"""
from .index import BM25Index, build_bm25_index, build_bm25_index_from_ids, OKAPI, BM25S  # noqa: F401

__all__ = ["BM25Index", "build_bm25_index", "build_bm25_index_from_ids", "OKAPI", "BM25S",
           "RetrievalEngine", "QdrantRetriever", "BM25Retriever", "HybridRetriever", "HipVectorStore"]


def __getattr__(name):
    # lazy: these import ctypes bindings (and build the library on first use)
    if name == "RetrievalEngine":
        from .engine import RetrievalEngine
        return RetrievalEngine
    if name in ("QdrantRetriever", "BM25Retriever", "HybridRetriever", "HipVectorStore"):
        from . import retrievers
        return getattr(retrievers, name)
    raise AttributeError(name)
