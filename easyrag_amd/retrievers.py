"""Drop-in retrievers with the reference's class names, constructor kwargs, methods and return
conventions, backed by libeasyrag_hip.so (MI355X).  Mirrors
/root/reference/src/easyrag/custom/retrievers.py:

  QdrantRetriever(vector_store, embed_model, similarity_top_k=2, filters=None)      ref :23-69
  tokenize_and_remove_stopwords(tokenizer, text, stopwords)                          ref :72-76
  BM25Retriever(nodes, tokenizer, similarity_top_k, ..., stopwords, embed_type, bm25_type)
      .get_scores(query, docs=None) / .from_defaults(...) / .filter(scores) / ._retrieve   ref :80-220
  HybridRetriever(dense_retriever, sparse_retriever, retrieval_type=1, topk=256)
      .fusion(lists, topk) / .reciprocal_rank_fusion(lists, K, topk) / ._aretrieve     ref :223-305

What differs, deliberately: nodes are addressed by their integer position in the caller's node list
(the pipeline keeps the same map, pipeline.py:221-223); the Qdrant collection is replaced by
HipVectorStore (the fp16 chunk matrix resident in HBM); equal scores are ordered by node index
(numpy's argsort()[::-1] tie order is implementation-defined); and every retriever also offers a
batched entry point (retrieve_batch) because the GPU path is built for batches.  Scores, cut-offs,
filters, fusion keys and list order follow the reference -- including HybridRetriever's two faces:
`aretrieve` (what the pipeline calls) selects the route and fuses with RRF, `retrieve` is the reference's
unmaintained sync path (sparse + dense concatenated, de-duplicated by node_id, no fusion, no filter push-down).

Engine sharing: retrievers built over the same node list may share one RetrievalEngine (one GPU replica of
the corpus): the vector store holds the chunk matrix, every BM25Retriever takes its own BM25 index slot of the
engine (4 per engine), and the node bookkeeping (content ids, filter classes) lives on the engine.  Passing an
engine that already serves a DIFFERENT node list raises.

No retrieval arithmetic happens in this file except BM25Retriever.filter(scores), which the reference
exposes as a host-side helper over a caller-supplied score vector; _retrieve does not use it.
"""
from __future__ import annotations

import asyncio
import logging
import threading
from typing import Any, Callable, Dict, Hashable, List, Optional, Sequence

import numpy as np

from . import _lib
from .engine import RetrievalEngine, queries_to_csr
from .index import BM25Index, build_bm25_index, vocab_ids
from .schema import NodeWithScore, QueryBundle, TextNode

logger = logging.getLogger(__name__)

DEFAULT_SIMILARITY_TOP_K = 2   # llama_index.core.constants.DEFAULT_SIMILARITY_TOP_K


# ---------------------------------------------------------------------------------------------------
# text variants and tokenisation (host side, as in the reference)
def _join_overlapping(head: str, tail: str) -> str:
    """`head` followed by `tail` with the longest suffix-of-head == prefix-of-tail written once (ref ingestion.py:20-31)."""
    keep = 0
    for i in range(1, min(len(head), len(tail)) + 1):
        if head[-i:] == tail[:i]:
            keep = i
    return head + tail[keep:]


def _previous_node_id(node):
    """node_id of the PREVIOUS relationship of a NodeWithScore's node (llama_index keys the dict by the
    NodeRelationship enum, whose value is the string "2"; the stand-in nodes may use either or the name)."""
    rel = node.node.relationships
    try:
        from llama_index.core.schema import NodeRelationship  # type: ignore
        return rel[NodeRelationship.PREVIOUS].node_id
    except ImportError:
        for key in ("PREVIOUS", "2", 2):
            if key in rel:
                return rel[key].node_id
        raise KeyError("PREVIOUS")


def _table_with_header(node, text: str, nodes, nodeid2idx) -> str:
    """embed_type 6 (ref ingestion.py:36-57): a chunk that looks like the body of a markdown table (>= 5 '|', no '---'
    rule) is joined with up to three preceding chunks until one of them carries the table's header rule; the result is the
    header line + everything from the rule on.  Without a rule within three chunks the text stays as it was."""
    if text.count("|") < 5 or text.count("---") != 0:
        return text
    cur, found = text, False
    for _ in range(3):
        # (the reference never advances past the first PREVIOUS link: it re-reads node.node's relationship on every
        # turn, so the same predecessor is merged up to three times -- kept, the merge is idempotent after the first)
        pre_text = nodes[nodeid2idx[_previous_node_id(node)]].text
        cur = _join_overlapping(pre_text, cur)
        if pre_text.count("---") >= 2:
            found = True
            break
    if not found:
        return text
    at = cur.index("---")
    return cur[:at].strip().split("\n")[-1] + cur[at:]


def get_node_content(node, embed_type: int = 0, nodes=None, nodeid2idx=None) -> str:
    """Text fed to a route, by embed_type (ref: src/easyrag/pipeline/ingestion.py:34-76).
    0 raw text; 1 '###\\n<file_path>\\n\\n<text>'; 2 same with know_path; 3 image captions expanded;
    4 file_path only; 5 know_path only; 6 = 3 after the table-header merge over the PREVIOUS relationship, which
    needs a NodeWithScore plus the pipeline's `nodes` / `nodeid2idx` (as the reference does: with a bare TextNode a
    table-like chunk raises AttributeError there and here)."""
    text = node.get_content()
    meta = node.metadata
    if embed_type == 6:
        text = _table_with_header(node, text, nodes, nodeid2idx)
    if embed_type == 1 and "file_path" in meta:
        return "###\n" + meta["file_path"] + "\n\n" + text
    if embed_type == 2 and "know_path" in meta:
        return "###\n" + meta["know_path"] + "\n\n" + text
    if embed_type in (3, 6):
        for img in meta.get("imgobjs") or []:
            text = text.replace(f"{img['cap']} {img['title']}\n", f"{img['cap']}.{img['title']}:{img['content']}\n")
        return text
    if embed_type == 4:
        return meta.get("file_path", "")
    if embed_type == 5:
        return meta.get("know_path", "")
    return text


def tokenize_and_remove_stopwords(tokenizer, text, stopwords):
    """``tokenizer.cut(text)`` minus stop-words and single spaces (ref retrievers.py:72-76)."""
    return [w for w in tokenizer.cut(text) if w not in stopwords and w != " "]


def _unit_f16(v, slab_rows: int = 65536) -> np.ndarray:
    """L2-normalise rows in float64, round once to float16: the query / chunk representation the kernels score.
    (Qdrant normalises both sides for Distance.COSINE; doing it here pins the fp16 rounding on the host.)
    Rows go through float64 in slabs, so a 1M x 1024 matrix needs ~0.5 GB of temporaries, not 24."""
    v = np.asarray(v)
    if v.ndim == 1:
        v = v[None, :]
    out = np.empty(v.shape, np.float16)
    for r0 in range(0, v.shape[0], slab_rows):
        v64 = v[r0:r0 + slab_rows].astype(np.float32).astype(np.float64)
        n = np.sqrt(np.sum(v64 * v64, axis=-1, keepdims=True))
        n = np.where(n != 0.0, n, 1e-12)
        out[r0:r0 + slab_rows] = (v64 / n).astype(np.float16)
    return out


def _as_bundle(q) -> QueryBundle:
    return q if isinstance(q, QueryBundle) else QueryBundle(query_str=str(q))


class _MetaClasses:
    """Per-document class ids for an equality filter over a fixed set of metadata keys."""

    def __init__(self, nodes, keys: Sequence[str]):
        self.keys = tuple(keys)
        self.values: Dict[tuple, int] = {}
        ids = np.empty(len(nodes), np.int16)
        for i, n in enumerate(nodes):
            v = tuple(n.metadata.get(k) for k in self.keys)
            j = self.values.get(v)
            if j is None:
                j = len(self.values)
                if j >= 32000:
                    raise ValueError("too many distinct filter values")
                self.values[v] = j
            ids[i] = j
        self.ids = ids

    def class_of(self, filter_dict: Dict[str, Any]) -> int:
        v = tuple(filter_dict.get(k) for k in self.keys)
        return self.values.get(v, 32767)          # 32767 matches no document


def _filter_to_dict(filters) -> Optional[Dict[str, Any]]:
    """Accept None, a {key: value} dict, or a qdrant-style Filter(must=[FieldCondition(key, match=MatchValue(value))])
    (what build_qdrant_filters returns, ref ingestion.py:207-216)."""
    if filters is None:
        return None
    if isinstance(filters, dict):
        return dict(filters) or None
    must = getattr(filters, "must", None)
    if must:
        out = {}
        for cond in must:
            out[getattr(cond, "key")] = getattr(getattr(cond, "match"), "value")
        return out or None
    raise TypeError(f"unsupported filter object: {filters!r}")


class _FilteredCorpus:
    """Shared bookkeeping: one engine, one node list, the content ids and the active filter column."""

    def __init__(self, nodes, engine: Optional[RetrievalEngine]):
        self.nodes = list(nodes)
        self.engine = engine if engine is not None else RetrievalEngine()
        self._classes: Optional[_MetaClasses] = None
        seen: Dict[Hashable, int] = {}
        cid = np.empty(len(self.nodes), np.int32)
        for i, n in enumerate(self.nodes):
            cid[i] = seen.setdefault(n.get_content(), i)      # smallest index with identical text
        self.content_id = cid
        self._push_meta()

    def _push_meta(self):
        self.engine.set_doc_meta(len(self.nodes), self.content_id, None if self._classes is None else self._classes.ids)

    def filter_class(self, filter_dict: Optional[Dict[str, Any]]) -> int:
        """Class id of `filter_dict` in the column over ITS key set.  The device holds one column at a time; the host tables
        are kept per key set, so alternating between two key sets (filter_dict on one route, filters on the other with
        different keys) costs the 2-bytes-per-node upload but not the walk over all nodes' metadata."""
        if not filter_dict:
            return -1
        keys = tuple(sorted(filter_dict))
        if self._classes is None or self._classes.keys != keys:
            cache = self.__dict__.setdefault("_class_tables", {})
            if keys not in cache:
                cache[keys] = _MetaClasses(self.nodes, keys)
            self._classes = cache[keys]
            self._push_meta()
        return self._classes.class_of(filter_dict)

    def filter_column(self, filter_dicts: Sequence[Optional[Dict[str, Any]]]) -> Optional[np.ndarray]:
        """int16[B] class column for a batch whose queries carry their OWN filters (None / {} = unfiltered: -1), or None
        when no query is filtered.  All filters of one column use one metadata key set (the device holds one class
        column at a time); `_by_key_set` splits batches that mix key sets."""
        keys = {tuple(sorted(fd)) for fd in filter_dicts if fd}
        if not keys:
            return None
        if len(keys) > 1:
            raise ValueError("the filters of one device batch must share their metadata keys")
        col = np.full(len(filter_dicts), -1, np.int16)
        for b, fd in enumerate(filter_dicts):
            if fd:
                col[b] = self.filter_class(fd)
        return col


_UNSET = object()      # "argument not given": the retriever's scalar attribute (filters / filter_dict) applies to every query


def _per_query(value, B: int, to_dict) -> List[Optional[Dict[str, Any]]]:
    """One filter per query.  `value` is the reference's scalar knob (None, a {key: value} dict, a qdrant Filter: applies
    to every query, as `retriever.filters = ...` / `.filter_dict = ...` do, ref pipeline.py:333-341) or a list / tuple
    parallel to the queries (the reference's evaluation loop sets a new filter before every question, main.py:48-52 ->
    pipeline.py:301-312: a batch of its questions carries one filter EACH)."""
    if isinstance(value, (list, tuple)):
        if len(value) != B:
            raise ValueError(f"per-query filters: {len(value)} filters for {B} queries")
        return [to_dict(v) for v in value]
    one = to_dict(value)
    return [one] * B


def _by_key_set(dicts: Sequence[Optional[Dict[str, Any]]]) -> List[List[int]]:
    """Query positions grouped so that each group's filters share one metadata key set (unfiltered queries join the
    first group).  The usual batch -- every filter on `dir` -- is ONE group."""
    groups: Dict[tuple, List[int]] = {}
    free: List[int] = []
    for b, fd in enumerate(dicts):
        if fd:
            groups.setdefault(tuple(sorted(fd)), []).append(b)
        else:
            free.append(b)
    out = list(groups.values())
    if not out:
        return [free] if free else []
    out[0] = sorted(out[0] + free)
    return out


def _corpus_for(nodes, engine: Optional[RetrievalEngine]) -> _FilteredCorpus:
    """Retrievers built over the same engine share one _FilteredCorpus (and so one metadata upload).  The
    bookkeeping lives on the engine object (no process-global cache pinning engines); an engine serves ONE node
    list -- the same node objects in the same order -- and anything else is an error, not a silent reuse."""
    if engine is not None and engine.corpus is not None:
        c = engine.corpus
        nodes = list(nodes)
        if len(c.nodes) != len(nodes) or any(a is not b for a, b in zip(c.nodes, nodes)):
            raise ValueError("this RetrievalEngine already serves a different node list; use one engine per corpus")
        return c
    c = _FilteredCorpus(nodes, engine)
    c.engine.corpus = c
    return c


# ---------------------------------------------------------------------------------------------------
class HipVectorStore:
    """Stand-in for the Qdrant collection (Distance.COSINE, ref ingestion.py:155-191): the chunk embeddings
    live as one fp16 matrix in HBM.  `embeddings` is [N, d] (numpy or torch, fp16/fp32), row i belongs to
    nodes[i]; normalize=True applies Qdrant's insert-time L2 normalisation."""

    def __init__(self, nodes, embeddings, engine: Optional[RetrievalEngine] = None, normalize: bool = True):
        self.corpus = _corpus_for(nodes, engine)
        self.engine = self.corpus.engine
        is_f16 = str(getattr(embeddings, "dtype", "")).endswith("float16")
        if normalize and not is_f16 and isinstance(embeddings, np.ndarray):
            self.engine.set_dense(_unit_f16(embeddings))             # host: fp16 rounding of the unit rows is pinned
        else:
            self.engine.set_dense(embeddings, normalize=normalize and not is_f16)   # device tensors: normalise on the GPU

    @property
    def nodes(self):
        return self.corpus.nodes

    # -- restart path ---------------------------------------------------------------------------------------------------
    # The reference embeds the corpus ONCE: when the Qdrant collection is already populated it skips ingestion
    # (ref pipeline.py:138-141, `collection_info.points_count == 0`), so after a restart the embeddings exist only in
    # Qdrant.  Two ways to come back without re-embedding: read them out of that collection (from_qdrant / afrom_qdrant),
    # or keep the fp16 matrix this store scores with in a file of its own (save / load).  These are loaders, not a storage
    # engine: one pass, then everything is resident in HBM as before.
    @staticmethod
    def _point_fields(point):
        """(id, vector, text, metadata) of one scrolled Qdrant point as llama-index-vector-stores-qdrant==0.2.0 writes it
        (ref requirements.txt:49; UNPINNED offline: point id = node_id, payload = flat metadata + `_node_content`, the
        node's JSON).  Named-vector collections hand back {name: vector}: the single entry is taken."""
        import json
        payload = getattr(point, "payload", None) or {}
        vec = getattr(point, "vector", None)
        if isinstance(vec, dict):
            if len(vec) != 1:
                raise ValueError("point carries several named vectors; pass vectors through a client wrapper that selects one")
            vec = next(iter(vec.values()))
        text, meta = payload.get("text"), None
        content = payload.get("_node_content")
        if content:
            node_json = json.loads(content) if isinstance(content, str) else dict(content)
            text = node_json.get("text", text)
            meta = node_json.get("metadata")
        if meta is None:
            meta = {k: v for k, v in payload.items() if not k.startswith("_") and k not in ("text", "document_id", "doc_id", "ref_doc_id")}
        return getattr(point, "id", None), vec, text, meta

    class _Scroll:
        """What a scroll over the collection leaves behind: per point its id, text and metadata, and the vectors as one float32
        block per scrolled batch -- a point's vector never stays around as a Python list (3584 floats are ~115 KB of objects:
        ~10 GB for 100k chunks), and a point without its vector is reported by the batch that carries it."""

        def __init__(self):
            self.meta: List[tuple] = []
            self.blocks: List[np.ndarray] = []
            self.d: Optional[int] = None

        def add(self, batch) -> None:
            rows = []
            for p in batch:
                pid, vec, text, meta = HipVectorStore._point_fields(p)
                if vec is None:
                    raise ValueError(f"point {pid!r} came back without its vector (scroll with with_vectors=True)")
                if self.d is None:
                    self.d = len(vec)
                if len(vec) != self.d:
                    raise ValueError("points of different vector sizes in one collection")
                rows.append(vec)
                self.meta.append((pid, text, meta))
            if rows:
                self.blocks.append(np.asarray(rows, dtype=np.float32))

        def row(self, j: int, starts) -> np.ndarray:
            b = int(np.searchsorted(starts, j, side="right")) - 1
            return self.blocks[b][j - starts[b]]

    @classmethod
    def _assemble(cls, scroll: "HipVectorStore._Scroll", nodes, engine, normalize):
        """Scrolled points -> (nodes, [N, d] float32 rows aligned with them).  With `nodes` given every node must find its
        vector: by point id == node_id when all ids match, otherwise by text content -- the key the reference itself joins
        the two routes on (RRF / fusion key = get_content(), ref retrievers.py:245,263): its sparse-route nodes come from a
        second run of the splitter (pipeline.py:160-167) and share no ids with the ingested points.  Among nodes with equal
        text a point with equal `file_path` is preferred, then first come first served.  Without `nodes` the node list is
        rebuilt from the payloads in scroll order (what the reference's QdrantRetriever returns)."""
        fields = scroll.meta
        if not fields:
            raise ValueError("the collection is empty (points_count == 0): run the ingestion first, as the reference does")
        d = int(scroll.d)
        if nodes is None:
            nodes = [TextNode(text=t or "", metadata=dict(m or {}), id_=str(pid)) for pid, t, m in fields]
            return cls(nodes, np.concatenate(scroll.blocks) if len(scroll.blocks) > 1 else scroll.blocks[0], engine=engine, normalize=normalize)
        nodes = list(nodes)
        by_id = {str(pid): j for j, (pid, _, _) in enumerate(fields)}
        if len(by_id) == len(fields) and all(str(n.node_id) in by_id for n in nodes):
            order = [by_id[str(n.node_id)] for n in nodes]
        else:
            by_text: Dict[Any, List[int]] = {}
            for j, (_, t, _) in enumerate(fields):
                by_text.setdefault(t, []).append(j)
            order, missing = [], 0
            for n in nodes:
                bucket = by_text.get(n.get_content())
                if not bucket:
                    missing += 1
                    order.append(-1)
                    continue
                fp = n.metadata.get("file_path")
                pick = next((j for j in bucket if (fields[j][2] or {}).get("file_path") == fp), bucket[0])
                bucket.remove(pick)
                order.append(pick)
            if missing:
                raise ValueError(f"{missing} of {len(nodes)} nodes have no point with their text in the collection "
                                 f"({len(fields)} points): the collection was built from a different corpus / splitter")
        starts = np.cumsum([0] + [blk.shape[0] for blk in scroll.blocks])[:-1]
        emb = np.empty((len(nodes), d), np.float32)                   # the one full-size array; filled block row by block row
        for i, j in enumerate(order):
            emb[i] = scroll.row(j, starts)
        return cls(nodes, emb, engine=engine, normalize=normalize)

    @classmethod
    def from_qdrant(cls, client, collection_name: str, nodes=None, engine: Optional[RetrievalEngine] = None,
                    batch_size: int = 1024, normalize: bool = True) -> "HipVectorStore":
        """The chunk matrix out of an already populated Qdrant collection (a synchronous `QdrantClient`, or anything with its
        `scroll(collection_name=, limit=, offset=, with_payload=, with_vectors=) -> (points, next_offset)`)."""
        scroll, offset = cls._Scroll(), None
        while True:
            res = client.scroll(collection_name=collection_name, limit=batch_size, offset=offset, with_payload=True,
                                with_vectors=True)
            if hasattr(res, "__await__"):
                if hasattr(res, "close"):
                    res.close()
                raise TypeError("this client's scroll() is a coroutine (AsyncQdrantClient): use `await HipVectorStore.afrom_qdrant(...)`")
            batch, offset = res
            scroll.add(batch)
            if offset is None or not batch:
                break
        return cls._assemble(scroll, nodes, engine, normalize)

    @classmethod
    async def afrom_qdrant(cls, client, collection_name: str, nodes=None, engine: Optional[RetrievalEngine] = None,
                           batch_size: int = 1024, normalize: bool = True) -> "HipVectorStore":
        """from_qdrant for the `AsyncQdrantClient` the reference's pipeline holds (ref ingestion.py:163-169)."""
        scroll, offset = cls._Scroll(), None
        while True:
            batch, offset = await client.scroll(collection_name=collection_name, limit=batch_size, offset=offset,
                                                with_payload=True, with_vectors=True)
            scroll.add(batch)
            if offset is None or not batch:
                break
        return cls._assemble(scroll, nodes, engine, normalize)

    @staticmethod
    def _fingerprint(nodes) -> str:
        import hashlib
        h = hashlib.sha256()
        for n in nodes:
            t = n.get_content().encode("utf-8", "surrogatepass")
            h.update(len(t).to_bytes(8, "little"))
            h.update(t)
        return h.hexdigest()

    @staticmethod
    def _paths(path):
        path = str(path)
        if not path.endswith(".npy"):
            path += ".npy"
        return path, path + ".meta.json"

    def save(self, path) -> str:
        """Write the resident matrix -- the unit-norm fp16 rows the kernels score, in node order -- as `<path>.npy` plus a
        small `<path>.npy.meta.json` (shape and a fingerprint of the node texts, checked by load).  2 bytes per element."""
        import json
        npy, meta = self._paths(path)
        eng = self.engine
        n, d = int(eng.n_dense), int(eng.d)
        if n != len(self.corpus.nodes):
            raise ValueError("the engine's chunk matrix does not belong to this store's nodes")
        mm = np.lib.format.open_memmap(npy, mode="w+", dtype=np.float16, shape=(n, d))
        slab = max(1, (64 << 20) // (2 * d))
        for r0 in range(0, n, slab):
            rows = min(slab, n - r0)
            eng.get_dense_rows(r0, rows, out=mm[r0:r0 + rows])
        mm.flush()
        del mm
        with open(meta, "w") as f:
            json.dump({"format": "easyrag_amd.HipVectorStore/1", "n": n, "d": d, "dtype": "float16", "unit_norm": True,
                       "nodes_sha256": self._fingerprint(self.corpus.nodes)}, f)
        return npy

    @classmethod
    def load(cls, path, nodes, engine: Optional[RetrievalEngine] = None, check: bool = True) -> "HipVectorStore":
        """The store of a previous process over the same nodes: rows go to the device as they were saved (no re-normalisation,
        no re-rounding), so every result is bit-identical to the process that saved them."""
        import json
        npy, meta = cls._paths(path)
        with open(meta) as f:
            info = json.load(f)
        nodes = list(nodes)
        if info.get("format") != "easyrag_amd.HipVectorStore/1" or info.get("dtype") != "float16":
            raise ValueError(f"{meta}: not a HipVectorStore file")
        if int(info["n"]) != len(nodes):
            raise ValueError(f"{npy} holds {info['n']} rows, the node list has {len(nodes)}")
        if check and info.get("nodes_sha256") != cls._fingerprint(nodes):
            raise ValueError(f"{npy} was saved for different node texts (fingerprint mismatch); pass check=False to override")
        x = np.load(npy, mmap_mode="r")
        if x.dtype != np.float16 or x.shape != (int(info["n"]), int(info["d"])):
            raise ValueError(f"{npy}: shape / dtype differ from its meta file")
        return cls(nodes, x, engine=engine, normalize=False)

    def query_batch(self, query_embeddings, similarity_top_k: int, filters=None, mode: int = _lib.ERH_DENSE_EXACT):
        """`filters`: one filter for the whole batch (None, dict, qdrant Filter) or a list with one filter per query."""
        q = _unit_f16(query_embeddings)
        B, k = q.shape[0], int(similarity_top_k)
        dicts = _per_query(filters, B, _filter_to_dict)
        groups = _by_key_set(dicts)
        if len(groups) <= 1:
            return self.engine.dense_topk(q, k, filter_dir=self.corpus.filter_column(dicts), mode=mode)
        ids, sc, ln = np.full((B, k), -1, np.int32), np.zeros((B, k), np.float64), np.zeros(B, np.int32)
        for g in groups:                              # filters over different metadata keys: one device column at a time
            gi, gs, gl = self.engine.dense_topk(q[g], k, filter_dir=self.corpus.filter_column([dicts[b] for b in g]), mode=mode)
            ids[g], sc[g], ln[g] = gi, gs, gl
        return ids, sc, ln

    def query(self, query_embedding, similarity_top_k: int, filters=None):
        ids, sc, ln = self.query_batch(query_embedding, similarity_top_k, filters)
        n = int(ln[0])
        return [self.corpus.nodes[i] for i in ids[0, :n]], [float(s) for s in sc[0, :n]]


try:  # the reference's base class (ref retrievers.py:7,23,80,223) whenever llama_index is installed
    from llama_index.core.base.base_retriever import BaseRetriever as _LlamaBaseRetriever  # type: ignore
except Exception:  # pragma: no cover - the build container has no llama_index
    _LlamaBaseRetriever = None

if _LlamaBaseRetriever is not None:
    class _RetrieverBase(_LlamaBaseRetriever):
        """llama_index's BaseRetriever itself: `retrieve` / `aretrieve` (callback events, recursive retrieval over
        `objects`) come from it, so `AutoMergingRetriever(sparse_retriever, ...)` (ref pipeline.py:212-217) and RETRIEVE
        callbacks work as with the reference's classes; this file supplies `_retrieve` / `_aretrieve`."""

        def _init_base(self, callback_manager=None, object_map=None, objects=None, verbose: bool = False) -> None:
            _LlamaBaseRetriever.__init__(self, callback_manager=callback_manager, object_map=object_map,
                                         objects=objects, verbose=verbose)
else:
    class _RetrieverBase:
        """Fallback without llama_index: the slice of BaseRetriever the pipeline uses -- retrieve / aretrieve accept str
        or QueryBundle; the ctor arguments of the real base class are kept as attributes."""

        def _init_base(self, callback_manager=None, object_map=None, objects=None, verbose: bool = False) -> None:
            self.callback_manager = callback_manager
            self.object_map = object_map or {}
            self.objects = objects
            self._verbose = verbose

        def retrieve(self, str_or_query_bundle) -> List[NodeWithScore]:
            return self._retrieve(_as_bundle(str_or_query_bundle))

        async def aretrieve(self, str_or_query_bundle) -> List[NodeWithScore]:
            return await self._aretrieve(_as_bundle(str_or_query_bundle))

        async def _aretrieve(self, query_bundle: QueryBundle) -> List[NodeWithScore]:
            return self._retrieve(query_bundle)


class QdrantRetriever(_RetrieverBase):
    def __init__(self, vector_store: HipVectorStore, embed_model, similarity_top_k: int = 2, filters=None) -> None:
        self._vector_store = vector_store
        self._embed_model = embed_model
        self._similarity_top_k = similarity_top_k
        self.filters = filters
        self._init_base()

    def _retrieve(self, query_bundle: QueryBundle) -> List[NodeWithScore]:
        emb = self._embed_model.get_query_embedding(query_bundle.query_str)
        nodes, sims = self._vector_store.query(emb, self._similarity_top_k, self.filters)
        return [NodeWithScore(node=n, score=s) for n, s in zip(nodes, sims)]

    def retrieve_batch(self, queries: Sequence[str], filters=_UNSET) -> List[List[NodeWithScore]]:
        """`_retrieve` for a batch.  `filters` (default: the `filters` attribute, for every query) may be a list parallel to
        `queries`: one qdrant Filter / dict / None per question, as the reference's evaluation loop sets them one at a time."""
        if len(queries) == 0:
            return []
        embs = np.asarray([self._embed_model.get_query_embedding(q) for q in queries], dtype=np.float32)
        ids, sc, ln = self._vector_store.query_batch(embs, self._similarity_top_k,
                                                     self.filters if filters is _UNSET else filters)
        nodes = self._vector_store.nodes
        return [[NodeWithScore(node=nodes[i], score=float(s)) for i, s in zip(ids[b, :ln[b]], sc[b, :ln[b]])]
                for b in range(len(queries))]


class BM25Retriever(_RetrieverBase):
    def __init__(self, nodes, tokenizer: Optional[Callable], similarity_top_k: int = DEFAULT_SIMILARITY_TOP_K,
                 callback_manager=None, objects=None, object_map=None, verbose: bool = False,
                 stopwords=("",), embed_type: int = 0, bm25_type: int = 0,
                 engine: Optional[RetrievalEngine] = None, payload_on_device: bool = False,
                 device_build: bool = True) -> None:
        self._nodes = list(nodes)
        self._tokenizer = tokenizer
        self._similarity_top_k = similarity_top_k
        self.embed_type = embed_type
        self.stopwords = stopwords
        # a NativeCutter tokenises, drops stop words and numbers the tokens inside the library (erh_text_encode): no Python
        # object per token.  Any other tokenizer (jieba.Tokenizer(), as the pipeline passes) goes through its own cut().
        # (a stop word the native side cannot express -- one with a NUL inside -- sends the build through the Python loop)
        native_text = device_build and hasattr(tokenizer, "encode_texts") and not self._str_stopwords(stopwords)[1]
        self._corpus = None if native_text else [
            tokenize_and_remove_stopwords(tokenizer, get_node_content(n, embed_type), stopwords) for n in self._nodes]
        self.bm25_type = bm25_type
        self.k1, self.b, self.epsilon = 1.5, 0.75, 0.25          # ref retrievers.py:103-105
        self._corpus_state = _corpus_for(self._nodes, engine)
        self.engine = self._corpus_state.engine
        # every retriever owns one BM25 index slot of the (possibly shared) engine: the content retriever and the
        # know_path retriever of the reference pipeline (pipeline.py:187-210) live side by side
        self._slot = self.engine.alloc_bm25_slot()
        variant = 1 if bm25_type == 1 else 0
        if device_build:
            # tokens -> ids on the host (dictionary lookups), everything else of the index build on the GPU
            # (erh_build_bm25_index: sort, run lengths, df, idf, payload); nothing but the vocabulary stays here
            if native_text:
                from .text import NativeVocab
                vocab = NativeVocab()
                flat, lens = tokenizer.encode_texts([get_node_content(n, embed_type) for n in self._nodes], vocab,
                                                    self._str_stopwords(stopwords)[0])
            else:
                vocab, flat, lens = vocab_ids(self._corpus)
            self.bm25: BM25Index = self.engine.build_bm25(flat, lens, max(len(vocab), 1), variant=variant, k1=self.k1,
                                                          b=self.b, epsilon=self.epsilon, slot=self._slot, fetch=False)
            self.bm25.vocab = vocab
        else:
            self.bm25 = build_bm25_index(self._corpus, variant=variant, k1=self.k1, b=self.b, epsilon=self.epsilon,
                                         compute_payload=not payload_on_device)
            self.engine.set_bm25(self.bm25, payload_on_device=payload_on_device, slot=self._slot)
        self.filter_dict = None
        self._init_base(callback_manager=callback_manager, object_map=object_map, objects=objects, verbose=verbose)

    @staticmethod
    def _str_stopwords(stopwords):
        """(usable stop words, the ones the native path cannot express): membership of '' never matters (no cutter emits an
        empty token); anything that is not a str can never equal a token."""
        sw = [w for w in stopwords if isinstance(w, str) and w != ""]
        return sw, [w for w in sw if "\x00" in w]

    def close(self):
        """Give the BM25 slot back to the engine."""
        if getattr(self, "_slot", None) is not None:
            self.engine.free_bm25_slot(self._slot)
            self._slot = None

    @classmethod
    def from_defaults(cls, index=None, nodes=None, docstore=None, tokenizer=None,
                      similarity_top_k: int = DEFAULT_SIMILARITY_TOP_K, verbose: bool = False,
                      stopwords=("",), embed_type: int = 0, bm25_type: int = 0, **kwargs) -> "BM25Retriever":
        if sum(bool(v) for v in (index, nodes, docstore)) != 1:
            raise ValueError("Please pass exactly one of index, nodes, or docstore.")
        if index is not None:
            docstore = index.docstore
        if docstore is not None:
            nodes = list(docstore.docs.values())
        assert nodes is not None, "Please pass exactly one of index, nodes, or docstore."
        return cls(nodes=nodes, tokenizer=tokenizer, similarity_top_k=similarity_top_k, verbose=verbose,
                   stopwords=stopwords, embed_type=embed_type, bm25_type=bm25_type, **kwargs)

    # -- scoring ----------------------------------------------------------------------------------
    def _query_ids(self, query: str, index: Optional[BM25Index] = None) -> np.ndarray:
        toks = tokenize_and_remove_stopwords(self._tokenizer, query, self.stopwords)
        if self.bm25_type == 1 and len(toks) == 0:
            raise IndexError("list index out of range")     # bm25s sniffs tokens[0] (ref behaviour)
        return (index or self.bm25).tokens_to_ids(toks)

    def get_scores(self, query: str, docs: Optional[Sequence[str]] = None) -> np.ndarray:
        """Score vector over all nodes (float64; float32 values widened for bm25_type 1).  With `docs` a
        throw-away index over those strings is built first (ref retrievers.py:131-147)."""
        if docs is None:
            return self._cast(self.engine.bm25_scores(self._query_ids(query), slot=self._slot))
        corpus = [tokenize_and_remove_stopwords(self._tokenizer, d, self.stopwords) for d in docs]
        idx = build_bm25_index(corpus, variant=1 if self.bm25_type == 1 else 0, k1=self.k1, b=self.b, epsilon=self.epsilon)
        # the throw-away index of a handful of sentences goes into the engine's scratch slot (kept across calls: no handle
        # creation, no kernel-attribute setup, no device allocation / free / synchronisation per call -- this sits on the
        # per-query path of the compressor)
        with _SCRATCH_LOCK:
            scratch = self.engine.scratch_bm25_slot()
            self.engine.set_bm25(idx, slot=scratch)
            return self._cast(self.engine.bm25_scores(self._query_ids(query, idx), slot=scratch))

    def _cast(self, s: np.ndarray) -> np.ndarray:
        return s.astype(np.float32) if self.bm25_type == 1 else s

    def filter(self, scores: np.ndarray) -> List[NodeWithScore]:
        """Host helper over a caller-supplied score vector (ref retrievers.py:191-210): descending walk
        (ties by node index), stop at score <= 0, metadata equality filter, first k."""
        scores = np.asarray(scores)
        order = np.lexsort((np.arange(scores.shape[0]), -scores.astype(np.float64)))
        out: List[NodeWithScore] = []
        for ix in order:
            if scores[ix] <= 0:
                break
            if self.filter_dict is not None and any(self._nodes[ix].metadata[k] != v for k, v in self.filter_dict.items()):
                continue
            out.append(NodeWithScore(node=self._nodes[ix], score=float(scores[ix])))
            if len(out) == self._similarity_top_k:
                break
        return out

    def _retrieve(self, query_bundle: QueryBundle) -> List[NodeWithScore]:
        if query_bundle.custom_embedding_strs or query_bundle.embedding:
            logger.warning("BM25Retriever does not support embeddings, skipping...")
        return self.retrieve_batch([query_bundle.query_str])[0]

    def retrieve_batch(self, queries: Sequence[str], filter_dicts=_UNSET) -> List[List[NodeWithScore]]:
        """get_scores + filter for a whole batch, fused on the GPU (no score vector leaves the chip).  `filter_dicts`
        (default: the `filter_dict` attribute, for every query) may be a list parallel to `queries`: one dict / None each."""
        B = len(queries)
        if B == 0:
            return []
        dicts = _per_query(self.filter_dict if filter_dicts is _UNSET else filter_dicts, B, lambda v: dict(v) if v else None)
        tok = [self._query_ids(q) for q in queries]
        out: List[List[NodeWithScore]] = [[] for _ in range(B)]
        for g in _by_key_set(dicts):
            qi, qt = queries_to_csr([tok[b] for b in g])
            ids, sc, ln = self.engine.bm25_topk(qi, qt, self._similarity_top_k, slot=self._slot,
                                                filter_dir=self._corpus_state.filter_column([dicts[b] for b in g]))
            for j, b in enumerate(g):
                out[b] = [NodeWithScore(node=self._nodes[i], score=float(s)) for i, s in zip(ids[j, :ln[j]], sc[j, :ln[j]])]
        return out


_SCRATCH_LOCK = threading.Lock()
_FUSION_ENGINE: Optional[RetrievalEngine] = None
_FUSION_LOCK = threading.Lock()


def _fusion_engine() -> RetrievalEngine:
    global _FUSION_ENGINE
    if _FUSION_ENGINE is None:
        _FUSION_ENGINE = RetrievalEngine()
    return _FUSION_ENGINE


def _lists_to_device_form(lists):
    """Flatten lists of NodeWithScore into local item ids + content ids for the fusion kernels."""
    items, cid, key = [], [], {}
    per_list = []
    for lst in lists:
        ids = []
        for it in lst:
            ids.append(len(items))
            cid.append(key.setdefault(it.get_content(), len(key)))
            items.append(it)
        per_list.append(ids)
    return items, np.asarray(cid, np.int32), per_list


class HybridRetriever(_RetrieverBase):
    def __init__(self, dense_retriever: QdrantRetriever, sparse_retriever: BM25Retriever, retrieval_type: int = 1,
                 topk: int = 256):
        self.dense_retriever = dense_retriever
        self.sparse_retriever = sparse_retriever
        self.retrieval_type = retrieval_type     # 1: dense only  2: sparse only  3: hybrid (RRF)
        self.filters = None
        self.filter_dict = None
        self.topk = topk
        self._init_base()

    # -- classmethods over already-retrieved lists (ref retrievers.py:239-274) ----------------------------
    # The pipeline calls them once per query on a few hundred NodeWithScore objects it already holds on the host
    # (pipeline.py:362, 408: 192 + 6 or 192 + 288 items).  A kernel launch plus six PCIe stagings for that is slower than
    # the reference's own loop, so lists of up to `fusion_device_min` items are merged right here on the objects (any
    # number of lists, as the reference accepts); the library's fusion kernels (erh_rrf / erh_fusion, and the fused
    # erh_hybrid_topk that `aretrieve` runs) take over for larger inputs and for everything that is already on the device.
    fusion_device_min = 512          # (and at most 2048 items: the kernels' LDS budget)

    @staticmethod
    def _fuse_host(lists, rrf: bool, K: int, topk: int):
        if not rrf:                                  # first occurrence of a content wins, raw scores, stable sort
            seen, merged = set(), []
            for lst in lists:
                for node in lst:
                    c = node.get_content()
                    if c not in seen:
                        merged.append(node)
                        seen.add(c)
            merged.sort(key=lambda node: node.score, reverse=True)
            return merged[:min(len(merged), topk)]
        score, last = {}, {}                         # dict order = first appearance; the LAST node of a content is returned
        for lst in lists:
            for rank, item in enumerate(lst, 1):
                c = item.get_content()
                last[c] = item
                score[c] = score.get(c, 0.0) + 1 / (rank + K)
        out = []
        for c, sc in sorted(score.items(), key=lambda kv: kv[1], reverse=True):
            node = last[c]
            node.score = sc                          # the reference overwrites the node's score with the RRF score
            out.append(node)
        return out[:min(topk, len(out))]

    @classmethod
    def _fuse(cls, lists, rrf: bool, K: int, topk: int):
        lists = [list(x) for x in lists]
        n_items = sum(len(x) for x in lists)
        if n_items <= cls.fusion_device_min or n_items > 2048 or len(lists) > 2:
            return cls._fuse_host(lists, rrf, K, topk)
        if len(lists) == 1:
            lists.append([])
        items, cid, per_list = _lists_to_device_form(lists)
        if not items:
            return []
        da, db = max(len(per_list[0]), 1), max(len(per_list[1]), 1)
        ia = np.full((1, da), -1, np.int32)
        ib = np.full((1, db), -1, np.int32)
        ia[0, :len(per_list[0])] = per_list[0]
        ib[0, :len(per_list[1])] = per_list[1]
        la = np.asarray([len(per_list[0])], np.int32)
        lb = np.asarray([len(per_list[1])], np.int32)
        with _FUSION_LOCK:                       # the classmethods share one small handle (item ids are list-local)
            eng = _fusion_engine()
            eng.set_doc_meta(len(items), cid, None)
            if rrf:
                ids, sc, ln = eng.rrf(ia, la, ib, lb, K=K, topk=topk)
            else:
                sa = np.zeros((1, da), np.float64)
                sb = np.zeros((1, db), np.float64)
                sa[0, :la[0]] = [items[i].score for i in per_list[0]]
                sb[0, :lb[0]] = [items[i].score for i in per_list[1]]
                ids, sc, ln = eng.fusion(ia, sa, la, ib, sb, lb, topk=topk)
        out = []
        for i, s in zip(ids[0, :ln[0]], sc[0, :ln[0]]):
            node = items[i]
            if rrf:
                node.score = float(s)        # the reference overwrites the node's score with the RRF score
            out.append(node)
        return out

    @classmethod
    def fusion(cls, list_of_list_ranks_system, topk: int = 256):
        return cls._fuse(list_of_list_ranks_system, rrf=False, K=0, topk=topk)

    @classmethod
    def reciprocal_rank_fusion(cls, list_of_list_ranks_system, K: int = 60, topk: int = 256):
        return cls._fuse(list_of_list_ranks_system, rrf=True, K=K, topk=topk)

    # -- routes ---------------------------------------------------------------------------------------------
    def _fused_possible(self) -> bool:
        vs = getattr(self.dense_retriever, "_vector_store", None)
        return (isinstance(vs, HipVectorStore) and vs.engine is self.sparse_retriever.engine
                and len(vs.nodes) == len(self.sparse_retriever._nodes))

    def _retrieve(self, query_bundle: QueryBundle) -> List[NodeWithScore]:
        """The reference's sync path (retrievers.py:293-305, marked unmaintained there): both routes with whatever
        filters the child retrievers currently hold, sparse list then dense list, de-duplicated by node_id; no
        route selection, no RRF, no filter push-down.  The pipeline calls `aretrieve`."""
        sparse_nodes = self.sparse_retriever.retrieve(query_bundle)
        dense_nodes = self.dense_retriever.retrieve(query_bundle)
        all_nodes, node_ids = [], set()
        for n in sparse_nodes + dense_nodes:
            if n.node.node_id not in node_ids:
                all_nodes.append(n)
                node_ids.add(n.node.node_id)
        return all_nodes

    async def _aretrieve(self, query_bundle: QueryBundle) -> List[NodeWithScore]:
        """Route selection + filter push-down + RRF (retrievers.py:276-291)."""
        return self.retrieve_batch([query_bundle.query_str])[0]

    def retrieve_batch(self, queries: Sequence[str], filter_dicts=_UNSET, filters=_UNSET) -> List[List[NodeWithScore]]:
        """`_aretrieve` for a batch of query strings.  Without the keyword arguments the `filter_dict` / `filters`
        attributes apply to every query and are pushed into the child retrievers as the reference does
        (retrievers.py:278,283).  `filter_dicts` (sparse route) / `filters` (dense route) may instead be lists parallel to
        `queries` -- the reference's evaluation loop changes both before every question (pipeline.py:301-312,331-341), so a
        batch of its questions needs one filter per query; the child retrievers' attributes are left alone then."""
        sp, de = self.sparse_retriever, self.dense_retriever
        B = len(queries)
        if B == 0:
            return []
        fs = self.filter_dict if filter_dicts is _UNSET else filter_dicts
        fdn = self.filters if filters is _UNSET else filters
        if self.retrieval_type != 1:
            if filter_dicts is _UNSET:
                sp.filter_dict = self.filter_dict
            if self.retrieval_type == 2:
                return sp.retrieve_batch(queries, filter_dicts=fs)
        if self.retrieval_type != 2:
            if filters is _UNSET:
                de.filters = self.filters
            if self.retrieval_type == 1:
                return de.retrieve_batch(queries, filters=fdn)
        # filter_dict restricts the sparse route only, filters the dense route only (retrievers.py:278,283)
        d_sparse = _per_query(fs, B, lambda v: dict(v) if v else None)
        d_dense = _per_query(fdn, B, _filter_to_dict)
        key_sets = {tuple(sorted(fd)) for fd in d_sparse + d_dense if fd}
        if not self._fused_possible() or len(key_sets) > 1:
            sparse, dense = sp.retrieve_batch(queries, filter_dicts=fs), de.retrieve_batch(queries, filters=fdn)
            return [self.reciprocal_rank_fusion([s, d], topk=self.topk) for s, d in zip(sparse, dense)]
        # both routes share one engine: BM25 -> dense -> RRF([sparse, dense]) in a single device pipeline, one class column
        # per route over the shared key set (a query's two filters may still differ, or one of them be absent)
        cstate = sp._corpus_state
        filt_s, filt_d = cstate.filter_column(d_sparse), cstate.filter_column(d_dense)
        qi, qt = queries_to_csr([sp._query_ids(q) for q in queries])
        embs = _unit_f16(np.asarray([de._embed_model.get_query_embedding(q) for q in queries], dtype=np.float32))
        ids, sc, ln = sp.engine.hybrid_topk(embs, qi, qt, k_dense=de._similarity_top_k,
                                            k_sparse=sp._similarity_top_k, K=60, topk=self.topk,
                                            filter_dir=filt_s, filter_dense=filt_d, slot=sp._slot)
        nodes = sp._nodes
        return [[NodeWithScore(node=nodes[i], score=float(s)) for i, s in zip(ids[b, :ln[b]], sc[b, :ln[b]])]
                for b in range(B)]
