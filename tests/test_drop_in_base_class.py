"""Drop-in boundary, no GPU: with llama_index importable the three retrievers subclass its BaseRetriever
(ref /root/reference/src/easyrag/custom/retrievers.py:23,80,121-126,223) -- `retrieve` / `aretrieve` are the base class's
(callback events around `_retrieve`), the constructor arguments reach `BaseRetriever.__init__`, and a wrapper written against
BaseRetriever (what AutoMergingRetriever is, ref pipeline.py:212-217) accepts them.  The real package is not installable in
the build container: tests/fake_llama_index carries the slice of its API that matters, and the check runs in a subprocess
with that directory on sys.path (the import decision is taken when easyrag_amd.retrievers is first imported).
Also: embed_type 6 (the table-header merge over the PREVIOUS relationship, ref ingestion.py:35-57)."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE = os.path.join(ROOT, "tests", "fake_llama_index")

_CHILD = textwrap.dedent('''
    import asyncio, sys
    import numpy as np
    from llama_index.core.base.base_retriever import BaseRetriever
    from llama_index.core.callbacks import CallbackManager
    from llama_index.core.schema import TextNode, NodeWithScore
    import easyrag_amd.schema as schema
    assert schema.HAVE_LLAMA_INDEX and schema.TextNode is TextNode          # the package's node types ARE llama_index's
    from easyrag_amd import retrievers as R
    for cls in (R.QdrantRetriever, R.BM25Retriever, R.HybridRetriever):
        assert issubclass(cls, BaseRetriever), cls
        assert "retrieve" not in cls.__dict__ and "aretrieve" not in cls.__dict__   # inherited from llama_index

    class FakeIndex:                                     # what engine.build_bm25 returns, as far as the ctor reads it
        vocab = None
        def tokens_to_ids(self, toks):
            return np.asarray([self.vocab[t] for t in toks if t in self.vocab], np.int32)

    class FakeEngine:                                    # no GPU in this test: the engine calls the ctor makes, recorded
        corpus = None
        def __init__(self): self.calls = []
        def set_doc_meta(self, *a): self.calls.append("set_doc_meta")
        def alloc_bm25_slot(self): return 0
        def build_bm25(self, flat, lens, V, **kw):
            self.calls.append(("build_bm25", len(flat), len(lens), V, kw["variant"]))
            return FakeIndex()
        def bm25_topk(self, qi, qt, k, filter_dir=None, slot=0):
            self.calls.append(("bm25_topk", k))
            B = len(qi) - 1
            return np.zeros((B, k), np.int32), np.ones((B, k)), np.full(B, 1, np.int32)

    class Tok:
        def cut(self, text): return text.split(" ")

    nodes = [TextNode(text="alpha beta", id_="a"), TextNode(text="beta gamma", id_="b")]
    cm = CallbackManager()
    eng = FakeEngine()
    sp = R.BM25Retriever.from_defaults(nodes=nodes, tokenizer=Tok(), similarity_top_k=1, verbose=True, stopwords=[""],
                                       bm25_type=1, engine=eng, callback_manager=cm, object_map={"k": "v"})
    assert isinstance(sp, BaseRetriever)
    assert sp.callback_manager is cm and sp.object_map == {"k": "v"} and sp._verbose is True   # reached BaseRetriever.__init__
    assert ("build_bm25", 4, 2, 3, 1) in eng.calls
    out = sp.retrieve("beta")                            # BaseRetriever.retrieve -> our _retrieve -> engine.bm25_topk
    assert [n.node.id_ for n in out] == ["a"] and isinstance(out[0], NodeWithScore)
    assert cm.events == [("retrieve:start", "beta"), ("retrieve:end", 1)]

    class Embed:
        def get_query_embedding(self, q): return [1.0, 0.0]
    class Store:                                         # QdrantRetriever only needs .query / .query_batch / .nodes
        nodes = nodes
        def query(self, emb, k, filters=None): return [nodes[1]], [0.5]
    de = R.QdrantRetriever(Store(), Embed(), similarity_top_k=1)
    hy = R.HybridRetriever(de, sp, retrieval_type=2, topk=4)
    assert isinstance(de, BaseRetriever) and isinstance(hy, BaseRetriever)
    assert [n.node.id_ for n in de.retrieve("q")] == ["b"]
    got = asyncio.run(hy.aretrieve("beta"))              # BaseRetriever.aretrieve -> our _aretrieve (route 2 = sparse)
    assert [n.node.id_ for n in got] == ["a"]
    assert ("retrieve:start", "beta") in hy.callback_manager.events

    class Wrapper(BaseRetriever):                        # the shape of AutoMergingRetriever: wraps ANY BaseRetriever
        def __init__(self, inner):
            assert isinstance(inner, BaseRetriever)
            self.inner = inner
            super().__init__(callback_manager=inner.callback_manager, object_map=inner.object_map, verbose=inner._verbose)
        def _retrieve(self, qb): return self.inner.retrieve(qb)
    assert [n.node.id_ for n in Wrapper(sp).retrieve("beta")] == ["a"]
    print("DROP-IN-OK")
''')


def _run_child(extra_path):
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([p for p in (extra_path, ROOT, env.get("PYTHONPATH", "")) if p])
    return subprocess.run([sys.executable, "-c", _CHILD], env=env, capture_output=True, text=True, timeout=300)


def test_retrievers_subclass_llama_index_base_retriever_when_importable():
    r = _run_child(FAKE)
    assert r.returncode == 0 and "DROP-IN-OK" in r.stdout, r.stdout + r.stderr


def test_fallback_base_without_llama_index():
    """In this container (no llama_index) the fallback base keeps the same surface and the ctor arguments."""
    from easyrag_amd import retrievers as R
    from easyrag_amd.schema import HAVE_LLAMA_INDEX
    if HAVE_LLAMA_INDEX:
        pytest.skip("llama_index is installed here")
    de = R.QdrantRetriever(object(), object(), similarity_top_k=3)
    hy = R.HybridRetriever(de, object(), retrieval_type=1, topk=7)
    for r in (de, hy):
        assert isinstance(r, R._RetrieverBase) and r.callback_manager is None and r.object_map == {} and r._verbose is False
        assert callable(r.retrieve) and callable(r.aretrieve)


def test_embed_type_6_merges_the_table_header():
    from easyrag_amd.retrievers import get_node_content
    from easyrag_amd.schema import NodeWithScore, TextNode

    class Rel:                                           # RelatedNodeInfo
        def __init__(self, node_id): self.node_id = node_id

    def node(text, id_, prev=None, meta=None):
        n = TextNode(text=text, id_=id_, metadata=meta or {})
        n.relationships = {"2": Rel(prev)} if prev else {}    # NodeRelationship.PREVIOUS == "2"
        return n

    head = node("intro\nname | port | role | x | y\n--- | --- | --- | --- | ---\na | 1 | x | p | q\nb | 2 | y | p | q", "n0")
    body = node("b | 2 | y | p | q\nc | 3 | z | p | q", "n1", prev="n0")
    plain = node("no table here", "n2", prev="n1")
    far = node("e | 5 | v | p | q\nf | 6 | u | p | q", "n3", prev="n2")        # predecessor has no rule: unchanged
    nodes = [head, body, plain, far]
    nid = {n.id_: i for i, n in enumerate(nodes)}
    got = get_node_content(NodeWithScore(node=body, score=1.0), 6, nodes, nid)
    # the line before the first "---" (stripped: the reference glues it to the rule) + everything from the rule on, the
    # overlapping row "b | 2 | y | p | q" written once
    assert got == ("name | port | role | x | y--- | --- | --- | --- | ---\na | 1 | x | p | q\nb | 2 | y | p | q\n"
                   "c | 3 | z | p | q")
    assert get_node_content(NodeWithScore(node=far, score=1.0), 6, nodes, nid) == far.text
    assert get_node_content(NodeWithScore(node=plain, score=1.0), 6, nodes, nid) == "no table here"
    assert get_node_content(NodeWithScore(node=head, score=1.0), 6, nodes, nid) == head.text   # has its own rule
    # the image-caption expansion of type 3 still applies to type 6
    img = node("Fig 1 title\nrest", "n4", meta={"imgobjs": [{"cap": "Fig 1", "title": "title", "content": "a chart"}]})
    assert get_node_content(NodeWithScore(node=img, score=1.0), 6, [img], {"n4": 0}) == "Fig 1.title:a chart\nrest"
    # a bare TextNode with a table-like text: the reference dereferences node.node and raises; so does this
    with pytest.raises(AttributeError):
        get_node_content(body, 6, nodes, nid)


def test_embed_type_6_with_the_stand_in_node_types_only():
    """without llama_index the package's OWN TextNode must carry `relationships` (embed_type 6 reads
    node.node.relationships) and offer a RelatedNodeInfo with node_id -- no attribute patched on from outside."""
    from easyrag_amd import schema
    from easyrag_amd.retrievers import get_node_content
    if schema.HAVE_LLAMA_INDEX:
        pytest.skip("llama_index is installed here")
    head = schema.TextNode(text="t\nname | port | role | x | y\n--- | --- | --- | --- | ---\na | 1 | x | p | q", id_="h")
    assert head.relationships == {}
    for key in ("PREVIOUS", "2", 2):
        body = schema.TextNode(text="a | 1 | x | p | q\nb | 2 | y | p | q", id_="b",
                               relationships={key: schema.RelatedNodeInfo(node_id="h")})
        got = get_node_content(schema.NodeWithScore(node=body, score=0.5), 6, [head, body], {"h": 0, "b": 1})
        assert got == "name | port | role | x | y--- | --- | --- | --- | ---\na | 1 | x | p | q\nb | 2 | y | p | q"
    # a table-like chunk WITHOUT a predecessor link: the reference raises KeyError on relationships[PREVIOUS]; so does this
    lone = schema.TextNode(text="a | 1 | x | p | q\nb | 2 | y | p | q", id_="l")
    with pytest.raises(KeyError):
        get_node_content(schema.NodeWithScore(node=lone, score=0.5), 6, [lone], {"l": 0})


def test_engine_slot_bookkeeping_without_a_device():
    """alloc_bm25_slot reserves (two allocations without a set in between must differ), the scratch slot is
    given back by release_scratch, and the 'all slots in use' error says where the fourth slot went."""
    from easyrag_amd import _lib
    from easyrag_amd.engine import RetrievalEngine
    eng = RetrievalEngine.__new__(RetrievalEngine)          # bookkeeping only: no handle, no device
    eng._h = None
    eng._bm25_slots = [None] * _lib.ERH_BM25_SLOTS
    eng._bm25_cur = 0
    a, b = eng.alloc_bm25_slot(), eng.alloc_bm25_slot()
    assert a != b and eng.bm25 is None
    s = eng.scratch_bm25_slot()
    assert s not in (a, b) and eng.scratch_bm25_slot() == s
    last = eng.alloc_bm25_slot()
    assert len({a, b, s, last}) == _lib.ERH_BM25_SLOTS
    with pytest.raises(RuntimeError, match="release_scratch"):
        eng.alloc_bm25_slot()
    eng.release_scratch()
    assert eng.alloc_bm25_slot() == s                        # the freed scratch slot is handed out again
    eng.free_bm25_slot(a)
    assert eng.scratch_bm25_slot() == a
