"""The restart path of the drop-in (missing 4): the reference skips ingestion when its Qdrant collection is already
populated (ref src/easyrag/pipeline/pipeline.py:138-141), so after a restart the embeddings exist only in Qdrant.
HipVectorStore.from_qdrant / afrom_qdrant read them back through the client's scroll(); save / load keep the fp16 matrix in a
file.  Host logic here (a fake client shaped like qdrant-client's scroll, a recording stand-in for the engine); the GPU
round trip is tests/test_gpu_retrievers.py::test_vector_store_restart_round_trip."""
import asyncio
import json

import numpy as np
import pytest

from easyrag_amd.retrievers import HipVectorStore
from easyrag_amd.schema import TextNode


class Point:
    def __init__(self, id, vector, payload):
        self.id, self.vector, self.payload = id, vector, payload


class FakeQdrant:
    """scroll(collection_name, limit, offset, with_payload, with_vectors) -> (points, next_offset), as qdrant-client 1.8.2."""

    def __init__(self, points):
        self.points, self.calls = points, []

    def scroll(self, collection_name, limit=10, offset=None, with_payload=True, with_vectors=False, **kw):
        self.calls.append((collection_name, limit, offset, with_payload, with_vectors))
        start = int(offset or 0)
        batch = self.points[start:start + limit]
        nxt = start + limit if start + limit < len(self.points) else None
        if not with_vectors:
            batch = [Point(p.id, None, p.payload) for p in batch]
        return batch, nxt


class AsyncFakeQdrant(FakeQdrant):
    async def scroll(self, *a, **kw):                     # AsyncQdrantClient: the same call, awaited
        return FakeQdrant.scroll(self, *a, **kw)


class StubEngine:
    """What HipVectorStore touches of a RetrievalEngine, recording the matrix it is given."""

    def __init__(self):
        self.corpus, self.x, self.normalize, self.n_dense, self.d = None, None, None, 0, 0

    def set_doc_meta(self, n, cid, did): self.n_meta = n

    def set_dense(self, x, normalize=False):
        self.x, self.normalize = np.array(x), normalize
        self.n_dense, self.d = self.x.shape

    def get_dense_rows(self, row0, rows, out=None):
        out[...] = self.x[row0:row0 + rows]
        return out


def _corpus(n=37, d=64, seed=0, dup=True):
    rng = np.random.default_rng(seed)
    nodes = [TextNode(text=f"chunk number {i}", metadata={"file_path": f"f{i % 5}.txt", "dir": "umac" if i % 2 else "rcp"}, id_=f"n{i}")
             for i in range(n)]
    if dup:                                                   # two pairs of identical texts in different files
        nodes[7].text = nodes[3].text
        nodes[20].text = nodes[11].text
    emb = rng.standard_normal((n, d)).astype(np.float32) * 3.0
    return nodes, emb


def _points(nodes, emb, ids=None, order=None, flat_text=False):
    order = list(range(len(nodes))) if order is None else order
    pts = []
    for j in order:
        n = nodes[j]
        payload = dict(n.metadata)
        if flat_text:
            payload["text"] = n.text
        else:                                                 # llama-index-vector-stores-qdrant: the node's JSON rides in the payload
            payload.update({"_node_content": json.dumps({"id_": n.id_, "text": n.text, "metadata": n.metadata}),
                            "_node_type": "TextNode", "doc_id": "None", "document_id": "None", "ref_doc_id": "None"})
        pts.append(Point(ids[j] if ids else n.id_, [float(v) for v in emb[j]], payload))
    return pts


def _unit16(emb):
    e = emb.astype(np.float64)
    return (e / np.linalg.norm(e, axis=1, keepdims=True)).astype(np.float16)


def test_from_qdrant_matches_points_to_nodes_by_id():
    nodes, emb = _corpus()
    order = list(np.random.default_rng(1).permutation(len(nodes)))             # the collection's scroll order is its own
    client = FakeQdrant(_points(nodes, emb, order=order))
    eng = StubEngine()
    vs = HipVectorStore.from_qdrant(client, "aiops24", nodes, engine=eng, batch_size=10)
    assert vs.nodes == nodes and eng.x.dtype == np.float16
    assert np.array_equal(eng.x, _unit16(emb))                                  # row i belongs to nodes[i]; unit rows, rounded once
    assert [c[2] for c in client.calls] == [None, 10, 20, 30] and all(c[0] == "aiops24" and c[4] for c in client.calls)


def test_from_qdrant_matches_by_content_when_ids_differ():
    """The reference's sparse-route nodes come from a SECOND run of the splitter (ref pipeline.py:160-167): fresh node ids, same
    texts.  Points are then joined on the text (the key RRF itself uses), equal texts by file_path."""
    nodes, emb = _corpus()
    ids = [f"uuid-{i:04d}" for i in range(len(nodes))]
    order = list(np.random.default_rng(2).permutation(len(nodes)))
    for flat in (False, True):
        eng = StubEngine()
        HipVectorStore.from_qdrant(FakeQdrant(_points(nodes, emb, ids=ids, order=order, flat_text=flat)), "c", nodes, engine=eng)
        assert np.array_equal(eng.x, _unit16(emb)), flat      # (nodes 3 / 7 and 11 / 20 share their text: file_path decides)


def test_from_qdrant_without_nodes_rebuilds_them_from_the_payloads():
    nodes, emb = _corpus(dup=False)
    eng = StubEngine()
    vs = HipVectorStore.from_qdrant(FakeQdrant(_points(nodes, emb)), "c", engine=eng)
    assert [n.get_content() for n in vs.nodes] == [n.text for n in nodes]
    assert [n.metadata for n in vs.nodes] == [n.metadata for n in nodes] and [n.node_id for n in vs.nodes] == [n.id_ for n in nodes]
    assert np.array_equal(eng.x, _unit16(emb))


def test_from_qdrant_errors():
    nodes, emb = _corpus()
    with pytest.raises(ValueError, match="empty"):
        HipVectorStore.from_qdrant(FakeQdrant([]), "c", nodes, engine=StubEngine())
    other = [TextNode(text=f"other {i}", id_=f"x{i}") for i in range(len(nodes))]
    with pytest.raises(ValueError, match="have no point with their text"):
        HipVectorStore.from_qdrant(FakeQdrant(_points(nodes, emb)), "c", other, engine=StubEngine())
    with pytest.raises(TypeError, match="afrom_qdrant"):
        HipVectorStore.from_qdrant(AsyncFakeQdrant(_points(nodes, emb)), "c", nodes, engine=StubEngine())
    pts = _points(nodes, emb)
    pts[4].vector = {"a": pts[4].vector, "b": pts[4].vector}
    with pytest.raises(ValueError, match="named vectors"):
        HipVectorStore.from_qdrant(FakeQdrant(pts), "c", nodes, engine=StubEngine())


def test_afrom_qdrant_with_the_async_client():
    nodes, emb = _corpus()
    eng = StubEngine()
    pts = _points(nodes, emb)
    pts[0].vector = {"": pts[0].vector}                       # a single named vector is taken as it is
    asyncio.run(HipVectorStore.afrom_qdrant(AsyncFakeQdrant(pts), "aiops24", nodes, engine=eng, batch_size=16))
    assert np.array_equal(eng.x, _unit16(emb))


def test_save_load_round_trip_and_its_checks(tmp_path):
    nodes, emb = _corpus()
    eng = StubEngine()
    vs = HipVectorStore(nodes, emb, engine=eng)
    path = vs.save(tmp_path / "chunks")
    assert path.endswith("chunks.npy")
    meta = json.load(open(path + ".meta.json"))
    assert meta["n"] == len(nodes) and meta["d"] == 64 and meta["dtype"] == "float16"
    eng2 = StubEngine()
    HipVectorStore.load(tmp_path / "chunks", nodes, engine=eng2)
    assert eng2.normalize is False and eng2.x.dtype == np.float16 and np.array_equal(eng2.x, eng.x)    # rows as saved: no second rounding
    with pytest.raises(ValueError, match="rows"):
        HipVectorStore.load(path, nodes[:-1], engine=StubEngine())
    changed = list(nodes)
    changed[5] = TextNode(text="edited", id_="n5")
    with pytest.raises(ValueError, match="fingerprint"):
        HipVectorStore.load(path, changed, engine=StubEngine())
    HipVectorStore.load(path, changed, engine=StubEngine(), check=False)
