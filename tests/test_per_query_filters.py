"""CPU: the per-query filter plumbing of the drop-in retrievers (retrieve_batch(queries, filter_dicts=[...]) / filters=[...]):
list handling, class columns, grouping by metadata key set.  No GPU: a stub engine records what would be launched."""
import numpy as np
import pytest

from easyrag_amd import retrievers as R
from easyrag_amd.schema import TextNode


class StubEngine:
    corpus = None

    def __init__(self):
        self.meta = []
        self.calls = []

    def set_doc_meta(self, n, cid, dir_id):
        self.meta.append(None if dir_id is None else np.array(dir_id))

    def dense_topk(self, q, k, filter_dir=None, mode=0):
        self.calls.append(None if filter_dir is None else np.array(filter_dir))
        B = q.shape[0]
        ids = np.tile(np.arange(k, dtype=np.int32), (B, 1))
        return ids, np.zeros((B, k)), np.full(B, k, np.int32)

    def set_dense(self, x, normalize=False):
        pass


def _nodes():
    dirs, kinds = ["a", "b", "c"], ["x", "y"]
    return [TextNode(text=f"t{i}", metadata={"dir": dirs[i % 3], "kind": kinds[i % 2]}, id_=f"n{i}") for i in range(12)]


def test_per_query_accepts_scalar_and_list():
    to = R._filter_to_dict
    assert R._per_query(None, 3, to) == [None, None, None]
    assert R._per_query({"dir": "a"}, 2, to) == [{"dir": "a"}, {"dir": "a"}]
    assert R._per_query([{"dir": "a"}, None, {}], 3, to) == [{"dir": "a"}, None, None]
    assert R._per_query(({"dir": "a"},), 1, to) == [{"dir": "a"}]
    with pytest.raises(ValueError):
        R._per_query([None], 2, to)


def test_by_key_set_groups():
    assert R._by_key_set([None, None]) == [[0, 1]]
    assert R._by_key_set([]) == []
    assert R._by_key_set([{"dir": "a"}, None, {"dir": "b"}]) == [[0, 1, 2]]
    assert R._by_key_set([{"dir": "a"}, None, {"kind": "x"}, {"dir": "b"}]) == [[0, 1, 3], [2]]


def test_filter_column_per_query_classes():
    eng = StubEngine()
    c = R._FilteredCorpus(_nodes(), eng)
    assert c.filter_column([None, {}, None]) is None
    col = c.filter_column([{"dir": "b"}, None, {"dir": "a"}, {"dir": "zzz"}])
    assert col.dtype == np.int16 and list(col) == [1, -1, 0, 32767]          # classes by first appearance; unknown matches nothing
    assert list(eng.meta[-1][:4]) == [0, 1, 2, 0]                             # the dir column went to the device once
    n_uploads = len(eng.meta)
    c.filter_column([{"dir": "c"}])
    assert len(eng.meta) == n_uploads                                         # same key set: no new upload
    with pytest.raises(ValueError):
        c.filter_column([{"dir": "a"}, {"kind": "x"}])


def test_vector_store_query_batch_with_mixed_key_sets():
    eng = StubEngine()
    nodes = _nodes()
    store = R.HipVectorStore(nodes, np.eye(12, 64, dtype=np.float32), engine=eng)
    q = np.eye(4, 64, dtype=np.float32)
    store.query_batch(q, 3, filters=[{"dir": "a"}, None, {"dir": "c"}, {"dir": "a"}])
    assert len(eng.calls) == 1 and list(eng.calls[0]) == [0, -1, 2, 0]
    eng.calls.clear()
    ids, sc, ln = store.query_batch(q, 3, filters=[{"dir": "a"}, {"kind": "y"}, None, {"kind": "x"}])
    assert [list(c) for c in eng.calls] == [[0, -1], [1, 0]]                 # two device batches, one column each
    assert ids.shape == (4, 3) and list(ln) == [3, 3, 3, 3]
    eng.calls.clear()
    store.query_batch(q, 3, filters={"dir": "b"})                             # the scalar knob: every query
    assert list(eng.calls[0]) == [1, 1, 1, 1]
    store.query_batch(q, 3)
    assert eng.calls[-1] is None
