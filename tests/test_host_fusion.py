"""Host logic of the shim, no GPU: the small-list branch of HybridRetriever.fusion / reciprocal_rank_fusion (the
classmethods the reference pipeline calls once per query, retrievers.py:239-274) against the oracle's restatement, on
random lists with duplicated contents, any number of lists, empty lists and ties."""
import random

import pytest
from hypothesis import given, settings, strategies as st

from easyrag_amd.retrievers import HybridRetriever
from easyrag_amd.schema import NodeWithScore, TextNode
from oracle.retrievers import Item, fusion, reciprocal_rank_fusion


def _make(lists_spec):
    """lists_spec: per list a sequence of (content number, score).  Nodes for the shim, Items for the oracle (the item index is
    the position in the flattened input, so 'which node object is returned' can be compared)."""
    nodes, items, n = [], [], 0
    for spec in lists_spec:
        ln, li = [], []
        for content, score in spec:
            node = NodeWithScore(node=TextNode(text=f"chunk {content}", id_=f"id-{n}"), score=score)
            ln.append(node)
            li.append(Item(n, f"chunk {content}", score))
            n += 1
        nodes.append(ln)
        items.append(li)
    return nodes, items


lists_strategy = st.lists(
    st.lists(st.tuples(st.integers(0, 12), st.sampled_from([0.5, 1.0, 1.0, 2.25, 3.0, 7.5])), min_size=0, max_size=24),
    min_size=0, max_size=4)


@settings(max_examples=200, deadline=None)
@given(lists_strategy, st.integers(1, 30))
def test_host_rrf_matches_oracle(spec, topk):
    nodes, items = _make(spec)
    want = reciprocal_rank_fusion(items, K=60, topk=topk)
    got = HybridRetriever.reciprocal_rank_fusion(nodes, K=60, topk=topk)
    assert [g.node.id_ for g in got] == [f"id-{w.idx}" for w in want]        # the LAST node seen for a content is returned
    assert [g.score for g in got] == [w.score for w in want]                  # fp64 sums, same order of additions


@settings(max_examples=200, deadline=None)
@given(lists_strategy, st.integers(1, 30))
def test_host_fusion_matches_oracle(spec, topk):
    nodes, items = _make(spec)
    want = fusion(items, topk=topk)
    got = HybridRetriever.fusion(nodes, topk=topk)
    assert [g.node.id_ for g in got] == [f"id-{w.idx}" for w in want]        # first occurrence wins, stable among equal scores
    assert [g.score for g in got] == [w.score for w in want]


def test_host_branch_is_what_small_lists_take():
    """The classmethods must not need a GPU for the list sizes the pipeline passes (192 + 6, 192 + 288 items)."""
    rnd = random.Random(5)
    spec = [[(rnd.randrange(400), rnd.random()) for _ in range(192)], [(rnd.randrange(400), rnd.random()) for _ in range(288)]]
    nodes, items = _make(spec)
    assert sum(len(x) for x in nodes) <= HybridRetriever.fusion_device_min
    got = HybridRetriever.reciprocal_rank_fusion(nodes, topk=256)
    want = reciprocal_rank_fusion(items, topk=256)
    assert [g.node.id_ for g in got] == [f"id-{w.idx}" for w in want]
