"""Oracle: the reference's own glue in src/easyrag/custom/retrievers.py.  TEST INFRASTRUCTURE ONLY.

Restated function by function (nodes are replaced by light `Item`s carrying the node index,
its text key and a score, because llama_index is not installable here):

  tokenize_and_remove_stopwords   retrievers.py:72-76
  bm25_filter                     BM25Retriever.filter, retrievers.py:191-210
  reciprocal_rank_fusion          HybridRetriever.reciprocal_rank_fusion, retrievers.py:256-274
  fusion                          HybridRetriever.fusion, retrievers.py:239-253
  hybrid_retrieve                 HybridRetriever._aretrieve, retrievers.py:276-291

Tie order.  `filter` walks ``scores.argsort()[::-1]`` (retrievers.py:192); numpy's default
argsort is not stable, so the order among *equal* scores is implementation-defined (SURVEY.md
A.5).  tie="literal" reproduces whatever this numpy does; tie="canonical" (score desc, index
asc) is the rule the GPU path implements and is bit-compared against.
"""
from __future__ import annotations

from collections import defaultdict
from dataclasses import dataclass
from typing import Hashable, Iterable, List, Optional, Sequence

import numpy as np


@dataclass
class Item:
    idx: int                 # index of the node in the retriever's node list
    content: Hashable        # node.get_content() (the text) -- the RRF / fusion key
    score: float


def tokenize_and_remove_stopwords(tokenizer, text, stopwords):
    """retrievers.py:72-76: ``tokenizer.cut(text)`` minus stop-words and single spaces."""
    words = tokenizer.cut(text)
    return [w for w in words if w not in stopwords and w != ' ']


def canonical_order(scores: np.ndarray) -> np.ndarray:
    """Indices sorted by (score desc, index asc)."""
    scores = np.asarray(scores)
    return np.lexsort((np.arange(scores.shape[0]), -scores.astype(np.float64)))


def bm25_filter(scores: np.ndarray, top_k: int, keep_mask: Optional[np.ndarray] = None,
                tie: str = "canonical") -> List[tuple]:
    """BM25Retriever.filter (retrievers.py:191-210): descending walk, stop at the first
    score <= 0, skip nodes failing the metadata filter, collect top_k, final stable
    re-sort by score.  Returns [(idx, float(score)), ...]."""
    if tie == "literal":
        top_n = scores.argsort()[::-1]
    else:
        top_n = canonical_order(scores)
    out = []
    for ix in top_n:
        if scores[ix] <= 0:
            break
        flag = True
        if keep_mask is not None and not keep_mask[ix]:
            flag = False
        if flag:
            out.append((int(ix), float(scores[ix])))
        if len(out) == top_k:
            break
    out = sorted(out, key=lambda x: x[1], reverse=True)  # stable
    return out


def reciprocal_rank_fusion(list_of_lists: Sequence[Sequence[Item]], K: int = 60, topk: int = 256) -> List[Item]:
    """retrievers.py:256-274.  Key = content; ``rrf[content] += 1 / (rank + K)`` per occurrence,
    rank 1-based inside each list; the node kept is the LAST one seen for that content; stable
    sort by score desc (ties keep dict insertion = first-seen order); score overwritten."""
    rrf_map = defaultdict(float)
    text_to_item = {}
    for rank_list in list_of_lists:
        for rank, item in enumerate(rank_list, 1):
            content = item.content
            text_to_item[content] = item
            rrf_map[content] += 1 / (rank + K)
    sorted_items = sorted(rrf_map.items(), key=lambda x: x[1], reverse=True)
    reranked = []
    for text, score in sorted_items:
        it = text_to_item[text]
        reranked.append(Item(it.idx, it.content, score))
    topk = min(topk, len(reranked))
    return reranked[:topk]


def fusion(list_of_lists: Sequence[Sequence[Item]], topk: int = 256) -> List[Item]:
    """retrievers.py:239-253.  De-duplicate by content keeping the first occurrence, stable
    sort by the raw route score desc, truncate."""
    all_items = []
    seen = set()
    for items in list_of_lists:
        for it in items:
            if it.content not in seen:
                all_items.append(it)
                seen.add(it.content)
    all_items = sorted(all_items, key=lambda it: it.score, reverse=True)
    topk = min(len(all_items), topk)
    return all_items[:topk]


def hybrid_retrieve(retrieval_type: int, sparse_fn, dense_fn, topk: int = 256) -> List[Item]:
    """retrievers.py:276-291.  1 dense only, 2 sparse only, 3 RRF([sparse, dense])."""
    sparse_items = dense_items = None
    if retrieval_type != 1:
        sparse_items = sparse_fn()
        if retrieval_type == 2:
            return sparse_items
    if retrieval_type != 2:
        dense_items = dense_fn()
        if retrieval_type == 1:
            return dense_items
    return reciprocal_rank_fusion([sparse_items, dense_items], topk=topk)
