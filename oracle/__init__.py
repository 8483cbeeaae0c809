"""CPU oracle for the EasyRAG coarse-ranking hot path.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the reference (/root/reference @ 2024-12-20) ships no tests, no golden
vectors and no known-answer fixtures for this path, and the libraries that hold its
arithmetic (rank-bm25==0.2.2, bm25s==0.1.7, qdrant-client==1.8.2, numpy==1.26.4; see
requirements.txt:61,75,102,113) are neither vendored nor installable here (no network).
This package therefore *restates* their published algorithms plus the reference's own glue
(src/easyrag/custom/retrievers.py) and is anchored on hand-computed known-answer cases
(tests/test_oracle_known_answers.py) and on the golden vectors it generated itself
(tests/golden/, made by tests/golden/make_golden.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  The product (easyrag_amd/) never does: it fails loudly without its HIP library.

Modules
  bm25_okapi   rank_bm25.BM25Okapi restatement            (retrievers.py:113-118,150)
  bm25s_lucene bm25s.BM25(method="lucene") restatement    (retrievers.py:107-111,150)
  dense        qdrant local-mode COSINE search + the exact fp64 dense ranking
  retrievers   the reference glue: filter / RRF / fusion / hybrid route selection
"""
from .bm25_okapi import BM25Okapi  # noqa: F401
from .bm25s_lucene import BM25SLucene  # noqa: F401
from .dense import (  # noqa: F401
    qdrant_cosine_search,
    dense_exact_scores,
    dense_exact_topk,
    to_f16_unit,
)
from .retrievers import (  # noqa: F401
    tokenize_and_remove_stopwords,
    bm25_filter,
    reciprocal_rank_fusion,
    fusion,
    hybrid_retrieve,
    canonical_order,
)
