"""Minimal node / query types with the attribute surface the reference retrievers touch.

The reference uses llama_index's ``TextNode``, ``NodeWithScore`` and ``QueryBundle``
(/root/reference/src/easyrag/custom/retrievers.py:5-18).  When llama_index is importable its
classes are re-exported unchanged; otherwise these stand-ins provide exactly what the hot path
reads: ``node.get_content()``, ``node.metadata``, ``node.node_id``, ``NodeWithScore.node/.score/
.get_content()``, ``QueryBundle.query_str/.embedding/.custom_embedding_strs``.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional

try:  # pragma: no cover - not installable in the build container
    from llama_index.core import QueryBundle  # type: ignore
    from llama_index.core.schema import NodeWithScore, TextNode  # type: ignore
    HAVE_LLAMA_INDEX = True
except Exception:
    HAVE_LLAMA_INDEX = False

    @dataclass
    class RelatedNodeInfo:
        """What llama_index stores per relationship; the hot path reads only ``node_id``."""
        node_id: str
        metadata: Dict[str, Any] = field(default_factory=dict)

    @dataclass
    class TextNode:
        text: str = ""
        metadata: Dict[str, Any] = field(default_factory=dict)
        id_: Optional[str] = None
        # embed_type 6 walks the PREVIOUS link (ref ingestion.py:36-57); keys: "PREVIOUS", "2" or 2 (llama_index's enum value)
        relationships: Dict[Any, Any] = field(default_factory=dict)

        def __post_init__(self):
            if self.id_ is None:
                self.id_ = f"node-{id(self):x}"

        @property
        def node_id(self) -> str:
            return self.id_

        def get_content(self, metadata_mode: Any = None) -> str:
            return self.text

    @dataclass
    class NodeWithScore:
        node: Any
        score: Optional[float] = None

        def get_content(self, metadata_mode: Any = None) -> str:
            return self.node.get_content()

        @property
        def node_id(self) -> str:
            return self.node.node_id

        @property
        def metadata(self) -> Dict[str, Any]:
            return self.node.metadata

        @property
        def text(self) -> str:
            return self.node.get_content()

    @dataclass
    class QueryBundle:
        query_str: str
        custom_embedding_strs: Optional[List[str]] = None
        embedding: Optional[List[float]] = None
