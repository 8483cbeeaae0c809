"""Stand-ins with the attribute surface of llama-index-core 0.10.29's schema types (test infrastructure)."""
from dataclasses import dataclass, field
from enum import Enum
from typing import Any, Dict, List, Optional


class NodeRelationship(str, Enum):
    SOURCE = "1"
    PREVIOUS = "2"
    NEXT = "3"
    PARENT = "4"
    CHILD = "5"


@dataclass
class RelatedNodeInfo:
    node_id: str


@dataclass
class TextNode:
    text: str = ""
    metadata: Dict[str, Any] = field(default_factory=dict)
    id_: Optional[str] = None
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
