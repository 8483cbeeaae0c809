"""BaseRetriever with the constructor signature and the retrieve / aretrieve wrapping of llama-index-core 0.10.29
(test infrastructure; see ../../../README.md)."""
from typing import List, Optional

from ..callbacks import CallbackManager
from ..schema import NodeWithScore, QueryBundle


class BaseRetriever:
    def __init__(self, callback_manager: Optional[CallbackManager] = None, object_map: Optional[dict] = None,
                 objects: Optional[list] = None, verbose: bool = False) -> None:
        self.callback_manager = callback_manager or CallbackManager()
        if objects is not None:
            object_map = {obj.index_id: obj.obj for obj in objects}
        self.object_map = object_map or {}
        self._verbose = verbose

    def _retrieve(self, query_bundle: QueryBundle) -> List[NodeWithScore]:
        raise NotImplementedError

    async def _aretrieve(self, query_bundle: QueryBundle) -> List[NodeWithScore]:
        return self._retrieve(query_bundle)

    def retrieve(self, str_or_query_bundle) -> List[NodeWithScore]:
        qb = QueryBundle(str_or_query_bundle) if isinstance(str_or_query_bundle, str) else str_or_query_bundle
        self.callback_manager.on_event("retrieve:start", qb.query_str)
        nodes = self._retrieve(qb)
        self.callback_manager.on_event("retrieve:end", len(nodes))
        return nodes

    async def aretrieve(self, str_or_query_bundle) -> List[NodeWithScore]:
        qb = QueryBundle(str_or_query_bundle) if isinstance(str_or_query_bundle, str) else str_or_query_bundle
        self.callback_manager.on_event("retrieve:start", qb.query_str)
        nodes = await self._aretrieve(qb)
        self.callback_manager.on_event("retrieve:end", len(nodes))
        return nodes
