"""Graph channel seam: ``GraphSearcher.search`` over the GPU-resident entity graph.

Interface of src/voice_agent/rag2/graph_search.py:274-318 (GraphSearcher,
GraphSearchResult, get_graph_searcher).  The reference asks PuppyGraph (Cypher
over HTTP) or walks three SQL tables (:141-271) and returns an unordered SET of
chunk ids; here keywords pick seed entities the way the SQL fallback does
(case-insensitive substring match on the entity name, first 5 keywords,
``limit // len(keywords)`` entities each, :151-176) and the bounded BFS + mention
scoring runs in thr_graph_topk, so ``chunk_ids`` comes back ordered by
(score desc, chunk asc).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional


@dataclass
class GraphNode:
    id: str
    label: str
    properties: Dict[str, Any] = field(default_factory=dict)

    def __hash__(self) -> int:
        return hash(self.id)


@dataclass
class GraphEdge:
    source_id: str
    target_id: str
    relationship: str
    properties: Dict[str, Any] = field(default_factory=dict)
    confidence: float = 1.0


@dataclass
class GraphSearchResult:
    nodes: List[GraphNode]
    edges: List[GraphEdge]
    paths: List[List[str]]
    chunk_ids: List[str]
    source: str


class GraphSearcher:
    def __init__(self, supabase_client: Any, puppygraph_url: Optional[str] = None, hops: int = 2):
        self.db = supabase_client
        self.hops = hops

    async def search(self, keywords: List[str], cypher_query: Optional[str], org_id: str,
                     top_k: int = 20) -> GraphSearchResult:
        client = self.db
        if not hasattr(client, "graph_chunks"):
            raise RuntimeError("backend has no GPU graph index")
        seeds = client.find_entities(keywords, limit=top_k)
        if not seeds:
            return GraphSearchResult([], [], [], [], "hip_csr")
        chunk_ids = client.graph_chunks(seeds, top_k, self.hops)
        nodes = [GraphNode(id=str(e), label="entity", properties={"name": client.entity_name(e)})
                 for e in seeds]
        return GraphSearchResult(nodes=nodes[:top_k], edges=[], paths=[], chunk_ids=chunk_ids,
                                 source="hip_csr")

    async def close(self) -> None:
        return None


def get_graph_searcher(supabase_client: Any) -> GraphSearcher:
    return GraphSearcher(supabase_client=supabase_client)
