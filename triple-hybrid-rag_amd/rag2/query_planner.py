"""Query plan record + planner seam.

``QueryPlan`` is the reference's dataclass field for field
(src/voice_agent/rag2/query_planner.py:23-50).  The reference's planner is a
GPT call (:130-190), an external API and out of scope; ``QueryPlanner`` here
keeps the interface and implements the reference's own failure path -- the
whitespace-split plan (:180-190) -- so ``plan_async`` always answers.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional

from ..config import SETTINGS


@dataclass
class QueryPlan:
    original_query: str
    keywords: List[str] = field(default_factory=list)
    lexical_top_k: int = 50
    semantic_query_text: str = ""
    semantic_top_k: int = 100
    cypher_query: Optional[str] = None
    graph_top_k: int = 50
    weights: Dict[str, float] = field(default_factory=lambda: {
        "lexical": 0.7, "semantic": 0.8, "graph": 1.0})
    intent: str = "general"
    requires_graph: bool = False


class QueryPlanner:
    """Rule-based stand-in with the reference planner's interface."""

    def __init__(self, graph: bool = False, **_kw: Any):
        self.graph = graph

    def plan(self, query: str, collection: Optional[str] = None) -> QueryPlan:
        return QueryPlan(
            original_query=query, keywords=query.split(), semantic_query_text=query,
            lexical_top_k=SETTINGS.rag2_lexical_top_k, semantic_top_k=SETTINGS.rag2_semantic_top_k,
            graph_top_k=SETTINGS.rag2_graph_top_k,
            weights={"lexical": SETTINGS.rag2_lexical_weight,
                     "semantic": SETTINGS.rag2_semantic_weight,
                     "graph": SETTINGS.rag2_graph_weight},
            requires_graph=self.graph,
            cypher_query="MATCH (e:Entity)-[:MENTIONED_IN]->(c:Chunk) RETURN c" if self.graph else None)

    async def plan_async(self, query: str, collection: Optional[str] = None) -> QueryPlan:
        return self.plan(query, collection)


def get_query_planner(**kwargs: Any) -> QueryPlanner:
    return QueryPlanner(**kwargs)
