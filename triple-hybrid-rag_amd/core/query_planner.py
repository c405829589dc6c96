"""Rule-based query planner of the standalone package.

Behaviour of ``QueryPlanner._simple_plan`` / ``_extract_keywords``
(triple-hybrid-rag/src/triple_hybrid_rag/core/query_planner.py:157-227): stop-word keyword
extraction (words compared in lower case, dropped when a stop word or shorter than 3
characters, then stripped of surrounding punctuation, de-duplicated in order), graph need
detected from relational phrases, intent from the opening words.  The GPT planner around it
(:60-155) is an external API call and out of scope.
"""
from __future__ import annotations

from typing import List, Optional

from ..config import RAGConfig, get_settings
from .types import QueryPlan

_STOP = frozenset("""a an the is are was were be been being have has had do does did will would
could should may might must shall can need dare ought used to of in for on with at by from as
into through during before after above below between under again further then once here there
when where why how all each few more most other some such no nor not only own same so than too
very just and but if or because until while what which who whom this that these those am i me my
myself we our ours ourselves you your yours""".split())
_GRAPH_HINTS = ("relationship", "related", "connected", "between", "who", "what company",
                "which organization", "works for", "belongs to", "part of")
_PUNCT = ".,!?;:'\"()[]{}"


class QueryPlanner:
    def __init__(self, config: Optional[RAGConfig] = None):
        self.config = config or get_settings()

    def _extract_keywords(self, query: str) -> List[str]:
        kept = [w.strip(_PUNCT) for w in query.lower().split() if w not in _STOP and len(w) > 2]
        return list(dict.fromkeys(kept))

    def _simple_plan(self, query: str) -> QueryPlan:
        low = query.lower()
        needs_graph = any(h in low for h in _GRAPH_HINTS)
        if low.startswith(("what is", "what are", "define")):
            intent = "factual"
        elif low.startswith(("how do", "how to", "how can")):
            intent = "procedural"
        elif "difference" in low or "compare" in low:
            intent = "comparative"
        elif needs_graph:
            intent = "relational"
        else:
            intent = "general"
        cfg = self.config
        return QueryPlan(
            original_query=query, keywords=self._extract_keywords(query), semantic_query_text=query,
            cypher_query=None, requires_graph=needs_graph, intent=intent,
            weights={"lexical": cfg.rag_lexical_weight, "semantic": cfg.rag_semantic_weight,
                     "graph": cfg.rag_graph_weight if needs_graph else 0.5},
            lexical_top_k=cfg.rag_lexical_top_k, semantic_top_k=cfg.rag_semantic_top_k,
            graph_top_k=cfg.rag_graph_top_k)

    def plan(self, query: str, collection: Optional[str] = None) -> QueryPlan:
        return self._simple_plan(query)

    async def plan_async(self, query: str, collection: Optional[str] = None) -> QueryPlan:
        return self._simple_plan(query)
