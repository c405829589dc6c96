"""Tool-layer result mapping for RAG 2.0 (the step AFTER the path).

Same entry points, dict schema, refusal mapping and millisecond timings as the agent tool of
src/voice_agent/tools/crm_knowledge.py: ``search_knowledge_base`` (:26-62, the dispatcher) and
``_search_knowledge_base_rag2`` (:69-182), so the function-calling layer can consume the
accelerated retriever unchanged.  The tenant is resolved the way the reference does it
(:89-101): first ``rag_documents.org_id``, then ``organizations.id``, through whatever
``get_supabase_client()`` returns -- here the registered ``GpuIndexClient``.  Pinned by
tests/golden/tool_layer.json (the reference's own function run over a fake retriever).
"""
from __future__ import annotations

import asyncio
import logging
from typing import Any, Dict, Optional

from ..backend import get_supabase_client
from ..config import SETTINGS
from ..rag2.retrieval import RAG2Retriever

log = logging.getLogger(__name__)


def _loop():
    try:
        loop = asyncio.get_event_loop()
        if loop.is_closed():
            raise RuntimeError
    except RuntimeError:
        loop = asyncio.new_event_loop()
        asyncio.set_event_loop(loop)
    return loop


def _resolve_org(supabase) -> Optional[str]:
    docs = supabase.table("rag_documents").select("org_id").limit(1).execute()
    if docs.data:
        return str(docs.data[0]["org_id"])
    orgs = supabase.table("organizations").select("id").limit(1).execute()
    if orgs.data:
        return str(orgs.data[0]["id"])
    return None


def _search_knowledge_base_rag2(query: str, category: Optional[str] = None, limit: int = 5,
                                org_id: Optional[str] = None, retriever: Any = None
                                ) -> Dict[str, Any]:
    """``retriever`` (not in the reference) injects a ready retriever instead of building one."""
    if retriever is None:
        if not org_id:
            org_id = _resolve_org(get_supabase_client())
            if org_id is None:
                return {"error": "Database error: no organization found", "query": query,
                        "category": category}
        retriever = RAG2Retriever(org_id=str(org_id), graph_enabled=SETTINGS.rag2_graph_enabled)
    result = _loop().run_until_complete(retriever.retrieve(query=query, collection=category,
                                                           top_k=limit))
    if result.refused:
        log.warning("RAG2 refused query: %s", result.refusal_reason)
        return {"success": True, "query": query, "category": category, "result_count": 0,
                "search_type": "rag2_triple_hybrid", "refused": True,
                "refusal_reason": result.refusal_reason, "results": []}
    rows = []
    for pos, ctx in enumerate(result.contexts):
        rows.append({
            "chunk_id": ctx.child_id, "parent_id": ctx.parent_id, "document_id": ctx.document_id,
            "category": category, "title": ctx.section_heading or "",
            "content": ctx.parent_text if ctx.parent_text else ctx.text,
            "source_document": None, "page": ctx.page, "chunk_index": None,
            "modality": ctx.modality, "relevance_rank": pos + 1,
            "similarity_score": round(ctx.rrf_score, 4) if ctx.rrf_score else None,
            "rerank_score": round(ctx.rerank_score, 4) if ctx.rerank_score else None,
            "ocr_confidence": None, "is_table": ctx.modality == "table", "table_context": None,
            "alt_text": None, "lexical_rank": ctx.lexical_rank, "semantic_rank": ctx.semantic_rank,
            "graph_rank": ctx.graph_rank})
    return {"success": True, "query": query, "category": category, "result_count": len(rows),
            "search_type": "rag2_triple_hybrid",
            "max_rerank_score": round(result.max_rerank_score, 4) if result.max_rerank_score else None,
            "timings_ms": {k: round(v * 1000, 2) for k, v in result.timings.items()},
            "results": rows}


def search_knowledge_base_rag2(query: str, category: Optional[str] = None, limit: int = 5,
                               org_id: Optional[str] = None, retriever: Any = None
                               ) -> Dict[str, Any]:
    return _search_knowledge_base_rag2(query, category, limit, org_id, retriever)


def search_knowledge_base(query: str, category: Optional[str] = None, limit: int = 5,
                          use_hybrid: bool = True) -> Dict[str, Any]:
    """The agent tool (crm_knowledge.py:26-62).  Only the RAG 2.0 branch is served by this
    package: with ``rag2_enabled`` off -- or when RAG 2.0 raises -- the reference goes on to its
    RAG 1.0 / ILIKE searches over tables the GPU index does not hold, and the answer here is
    the error dict the reference returns when those fail too."""
    try:
        if SETTINGS.rag2_enabled:
            return _search_knowledge_base_rag2(query, category, limit)
        raise RuntimeError("rag2_enabled is off and no RAG 1.0 tables are attached")
    except Exception as e:  # noqa: BLE001 -- the tool never raises into the agent (:51-62)
        log.error("Error searching knowledge base: %s", e)
        return {"error": f"Database error: {e}", "query": query, "category": category}
