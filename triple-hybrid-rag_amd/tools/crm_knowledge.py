"""Tool-layer result mapping for RAG 2.0 (the step AFTER the path).

Same dict schema, refusal mapping and millisecond timings as
``_search_knowledge_base_rag2`` (src/voice_agent/tools/crm_knowledge.py:69-182), so the
agent's function-calling layer can consume the accelerated retriever unchanged.  The
reference resolves ``org_id`` from the ``rag_documents`` / ``organizations`` tables
(:89-101); here it is the registered client's tenant unless given.
"""
from __future__ import annotations

import asyncio
from typing import Any, Dict, Optional

from ..config import SETTINGS
from ..rag2.retrieval import RAG2Retriever


def search_knowledge_base_rag2(query: str, category: Optional[str] = None, limit: int = 5,
                               org_id: Optional[str] = None, retriever: Any = None
                               ) -> Dict[str, Any]:
    if retriever is None:
        if org_id is None:
            from ..backend import get_supabase_client
            org_id = get_supabase_client().org_id or "default"
        retriever = RAG2Retriever(org_id=str(org_id), graph_enabled=SETTINGS.rag2_graph_enabled)
    try:
        loop = asyncio.get_event_loop()
        if loop.is_closed():
            raise RuntimeError
    except RuntimeError:
        loop = asyncio.new_event_loop()
        asyncio.set_event_loop(loop)
    result = loop.run_until_complete(retriever.retrieve(query=query, collection=category,
                                                        top_k=limit))
    if result.refused:
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
