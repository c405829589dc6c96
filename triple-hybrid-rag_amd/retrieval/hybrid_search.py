"""``SearchResult`` (what ``rerank()`` consumes) and the legacy unweighted RRF.

Mirrors src/voice_agent/retrieval/hybrid_search.py:52-77 (record) and :460-501
(``_rrf_fusion``: 1/(k + rank0 + 1), first-seen object kept, best per-channel
scores kept, stable sort).  The RAG 1.0 searcher around them is a "next" row
(SURVEY.md section 8f.4).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional


@dataclass
class SearchResult:
    chunk_id: str
    content: str
    modality: str
    source_document: str
    page: int
    chunk_index: int
    similarity_score: float = 0.0
    bm25_score: float = 0.0
    rrf_score: float = 0.0
    rerank_score: Optional[float] = None
    ocr_confidence: Optional[float] = None
    is_table: bool = False
    table_context: Optional[str] = None
    alt_text: Optional[str] = None
    category: Optional[str] = None
    title: Optional[str] = None
    retrieval_method: str = ""


def rrf_fusion(results_lists: List[List[SearchResult]], k: int = 60) -> List[SearchResult]:
    kept: Dict[str, SearchResult] = {}
    score: Dict[str, float] = {}
    for ranked in results_lists:
        for pos, hit in enumerate(ranked):
            first = kept.setdefault(hit.chunk_id, hit)
            score[hit.chunk_id] = score.get(hit.chunk_id, 0.0) + 1.0 / (k + pos + 1)
            first.similarity_score = max(first.similarity_score, hit.similarity_score)
            first.bm25_score = max(first.bm25_score, hit.bm25_score)
    for cid, hit in kept.items():
        hit.rrf_score = score[cid]
        hit.retrieval_method = "hybrid"
    fused = list(kept.values())
    fused.sort(key=lambda h: h.rrf_score, reverse=True)
    return fused
