"""Legacy RAG 1.0 hybrid search on the same kernels.

Mirrors src/voice_agent/retrieval/hybrid_search.py: ``SearchResult`` (:52-77, what
``rerank()`` consumes), ``SearchConfig`` (:26-49), ``HybridSearcher.search`` (:114-199:
embed -> vector + lexical channels -> unweighted RRF ``1/(k + rank0 + 1)`` -> filters ->
``[:top_k]``) and ``_rrf_fusion`` (:460-501: first-seen object kept, best per-channel scores
kept, stable sort).  The two channel RPCs (``kb_chunks_vector_search``,
``kb_chunks_fts_pt``) are answered by ``GpuIndexClient`` with thr_dense_topk / thr_bm25_topk, the
image channel (``_image_search`` :423-458, RPC ``kb_chunks_image_search``) by a second dense index
over the chunks' image vectors; the ILIKE / np.dot client-side fallbacks (:260-421) have no
counterpart (there is no second, slower path to fall back to).
"""
from __future__ import annotations

import asyncio
import logging
from dataclasses import dataclass
from typing import Any, Dict, List, Optional

log = logging.getLogger(__name__)


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


@dataclass
class SearchConfig:
    use_hybrid: bool = True
    use_vector: bool = True
    use_bm25: bool = True
    use_image_search: bool = False
    top_k_retrieve: int = 50
    top_k_image: int = 3
    top_k_final: int = 10
    rrf_k: int = 60
    fts_language: str = "portuguese"
    category_filter: Optional[str] = None
    source_filter: Optional[str] = None
    min_similarity: float = 0.0


def _row_to_result(row: Dict[str, Any], method: str) -> SearchResult:
    return SearchResult(
        chunk_id=row["id"], content=row["content"], modality=row["modality"],
        source_document=row["source_document"], page=row.get("page") or 1,
        chunk_index=row.get("chunk_index") or 0,
        similarity_score=row.get("similarity", 0.0) if method == "vector" else 0.0,
        bm25_score=row.get("rank", 0.0) if method == "bm25" else 0.0,
        ocr_confidence=row.get("ocr_confidence"), is_table=row.get("is_table", False),
        table_context=row.get("table_context"), alt_text=row.get("alt_text"),
        retrieval_method=method)


class HybridSearcher:
    def __init__(self, org_id: str, embedder: Any = None, config: Optional[SearchConfig] = None):
        self.org_id = org_id
        self.embedder = embedder
        self.config = config or SearchConfig()
        self._supabase = None

    @property
    def supabase(self) -> Any:
        if self._supabase is None:
            from ..backend import get_supabase_client
            self._supabase = get_supabase_client()
        return self._supabase

    def _with(self, client: Any) -> "HybridSearcher":
        """Bind a backend client explicitly (tests; multi-index processes)."""
        self._supabase = client
        return self

    async def search(self, query: str, top_k: Optional[int] = None, category: Optional[str] = None,
                     source_document: Optional[str] = None) -> List[SearchResult]:
        top_k = top_k or self.config.top_k_final
        embedded = self.embedder.embed_query(query)
        if asyncio.iscoroutine(embedded):
            embedded = await embedded
        # the RAG 1.0 embedder returns (text_embedding, image_embedding) (:136)
        text_vec = embedded[0] if isinstance(embedded, tuple) else embedded
        image_vec = embedded[1] if isinstance(embedded, tuple) and len(embedded) > 1 else None
        jobs = []
        if self.config.use_vector:
            jobs.append(self._vector_search(text_vec, category, source_document))
        if self.config.use_bm25:
            jobs.append(self._bm25_search(query, category, source_document))
        if self.config.use_image_search and image_vec:
            jobs.append(self._image_search(image_vec))
        lists = await asyncio.gather(*jobs)
        if self.config.use_hybrid and len(lists) > 1:
            combined = self._rrf_fusion(list(lists))
        elif lists:
            combined = lists[0]
        else:
            combined = []
        return self._apply_filters(combined, category, source_document)[:top_k]

    async def _vector_search(self, embedding, category=None, source_document=None):
        reply = self.supabase.rpc("kb_chunks_vector_search", {
            "p_org_id": self.org_id, "p_embedding": embedding,
            "p_limit": self.config.top_k_retrieve, "p_category": category,
            "p_source_document": source_document}).execute()
        return [_row_to_result(r, "vector") for r in reply.data]

    async def _bm25_search(self, query: str, category=None, source_document=None):
        reply = self.supabase.rpc("kb_chunks_fts_pt", {
            "p_org_id": self.org_id, "p_query": query,
            "p_limit": self.config.top_k_retrieve}).execute()
        return [_row_to_result(r, "bm25") for r in reply.data]

    async def _image_search(self, image_embedding):
        """hybrid_search.py:423-458: top ``top_k_image`` chunks by image-vector cosine; a failure
        of the channel degrades to an empty list, as in the reference."""
        try:
            reply = self.supabase.rpc("kb_chunks_image_search", {
                "p_org_id": self.org_id, "p_image_embedding": image_embedding,
                "p_limit": self.config.top_k_image}).execute()
            return [SearchResult(chunk_id=r["id"], content=r["content"], modality=r["modality"],
                                 source_document=r["source_document"], page=r.get("page") or 1,
                                 chunk_index=0, similarity_score=r.get("similarity", 0.0),
                                 alt_text=r.get("alt_text"), retrieval_method="image")
                    for r in reply.data]
        except Exception as e:  # noqa: BLE001 -- the reference logs and returns []
            log.error("Image search failed: %s", e)
            return []

    def _rrf_fusion(self, results_lists: List[List[SearchResult]]) -> List[SearchResult]:
        return rrf_fusion(results_lists, self.config.rrf_k)

    def _apply_filters(self, results, category=None, source_document=None):
        kept = results
        if self.config.min_similarity > 0:
            kept = [r for r in kept
                    if r.similarity_score >= self.config.min_similarity or r.bm25_score > 0]
        if category:
            kept = [r for r in kept if r.category == category]
        if source_document:
            kept = [r for r in kept if r.source_document == source_document]
        return kept
