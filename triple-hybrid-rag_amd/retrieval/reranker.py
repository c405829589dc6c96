"""Rerank surface: ``Reranker.rerank(query, results, top_k)`` and friends.

Same classes and call contract as src/voice_agent/retrieval/reranker.py
(``Qwen3VLReranker`` :48-529 with ``Reranker`` as its alias :529,
``LightweightReranker`` :532-587, ``get_reranker`` :768-797); the scorer behind
``_rerank_batch_native`` is no longer an HTTP POST to a vLLM /rerank endpoint
(:287-354) but late-interaction MaxSim on the MI355X matrix cores
(thr_maxsim), reached through the backend client.  Ordering, truncation and
failure semantics are the reference's:
  * disabled / not local / empty  -> results[:top_k or self.top_k]         (:376-377)
  * only the first 50 results are scored                                   (:383)
  * a missing score is 0.5; stable descending sort; top_k returned         (:340-348, :426-441)
  * any exception -> original order, truncated                             (:459-466)
"""
from __future__ import annotations

import logging
from dataclasses import dataclass
from typing import Any, List, Optional, Tuple

from ..config import SETTINGS
from .hybrid_search import SearchResult

log = logging.getLogger(__name__)
_DATA_WORDS = {"table", "data", "numbers", "statistics", "chart", "tabela", "dados"}


@dataclass
class RerankResult:
    search_result: SearchResult
    rerank_score: float
    original_rank: int


class Qwen3VLReranker:
    """Name kept for the seam (retrieval.py:422-427 imports it by this name); the model
    behind it here is ColBERT-style MaxSim over stored token matrices."""

    def __init__(self, model_name: Optional[str] = None, api_base: Optional[str] = None,
                 top_k: Optional[int] = None, enabled: Optional[bool] = None,
                 use_local: bool = True, client: Any = None):
        self.model_name = model_name or "maxsim-late-interaction"
        self.api_base = api_base
        self.top_k = top_k or SETTINGS.rag_top_k_rerank
        self.enabled = SETTINGS.rag_enable_reranking if enabled is None else enabled
        self.use_local = use_local
        self._client = client
        self._bound_ids: Optional[List[str]] = None

    # -- backend plumbing ---------------------------------------------------
    @property
    def client(self) -> Any:
        if self._client is None:
            from ..backend import get_supabase_client
            self._client = get_supabase_client()
        return self._client

    def bind_candidates(self, child_ids: List[str], client: Any = None) -> None:
        """RAG2Retriever._rerank knows the chunk ids of the texts it sends; binding them
        avoids resolving texts back to rows."""
        self._bound_ids = list(child_ids)
        if client is not None and hasattr(client, "maxsim_scores"):
            self._client = client

    def _prepare_document(self, result: SearchResult) -> Tuple[str, Optional[str]]:
        """reference :154-192 -- Title:/Table:/Description: lines, then the content."""
        lines = []
        if result.title:
            lines.append(f"Title: {result.title}")
        if result.is_table and result.table_context:
            lines.append(f"Table: {result.table_context}")
        if result.alt_text:
            lines.append(f"Description: {result.alt_text}")
        lines.append(result.content)
        return "\n".join(lines), None

    async def _rerank_batch_native(self, query: str, documents: List[str]) -> List[float]:
        """One batched scoring call (reference :287-354).  Raises when the GPU scorer is not
        available, which sends the caller to its per-pair fallback exactly like a 404 does."""
        ids = self._bound_ids
        self._bound_ids = None
        if ids is None or len(ids) != len(documents):
            ids = [self.client.text_to_child_id(d) for d in documents]
        known = [i for i in ids if i is not None]
        scores = iter(self.client.maxsim_scores(query, known)) if known else iter(())
        return [next(scores) if i is not None else 0.5 for i in ids]

    async def _score_pair(self, query: str, document: str, image_base64: Optional[str] = None
                          ) -> float:
        """Per-pair fallback (reference :194-285): neutral 0.5 on any error."""
        try:
            cid = self.client.text_to_child_id(document)
            return self.client.maxsim_scores(query, [cid])[0] if cid is not None else 0.5
        except Exception as exc:  # noqa: BLE001
            log.error("Error scoring pair: %s", exc)
            return 0.5

    async def rerank(self, query: str, results: List[SearchResult], top_k: Optional[int] = None
                     ) -> List[SearchResult]:
        if not self.enabled or not self.use_local or not results:
            return results[:top_k or self.top_k]
        top_k = top_k or self.top_k
        head = results[:min(len(results), 50)]
        try:
            docs = [self._prepare_document(r)[0] for r in head]
            try:
                self.bind_candidates([r.chunk_id for r in head])
                scores = await self._rerank_batch_native(query, docs)
            except Exception as exc:  # noqa: BLE001
                log.debug("Native rerank unavailable (%s), scoring pairs", exc)
                scores = [await self._score_pair(query, d, None) for d in docs]
            ranked = []
            for pos, (hit, value) in enumerate(zip(head, scores)):
                hit.rerank_score = value
                ranked.append(RerankResult(hit, value, pos))
            ranked.sort(key=lambda rr: rr.rerank_score, reverse=True)
            return [rr.search_result for rr in ranked[:top_k]]
        except Exception as exc:  # noqa: BLE001
            log.error("Reranking failed: %s", exc)
            return results[:top_k]

    def rerank_sync(self, query: str, results: List[SearchResult], top_k: Optional[int] = None
                    ) -> List[SearchResult]:
        """Synchronous variant (reference :468-526): same gate, same first-50 rule; sorts the
        scored head in place on ``rerank_score or 0``."""
        if not self.enabled or not self.use_local or not results:
            return results[:top_k or self.top_k]
        top_k = top_k or self.top_k
        head = results[:min(len(results), 50)]
        try:
            ids = [r.chunk_id for r in head]
            for hit, value in zip(head, self.client.maxsim_scores(query, ids)):
                hit.rerank_score = value
            head.sort(key=lambda h: h.rerank_score or 0, reverse=True)
            return head[:top_k]
        except Exception as exc:  # noqa: BLE001
            log.error("Sync reranking failed: %s", exc)
            return results[:top_k]


Reranker = Qwen3VLReranker


class LightweightReranker:
    """Heuristic fallback (reference :532-587): 0.5*rrf + 0.3*similarity + 0.2*term overlap,
    x1.2 for tables when the query asks for data; mutates and sorts the caller's list."""

    def __init__(self, top_k: int = 5):
        self.top_k = top_k

    async def rerank(self, query: str, results: List[SearchResult], top_k: Optional[int] = None
                     ) -> List[SearchResult]:
        top_k = top_k or self.top_k
        if not results:
            return []
        wanted = set(query.lower().split())
        for hit in results:
            have = set(hit.content.lower().split())
            overlap = len(wanted & have) / max(len(wanted), 1)
            value = hit.rrf_score * 0.5 + hit.similarity_score * 0.3 + overlap * 0.2
            if hit.is_table and wanted & _DATA_WORDS:
                value *= 1.2
            hit.rerank_score = value
        results.sort(key=lambda h: h.rerank_score or 0, reverse=True)
        return results[:top_k]


def get_reranker(use_local: bool = True, model_name: Optional[str] = None, **kwargs: Any):
    """reference :768-797 -- local => the GPU reranker; otherwise the heuristic one (the
    reference's third option, a sentence-transformers CrossEncoder, is a model download and
    not reproducible offline)."""
    if use_local:
        return Qwen3VLReranker(model_name=model_name, **kwargs)
    return LightweightReranker(**{k: v for k, v in kwargs.items() if k == "top_k"})
