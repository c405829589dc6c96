"""RAG 2.0 retrieval pipeline -- the drop-in ``RAG2Retriever.retrieve()``.

Behavioural mirror of src/voice_agent/rag2/retrieval.py (reference file:line in
each docstring): same constructor and ``retrieve`` signature, same result
types, same per-stage ``timings`` keys, same degrade-don't-raise error
convention, same quirks.  What changes is what answers
the calls: the backend seam (``self.supabase``) is a ``GpuIndexClient`` whose
``rpc()`` / ``table()`` run the HIP scorers on an HBM-resident index, and the
rerank seam is the MaxSim reranker instead of an HTTP cross-encoder.

This is the one-query-at-a-time surface; throughput goes through
``GpuIndex.retrieve_batch`` (index.py), which computes the same ranks with the
fusion on the device.
"""
from __future__ import annotations

import asyncio
import logging
import time
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Tuple

from ..config import SETTINGS
from .query_planner import QueryPlan, get_query_planner

log = logging.getLogger(__name__)

RRF_K = 60


@dataclass(slots=True)
class RetrievalCandidate:
    """reference retrieval.py:26-47 (slots: ~150 of these are built per call)"""
    child_id: str
    parent_id: str
    document_id: str
    text: str
    page: int
    modality: str
    lexical_rank: Optional[int] = None
    semantic_rank: Optional[int] = None
    graph_rank: Optional[int] = None
    rrf_score: float = 0.0
    parent_text: Optional[str] = None
    section_heading: Optional[str] = None
    rerank_score: Optional[float] = None


@dataclass
class RetrievalResult:
    """reference retrieval.py:50-63"""
    success: bool
    contexts: List[RetrievalCandidate]
    max_rerank_score: float = 0.0
    refused: bool = False
    refusal_reason: Optional[str] = None
    query_plan: Optional[QueryPlan] = None
    timings: Dict[str, float] = field(default_factory=dict)


_CHANNELS = (("lexical", "lexical_rank", 0.7), ("semantic", "semantic_rank", 0.8),
             ("graph", "graph_rank", 1.0))


def _rrf_key(c: RetrievalCandidate) -> float:
    return c.rrf_score


def _effective_score(c: RetrievalCandidate) -> float:
    # a rerank score of exactly 0.0 is falsy and falls through to the RRF score (:476, :489)
    return c.rerank_score or c.rrf_score


class RAG2Retriever:
    """plan -> three channels -> rank merge -> weighted RRF -> parent expansion -> rerank ->
    safety/denoise (reference retrieval.py:66-495)."""

    def __init__(self, org_id: str, embedder: Any = None, query_planner: Any = None,
                 graph_enabled: bool = False):
        # reference :79-101 -- the flag is AND-ed with the global setting (:98)
        self.org_id = org_id
        if embedder is None:
            from .embedder import get_rag2_embedder
            embedder = get_rag2_embedder()
        self.embedder = embedder
        self.query_planner = query_planner or get_query_planner()
        self.graph_enabled = graph_enabled and SETTINGS.rag2_graph_enabled
        self._supabase = None
        self._reranker = None

    @property
    def supabase(self) -> Any:
        """Backend seam (:103-108): lazily the process-wide GpuIndexClient; tests and
        callers may assign ``_supabase`` directly."""
        if self._supabase is None:
            from ..backend import get_supabase_client
            self._supabase = get_supabase_client()
        return self._supabase

    @property
    def reranker(self) -> Any:
        """(:110-116) kept for interface parity; like the reference, ``_rerank`` builds its own."""
        if self._reranker is None:
            from ..retrieval.reranker import Reranker
            self._reranker = Reranker()
        return self._reranker

    # ------------------------------------------------------------------ pipeline
    async def retrieve(self, query: str, collection: Optional[str] = None,
                       top_k: Optional[int] = None, skip_planning: bool = False,
                       skip_rerank: bool = False) -> RetrievalResult:
        """reference :118-201"""
        timings: Dict[str, float] = {}
        top_k = top_k or SETTINGS.rag2_final_top_k

        tick = time.time()
        if skip_planning:
            plan = QueryPlan(original_query=query, keywords=query.split(),
                             semantic_query_text=query)
        else:
            plan = await self.query_planner.plan_async(query, collection)
        timings["planning"] = time.time() - tick

        tick = time.time()
        candidates = await self._retrieve_candidates(plan, collection)
        timings["retrieval"] = time.time() - tick
        if not candidates:
            return RetrievalResult(success=True, contexts=[], refused=True,
                                   refusal_reason="No candidates found", query_plan=plan,
                                   timings=timings)

        tick = time.time()
        fused = self._fuse_rrf(candidates, plan.weights)
        timings["fusion"] = time.time() - tick

        # truncation happens before rerank/safety even when rerank is skipped (:177)
        tick = time.time()
        expanded = await self._expand_to_parents(fused[:SETTINGS.rag2_rerank_top_k])
        timings["expansion"] = time.time() - tick

        if not skip_rerank and SETTINGS.rag2_rerank_enabled:
            tick = time.time()
            ranked = await self._rerank(query, expanded)
            timings["rerank"] = time.time() - tick
        else:
            ranked = expanded

        tick = time.time()
        final, refused, reason, max_score = self._apply_safety(ranked, top_k)
        timings["safety"] = time.time() - tick
        return RetrievalResult(success=True, contexts=final, max_rerank_score=max_score,
                               refused=refused, refusal_reason=reason, query_plan=plan,
                               timings=timings)

    async def _retrieve_candidates(self, plan: QueryPlan, collection: Optional[str]
                                   ) -> List[RetrievalCandidate]:
        """reference :203-271 -- channels run one after the other; 1-based ranks; candidates
        keep first-sighting order; first-seen row supplies the payload; a repeated id inside
        one channel keeps its LAST rank; lexical is skipped without keywords; graph runs only
        if enabled AND the plan asks for it AND carries a cypher query."""
        merged: Dict[str, RetrievalCandidate] = {}

        def absorb(rows, rank_attr):
            # (the per-row work of a call: ~150 rows; positional construction, one dict probe)
            get, make = merged.get, RetrievalCandidate
            for rank, row in enumerate(rows, 1):
                key = row["child_id"]
                cand = get(key)
                if cand is None:
                    cand = merged[key] = make(key, row["parent_id"], row["document_id"], row["text"],
                                              row.get("page", 1), row.get("modality", "text"))
                setattr(cand, rank_attr, rank)

        # the channels are CALLED in the reference's order; their rows are absorbed in that order
        # too, but only after the semantic call: a backend that defers its read-back
        # (GpuIndexClient, ``_defer``) then has the lexical kernels running beside the dense ones
        lex_rows = []
        if plan.keywords:
            lex_rows = await self._lexical_search(keywords=plan.keywords, collection=collection,
                                                  limit=plan.lexical_top_k)
        sem_rows = await self._semantic_search(query_text=plan.semantic_query_text,
                                               collection=collection, limit=plan.semantic_top_k)
        if plan.keywords:
            absorb(lex_rows, "lexical_rank")
        absorb(sem_rows, "semantic_rank")
        if self.graph_enabled and plan.requires_graph and plan.cypher_query:
            absorb(await self._graph_search(cypher=plan.cypher_query, keywords=plan.keywords,
                                            collection=collection, limit=plan.graph_top_k),
                   "graph_rank")
        return list(merged.values())

    # ------------------------------------------------------------------ channel seams
    async def _lexical_search(self, keywords: List[str], collection: Optional[str],
                              limit: int) -> List[Dict[str, Any]]:
        """reference :273-292 -- one RPC, keywords joined by a space."""
        params = {"p_org_id": self.org_id, "p_query": " ".join(keywords), "p_limit": limit,
                  "p_collection": collection}
        if getattr(self.supabase, "defers_readback", False):
            params["_defer"] = True     # (not a reference parameter: only sent to a backend that asks)
        data = self.supabase.rpc("rag2_lexical_search", params).execute().data
        return data if data is not None else []

    async def _semantic_search(self, query_text: str, collection: Optional[str],
                               limit: int) -> List[Dict[str, Any]]:
        """reference :294-314 -- sync embed (an embed failure raises ValueError out of
        retrieve(), embedder.py:238-241), then one RPC carrying the vector as a list."""
        vector = self.embedder.embed_query(query_text)
        reply = self.supabase.rpc("rag2_semantic_search", {
            "p_org_id": self.org_id, "p_embedding": vector, "p_limit": limit,
            "p_collection": collection}).execute()
        return reply.data or []

    async def _graph_search(self, cypher: str, keywords: List[str], collection: Optional[str],
                            limit: int) -> List[Dict[str, Any]]:
        """reference :316-356 -- graph searcher -> chunk ids -> row fetch; ANY failure
        degrades to an empty channel."""
        from .graph_search import get_graph_searcher
        try:
            found = await get_graph_searcher(self.supabase).search(
                keywords=keywords, cypher_query=cypher, org_id=self.org_id, top_k=limit)
            if not found.chunk_ids:
                return []
            rows = self.supabase.table("rag_child_chunks").select(
                "id, parent_id, document_id, text, page, modality"
            ).in_("id", found.chunk_ids[:limit]).execute()
            return [{"child_id": r["id"], "parent_id": r["parent_id"],
                     "document_id": r["document_id"], "text": r["text"],
                     "page": r.get("page", 1), "modality": r.get("modality", "text")}
                    for r in rows.data]
        except Exception as exc:  # noqa: BLE001 - the reference swallows everything here
            log.warning("Graph search failed: %s", exc)
            return []

    # ------------------------------------------------------------------ fusion / expansion
    def _fuse_rrf(self, candidates: List[RetrievalCandidate], weights: Dict[str, float],
                  k: int = RRF_K) -> List[RetrievalCandidate]:
        """reference :358-376 -- float64, add order lexical -> semantic -> graph, missing
        weights default to 0.7/0.8/1.0, stable descending sort."""
        w_l, w_s, w_g = (weights.get(name, default) for name, _attr, default in _CHANNELS)
        for cand in candidates:
            total = 0.0          # (same adds in the same order as the loop over the channels)
            if cand.lexical_rank:
                total += w_l / (k + cand.lexical_rank)
            if cand.semantic_rank:
                total += w_s / (k + cand.semantic_rank)
            if cand.graph_rank:
                total += w_g / (k + cand.graph_rank)
            cand.rrf_score = total
        return sorted(candidates, key=_rrf_key, reverse=True)

    async def _expand_to_parents(self, candidates: List[RetrievalCandidate]
                                 ) -> List[RetrievalCandidate]:
        """reference :378-403 -- one fetch of the distinct parents, attach text + heading."""
        if not candidates:
            return []
        wanted = list({c.parent_id for c in candidates})
        reply = self.supabase.table("rag_parent_chunks").select(
            "id, text, section_heading").in_("id", wanted).execute()
        parents = {row["id"]: row for row in reply.data}
        for cand in candidates:
            row = parents.get(cand.parent_id)
            if row is not None:
                cand.parent_text = row["text"]
                cand.section_heading = row.get("section_heading")
        return candidates

    # ------------------------------------------------------------------ rerank / safety
    async def _rerank(self, query: str, candidates: List[RetrievalCandidate]
                      ) -> List[RetrievalCandidate]:
        """reference :405-459 -- batch scorer first; if it raises, per-pair scoring under a
        5-way semaphore (failed pairs keep their old score); stable sort on
        ``rerank_score or 0``; any outer failure returns the input order unchanged."""
        if not candidates:
            return []
        texts = [c.parent_text or c.text for c in candidates]
        try:
            from ..retrieval.reranker import Qwen3VLReranker
            scorer = Qwen3VLReranker()
            try:
                if hasattr(scorer, "bind_candidates"):
                    scorer.bind_candidates([c.child_id for c in candidates], self.supabase)
                scores = await scorer._rerank_batch_native(query, texts)
                for i, value in enumerate(scores):
                    candidates[i].rerank_score = value
            except Exception as native_err:  # noqa: BLE001
                log.debug("Native rerank unavailable (%s), using per-pair fallback", native_err)
                gate = asyncio.Semaphore(5)

                async def one(i: int, text: str):
                    async with gate:
                        return i, await scorer._score_pair(query, text, None)

                outcomes = await asyncio.gather(*(one(i, t) for i, t in enumerate(texts)),
                                                return_exceptions=True)
                for outcome in outcomes:
                    if isinstance(outcome, BaseException):
                        log.warning("Rerank error: %s", outcome)
                    elif isinstance(outcome, tuple) and len(outcome) == 2:
                        candidates[outcome[0]].rerank_score = outcome[1]
            return sorted(candidates, key=lambda c: c.rerank_score or 0, reverse=True)
        except Exception as exc:  # noqa: BLE001
            log.warning("Reranking failed: %s", exc)
            return candidates

    def _apply_safety(self, candidates: List[RetrievalCandidate], top_k: int
                      ) -> Tuple[List[RetrievalCandidate], bool, Optional[str], float]:
        """reference :461-495 -- refuse below the threshold; keep >= alpha * max; no re-sort."""
        if not candidates:
            return [], True, "No candidates after reranking", 0.0
        best = max(_effective_score(c) for c in candidates)
        threshold = SETTINGS.rag2_safety_threshold
        if best < threshold:
            return [], True, f"Max score {best:.2f} below threshold {threshold}", best
        floor = SETTINGS.rag2_denoise_alpha * best
        kept = [c for c in candidates if _effective_score(c) >= floor]
        return kept[:top_k], False, None, best


async def retrieve(org_id: str, query: str, **kwargs: Any) -> RetrievalResult:
    """reference :498-505 -- a fresh retriever per call."""
    return await RAG2Retriever(org_id=org_id).retrieve(query, **kwargs)
