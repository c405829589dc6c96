"""Standalone ``RRFFusion`` (second fusion variant of the reference).

Behaviour of triple-hybrid-rag/src/triple_hybrid_rag/core/fusion.py:
  fuse (:52-165)            per-channel score table keyed by str(chunk_id) (a later duplicate
                            overwrites, :167-185); merge lexical -> semantic -> graph adding
                            the table value once per OCCURRENCE; sort; safety threshold on the
                            best channel score (:187-216); percentile denoise when >= 3 rows
                            (:218-247); [:top_k]
  fuse_two_channels (:249-292), normalize_scores (:294-318)
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np

from ..config import RAGConfig, get_settings
from .types import QueryPlan, SearchChannel, SearchResult

RRF_K = 60


class RRFFusion:
    def __init__(self, config: Optional[RAGConfig] = None):
        self.config = config or get_settings()
        self.default_weights = {"lexical": self.config.rag_lexical_weight,
                                "semantic": self.config.rag_semantic_weight,
                                "graph": self.config.rag_graph_weight}
        self.safety_threshold = self.config.rag_safety_threshold
        self.denoise_enabled = self.config.rag_denoise_enabled
        self.denoise_alpha = self.config.rag_denoise_alpha

    @staticmethod
    def _compute_rrf_scores(results: List[SearchResult], weight: float) -> Dict[str, float]:
        return {str(hit.chunk_id): weight * (1.0 / (RRF_K + rank))
                for rank, hit in enumerate(results, start=1)}

    def fuse(self, lexical_results: List[SearchResult], semantic_results: List[SearchResult],
             graph_results: List[SearchResult], query_plan: Optional[QueryPlan] = None,
             top_k: Optional[int] = None) -> List[SearchResult]:
        weights = query_plan.weights if query_plan else self.default_weights
        channels = (
            (lexical_results, "lexical", "lexical_score", SearchChannel.LEXICAL),
            (semantic_results, "semantic", "semantic_score", SearchChannel.SEMANTIC),
            (graph_results, "graph", "graph_score", SearchChannel.GRAPH),
        )
        merged: Dict[str, dict] = {}
        for hits, name, score_attr, channel in channels:
            table = self._compute_rrf_scores(hits, weights.get(name, self.default_weights[name]))
            for hit in hits:
                key = str(hit.chunk_id)
                slot = merged.setdefault(key, {"rrf": 0.0, "lexical_score": 0.0,
                                               "semantic_score": 0.0, "graph_score": 0.0,
                                               "hit": hit, "sources": set()})
                slot["rrf"] += table.get(key, 0.0)
                slot[score_attr] = getattr(hit, score_attr)
                slot["sources"].add(channel)
        fused = []
        for slot in merged.values():
            hit = slot["hit"]
            hit.rrf_score = hit.final_score = slot["rrf"]
            hit.lexical_score = slot["lexical_score"]
            hit.semantic_score = slot["semantic_score"]
            hit.graph_score = slot["graph_score"]
            hit.metadata["source_channels"] = [c.value for c in slot["sources"]]
            fused.append(hit)
        fused.sort(key=lambda h: h.rrf_score, reverse=True)
        fused = self._apply_safety_threshold(fused)
        if self.denoise_enabled:
            fused = self._apply_conformal_denoising(fused)
        return fused[:top_k] if top_k else fused

    def _apply_safety_threshold(self, results: List[SearchResult]) -> List[SearchResult]:
        if self.safety_threshold <= 0:
            return results
        return [h for h in results
                if max(h.semantic_score or 0.0, h.lexical_score or 0.0, h.graph_score or 0.0)
                >= self.safety_threshold]

    def _apply_conformal_denoising(self, results: List[SearchResult]) -> List[SearchResult]:
        if len(results) < 3:
            return results
        cut = np.percentile(np.array([h.rrf_score for h in results]),
                            (1 - self.denoise_alpha) * 100)
        return [h for h in results if h.rrf_score >= cut]

    def fuse_two_channels(self, results_a: List[SearchResult], results_b: List[SearchResult],
                          weight_a: float = 1.0, weight_b: float = 1.0,
                          top_k: Optional[int] = None) -> List[SearchResult]:
        table_a = self._compute_rrf_scores(results_a, weight_a)
        table_b = self._compute_rrf_scores(results_b, weight_b)
        merged: Dict[str, list] = {}
        for hit in results_a:
            key = str(hit.chunk_id)
            merged[key] = [table_a.get(key, 0.0), hit]
        for hit in results_b:
            key = str(hit.chunk_id)
            if key in merged:
                merged[key][0] += table_b.get(key, 0.0)
            else:
                merged[key] = [table_b.get(key, 0.0), hit]
        ordered = sorted(merged.values(), key=lambda pair: pair[0], reverse=True)
        for value, hit in ordered:
            hit.rrf_score = hit.final_score = value
        out = [hit for _, hit in ordered]
        return out[:top_k] if top_k else out

    def fuse_batch(self, lexical_ids, semantic_ids, graph_ids, lexical_scores=None, semantic_scores=None,
                   graph_scores=None, weights: Optional[Dict[str, float]] = None,
                   top_k: Optional[int] = None, normalize: bool = False):
        """``fuse`` for a BATCH of queries on the device (thr_rrf_fuse_standalone + thr_fuse_post):
        id tensors int64 [nq, n_c] (best first, -1 padded; None = channel absent), channel score
        tensors float64 [nq, n_c] (needed for the safety threshold).  Same arithmetic as ``fuse``
        row by row: table value weight * (1 / (60 + rank)), safety threshold, percentile denoise,
        [:top_k]; ``normalize`` = normalize_scores on the result.  -> (ids, scores, counts)."""
        from .. import _native as N
        w = dict(self.default_weights)
        w.update(weights or {})
        width = sum(t.shape[1] for t in (lexical_ids, semantic_ids, graph_ids) if t is not None)
        ids, sc, rk, cnt = N.rrf_fuse_standalone(lexical_ids, semantic_ids, graph_ids, width,
                                                 w["lexical"], w["semantic"], w["graph"])
        q = None
        if self.denoise_enabled:
            q = float(np.true_divide((1 - self.denoise_alpha) * 100, 100))
        return N.fuse_post(ids, sc, rk, cnt, (lexical_scores, semantic_scores, graph_scores),
                           self.safety_threshold, q, normalize, top_k or 0)

    def normalize_scores(self, results: List[SearchResult], score_field: str = "final_score"
                         ) -> List[SearchResult]:
        if not results:
            return results
        values = [getattr(h, score_field, 0.0) for h in results]
        lo, hi = min(values), max(values)
        for h in results:
            setattr(h, score_field,
                    1.0 if hi == lo else (getattr(h, score_field, 0.0) - lo) / (hi - lo))
        return results
