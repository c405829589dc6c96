"""Legacy RAG 1.0 record type + the rerank surface (drop-in for ``voice_agent.retrieval``)."""
from .hybrid_search import HybridSearcher, SearchConfig, SearchResult, rrf_fusion  # noqa: F401
from .reranker import (LightweightReranker, Qwen3VLReranker, Reranker, RerankResult,  # noqa: F401
                       get_reranker)
