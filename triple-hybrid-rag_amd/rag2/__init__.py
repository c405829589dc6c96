"""RAG 2.0 orchestration layer (drop-in for ``voice_agent.rag2``)."""
from .embedder import normalize_l2, truncate_matryoshka  # noqa: F401
from .query_planner import QueryPlan, QueryPlanner, get_query_planner  # noqa: F401
from .retrieval import RAG2Retriever, RetrievalCandidate, RetrievalResult, retrieve  # noqa: F401
