"""Standalone-package surface (drop-in for ``triple_hybrid_rag.core``)."""
from .fusion import RRFFusion  # noqa: F401
from .types import QueryPlan, SearchChannel, SearchResult  # noqa: F401
from .query_planner import QueryPlanner  # noqa: F401,E402
