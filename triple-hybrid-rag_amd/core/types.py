"""Record types of the standalone package that the fusion path touches
(triple-hybrid-rag/src/triple_hybrid_rag/types.py:273-340)."""
from __future__ import annotations

from dataclasses import dataclass, field
from enum import Enum
from typing import Any, Dict, List, Optional
from uuid import UUID, uuid4


class SearchChannel(str, Enum):
    LEXICAL = "lexical"
    SEMANTIC = "semantic"
    GRAPH = "graph"


@dataclass
class SearchResult:
    chunk_id: UUID = field(default_factory=uuid4)
    parent_id: UUID = field(default_factory=uuid4)
    document_id: UUID = field(default_factory=uuid4)
    text: str = ""
    page: Optional[int] = None
    modality: str = "text"
    lexical_score: float = 0.0
    semantic_score: float = 0.0
    graph_score: float = 0.0
    rrf_score: float = 0.0
    rerank_score: Optional[float] = None
    final_score: float = 0.0
    source_channel: SearchChannel = SearchChannel.SEMANTIC
    title: Optional[str] = None
    collection: Optional[str] = None
    is_table: bool = False
    table_context: Optional[str] = None
    alt_text: Optional[str] = None
    image_data: Optional[bytes] = None
    metadata: Dict[str, Any] = field(default_factory=dict)
    parent_chunk: Optional[Any] = None


@dataclass
class QueryPlan:
    original_query: str = ""
    keywords: List[str] = field(default_factory=list)
    lexical_top_k: int = 50
    semantic_query_text: str = ""
    semantic_top_k: int = 100
    cypher_query: Optional[str] = None
    graph_top_k: int = 50
    requires_graph: bool = False
    weights: Dict[str, float] = field(default_factory=lambda: {
        "lexical": 0.7, "semantic": 0.8, "graph": 1.0})
    intent: str = "general"
