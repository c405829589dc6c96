"""Settings with the reference's names, defaults and environment aliases.

Mirrors the RAG 2.0 block of src/voice_agent/config.py:282-314 (``SETTINGS``,
env aliases ``RAG2_*``) and the standalone package's ``RAGConfig``
(triple-hybrid-rag/src/triple_hybrid_rag/config.py:129-207, ``rag_*`` names).
Plain dataclasses: the module-level ``SETTINGS`` singleton is mutable, as the
reference's tests rely on (tests/test_rag2_triple_hybrid.py:851-860).
"""
from __future__ import annotations

import os
from dataclasses import dataclass, fields


def _env(name, default):
    raw = os.environ.get(name)
    if raw is None:
        return default
    if isinstance(default, bool):
        return raw.strip().lower() in ("1", "true", "yes", "on")
    return type(default)(raw)


@dataclass
class Settings:
    rag2_enabled: bool = False
    rag2_graph_enabled: bool = False
    rag2_rerank_enabled: bool = True
    rag2_denoise_enabled: bool = True
    rag2_embed_dim_store: int = 1024
    rag2_embed_dim_model: int = 4096
    rag2_safety_threshold: float = 0.6
    rag2_denoise_alpha: float = 0.6
    rag2_lexical_weight: float = 0.7
    rag2_semantic_weight: float = 0.8
    rag2_graph_weight: float = 1.0
    rag2_lexical_top_k: int = 50
    rag2_semantic_top_k: int = 100
    rag2_graph_top_k: int = 50
    rag2_rerank_top_k: int = 20
    rag2_final_top_k: int = 5
    # legacy RAG 1.0 reranker knobs (src/voice_agent/config.py:255-260)
    rag_enable_reranking: bool = True
    rag_top_k_rerank: int = 5

    @classmethod
    def from_env(cls) -> "Settings":
        return cls(**{f.name: _env(f.name.upper(), f.default) for f in fields(cls)})


SETTINGS = Settings.from_env()


@dataclass
class RAGConfig:
    """Standalone-package names (``RAG_*`` env aliases)."""
    rag_lexical_weight: float = 0.7
    rag_semantic_weight: float = 0.8
    rag_graph_weight: float = 1.0
    rag_safety_threshold: float = 0.6
    rag_denoise_enabled: bool = True
    rag_denoise_alpha: float = 0.6
    rag_lexical_top_k: int = 50
    rag_semantic_top_k: int = 100
    rag_graph_top_k: int = 50
    rag_rerank_top_k: int = 20
    rag_final_top_k: int = 5
    rag_embed_dim_store: int = 1024

    @classmethod
    def from_env(cls) -> "RAGConfig":
        return cls(**{f.name: _env(f.name.upper(), f.default) for f in fields(cls)})


_settings = None


def get_settings() -> RAGConfig:
    global _settings
    if _settings is None:
        _settings = RAGConfig.from_env()
    return _settings
