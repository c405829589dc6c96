"""Query-embedding post-processing (a1) and the embedder seam.

Host forms with the reference's signatures and float32 artefacts
(src/voice_agent/rag2/embedder.py:31-68: list in, list out, one query at a
time) and the batched device form (``postproc_batch`` -> thr_embed_postproc).
The model forward itself is an external server in the reference
(embedder.py:226-241) and stays a duck type here: anything with a sync
``embed_query(str) -> List[float]``.
"""
from __future__ import annotations

import hashlib
from typing import Any, List, Sequence

import numpy as np

from ..config import SETTINGS


def normalize_l2(embedding: Sequence[float]) -> List[float]:
    vec = np.asarray(embedding, dtype=np.float32)
    length = np.linalg.norm(vec)
    if length > 0:
        vec = vec / length
    return [float(v) for v in vec]


def truncate_matryoshka(embedding: Sequence[float], target_dim: int = 1024,
                        normalize: bool = True) -> List[float]:
    head = embedding if len(embedding) <= target_dim else embedding[:target_dim]
    if normalize:
        return normalize_l2(head)
    return head if isinstance(head, list) else list(head)


def postproc_batch(full, store_dim: int):
    """[n, full_dim] float32 device tensor -> truncated + normalised, on the GPU."""
    from .. import _native
    return _native.embed_postproc(full, store_dim)


class PrecomputedEmbedder:
    """Embedder seam for corpora whose query vectors are already known (synthetic
    benchmarks, tests): text -> the registered raw model output, post-processed exactly as
    RAG2Embedder.embed_text does (truncate to the store dim, L2-normalise)."""

    def __init__(self, store_dim: int = 0):
        self.store_dim = store_dim or SETTINGS.rag2_embed_dim_store
        self._table = {}

    def register(self, text: str, raw: Sequence[float]) -> None:
        self._table[text] = list(map(float, raw))

    def embed_query(self, query: str) -> List[float]:
        raw = self._table.get(query)
        if raw is None:
            raise ValueError(f"Query embedding failed: no vector registered for {query!r}")
        return truncate_matryoshka(raw, self.store_dim, normalize=True)


class HashEmbedder:
    """Deterministic text -> vector stand-in (the reference's e2e tests seed fake embeddings
    from a text hash the same way, tests/test_rag2_e2e.py:48-63)."""

    def __init__(self, model_dim: int = 0, store_dim: int = 0):
        self.model_dim = model_dim or SETTINGS.rag2_embed_dim_model
        self.store_dim = store_dim or SETTINGS.rag2_embed_dim_store

    def embed_query(self, query: str) -> List[float]:
        seed = int.from_bytes(hashlib.sha256(query.encode()).digest()[:8], "little")
        raw = np.random.default_rng(seed).standard_normal(self.model_dim).astype(np.float32)
        return truncate_matryoshka(raw.tolist(), self.store_dim, normalize=True)


def get_rag2_embedder(**kwargs: Any):
    return HashEmbedder(**kwargs)
