"""MI355X-native retrieval hot path of triple-hybrid-rag (RAG 2.0).

Drop-in for the reference's ``retrieve()`` / ``rerank()`` surface with the
scorers as hand-written HIP kernels for gfx950 behind ``include/thr_hip.h``.
Import as ``triple_hybrid_rag_amd`` (see the alias module at the repo root).
"""
__version__ = "0.1.0"

from . import _build, _native, synth  # noqa: F401
from ._native import NativeError  # noqa: F401
from .index import BatchResult, GpuIndex  # noqa: F401
