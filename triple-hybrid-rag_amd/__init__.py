"""MI355X-native retrieval hot path of triple-hybrid-rag (RAG 2.0).

Drop-in for the reference's ``retrieve()`` / ``rerank()`` surface with the
scorers as hand-written HIP kernels for gfx950 behind ``include/thr_hip.h``.
Import as ``triple_hybrid_rag_amd`` (see the alias module at the repo root).
"""
__version__ = "0.1.0"

from . import _build, _native, synth  # noqa: F401
from ._native import NativeError  # noqa: F401
from .index import BatchResult, GpuIndex  # noqa: F401
from . import backend, config, core, rag2, retrieval  # noqa: F401,E402
from .backend import CorpusStore, GpuIndexClient  # noqa: F401,E402
from .rag2.retrieval import RAG2Retriever, RetrievalCandidate, RetrievalResult, retrieve  # noqa: F401,E402
from .retrieval.reranker import Reranker, get_reranker  # noqa: F401,E402
