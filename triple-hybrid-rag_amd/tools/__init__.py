"""Agent tool layer (the caller of the path)."""
from .crm_knowledge import search_knowledge_base_rag2  # noqa: F401
