"""Backend seam: a Supabase-shaped client answered by the GPU index.

The reference's retriever talks to PostgREST through four calls (SURVEY.md
section 8b; src/voice_agent/rag2/retrieval.py:282-290, 304-312, 339-341, 390-392):

    client.rpc("rag2_lexical_search",  {p_org_id, p_query, p_limit, p_collection}).execute().data
    client.rpc("rag2_semantic_search", {p_org_id, p_embedding, p_limit, p_collection}).execute().data
    client.table("rag_child_chunks").select(...).in_("id", ids).execute().data
    client.table("rag_parent_chunks").select(...).in_("id", ids).execute().data

``GpuIndexClient`` offers exactly that duck type.  The two RPCs run the HIP
scorers (thr_bm25_topk / thr_dense_topk + exhaustive rescue) on a ``GpuIndex``;
the table fetches are answered from ``CorpusStore``, the host-side row payloads
(ids, text, page, modality, parents) that the SQL rows would carry.
Rows come back in REQUEST order from ``in_()`` (PostgreSQL's order there is
unspecified; the reference ranks graph hits by that order, Appendix A.7).
"""
from __future__ import annotations

import collections.abc
import re
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _native as N
from .index import GpuIndex

_TOKEN = re.compile(r"[\w]+", re.UNICODE)


def tokenize(text: str) -> List[str]:
    """Lower-cased word tokens (the reference leaves tokenisation to PostgreSQL's
    'portuguese' text-search configuration, which is not reproduced: no stemming)."""
    return _TOKEN.findall(text.lower())


@dataclass
class CorpusStore:
    child_ids: List[str]
    parent_ids: List[str]
    document_ids: List[str]
    texts: List[str]
    pages: List[int]
    modalities: List[str]
    parents: Dict[str, Dict[str, Any]] = field(default_factory=dict)  # id -> row
    collections: Optional[List[Optional[str]]] = None
    vocab: Dict[str, int] = field(default_factory=dict)               # term -> id
    entity_names: List[str] = field(default_factory=list)
    doc_base: int = 0

    def __post_init__(self):
        self._row_of = {cid: i for i, cid in enumerate(self.child_ids)}

    def row_index(self, child_id: str) -> Optional[int]:
        return self._row_of.get(child_id)

    def child_row(self, i: int) -> Dict[str, Any]:
        return {"id": self.child_ids[i], "parent_id": self.parent_ids[i],
                "document_id": self.document_ids[i], "text": self.texts[i],
                "page": self.pages[i], "modality": self.modalities[i]}

    def result_row(self, i: int) -> Dict[str, Any]:
        row = self.child_row(i)
        row["child_id"] = row.pop("id")
        return row

    @classmethod
    def synthetic(cls, n: int, doc_base: int = 0, children_per_parent: int = 4,
                  vocab_size: int = 0, n_entities: int = 0) -> "CorpusStore":
        """Row payloads for a synthetic corpus: ids derive from the global doc index."""
        g = [doc_base + i for i in range(n)]
        parents = {f"p{j}": {"id": f"p{j}", "text": f"parent text {j}",
                             "section_heading": f"Section {j}"}
                   for j in sorted({x // children_per_parent for x in g})}
        return cls(child_ids=[f"c{x}" for x in g],
                   parent_ids=[f"p{x // children_per_parent}" for x in g],
                   document_ids=[f"d{x // 64}" for x in g], texts=[f"chunk {x}" for x in g],
                   pages=[x % 9 + 1 for x in g], modalities=["text"] * n, parents=parents,
                   vocab={f"t{t}": t for t in range(vocab_size)},
                   entity_names=[f"entity{e}" for e in range(n_entities)], doc_base=doc_base)


class _Reply:
    def __init__(self, data):
        self.data = data

    def execute(self):
        return self


class LazyRows(collections.abc.Sequence):
    """Rows of an RPC whose kernels are in flight: the read-back happens when the rows are first
    looked at, so the caller can issue its next RPC in the meantime.  A failure of the deferred
    part (the read-back is where an asynchronous HIP error of the lexical kernels surfaces) RAISES
    at that first look and again on every later one -- the reference's ``_lexical_search`` has no
    try/except either (retrieval.py:273-292: only the graph channel swallows errors), so a failing
    lexical RPC propagates out of ``retrieve()``; a GPU fault never becomes an empty channel.
    Not a ``list`` subclass: every consumer goes through the Sequence protocol, so nothing can
    read an unfilled list storage behind the fetch (``json.dumps(rows.materialize())`` for C-level
    consumers that insist on a real list)."""

    __slots__ = ("_fetch", "_rows", "_error")

    def __init__(self, fetch):
        self._fetch = fetch
        self._rows = None
        self._error = None

    def materialize(self) -> list:
        """The rows as a plain list (fetched once; a failed fetch re-raises its exception)."""
        if self._error is not None:
            raise self._error
        if self._rows is None:
            fetch, self._fetch = self._fetch, None
            try:
                self._rows = list(fetch())
            except BaseException as exc:
                self._error = exc
                raise
        return self._rows

    _ready = materialize

    def __iter__(self):
        return iter(self.materialize())

    def __len__(self):
        return len(self.materialize())

    def __getitem__(self, i):
        return self.materialize()[i]

    def __eq__(self, other):
        if isinstance(other, LazyRows):
            other = other.materialize()
        return self.materialize() == other

    def __add__(self, other):
        return self.materialize() + list(other)

    def __radd__(self, other):
        return list(other) + self.materialize()

    def __repr__(self):
        return f"LazyRows({'pending' if self._rows is None and self._error is None else self._rows!r})"

    __hash__ = None


class _TableQuery:
    def __init__(self, fetch):
        self._fetch = fetch
        self._ids: Optional[List[Any]] = None
        self._limit: Optional[int] = None

    def select(self, *_cols, **_kw):
        return self

    def eq(self, *_a, **_kw):
        return self

    def limit(self, n: int):
        self._limit = n
        return self

    def in_(self, column: str, values: Sequence[Any]):
        if column != "id":
            raise ValueError("only id lookups are served from the GPU index store")
        self._ids = list(values)
        return self

    def execute(self):
        rows = self._fetch(self._ids or [])
        return _Reply(rows[: self._limit] if self._limit is not None else rows)


class GpuIndexClient:
    """Supabase-shaped facade over (GpuIndex, CorpusStore)."""

    defers_readback = True   # rag2_lexical_search honours ``_defer`` (see _lexical)
    sets_collections = True  # derives the index's per-doc collection ids from the store's rows

    def __init__(self, index: GpuIndex, store: CorpusStore, org_id: Optional[str] = None,
                 token_embedder: Any = None, lexical_and: bool = False,
                 image_index: Optional[GpuIndex] = None, image_rows: Any = None,
                 collection_names: Optional[Sequence[str]] = None):
        """lexical_and: rank only chunks holding EVERY query term, as the reference's
        ``plainto_tsquery`` does (rag2_schema.sql:365); default is BM25's OR form, the
        north star's and the oracle's.
        image_index / image_rows: the legacy image channel (``kb_chunks_image_search``,
        20260113_add_kb_chunks.sql:236-268): a second dense index over the ``vector_image`` of the
        chunks that have one (``vector_image IS NOT NULL``), row j of it being store row
        ``image_rows[j]``.
        collection_names: fixes the collection name -> id mapping (sorted names of the WHOLE
        corpus): the shards of a document-sharded index must agree on it (sharded_client.py);
        default: the names this store's rows carry."""
        self.lexical_and = bool(lexical_and)
        self.image_index = image_index
        self.image_rows = None if image_rows is None else [int(r) for r in image_rows]
        if (image_index is None) != (image_rows is None):
            raise ValueError("image_index and image_rows come together")
        self.index = index
        self.store = store
        self.org_id = org_id
        self.token_embedder = token_embedder
        self._pin: Dict[Any, list] = {}   # reusable pinned staging buffers (one query per call)
        # collection names -> ids; the filter itself runs on the device, before the ranking
        self._coll_id: Dict[str, int] = {}
        if collection_names is not None:
            self._coll_id = {c: i for i, c in enumerate(sorted(set(collection_names)))}
        if store.collections is not None:
            if collection_names is None:
                names = sorted({c for c in store.collections if c is not None})
                self._coll_id = {c: i for i, c in enumerate(names)}
            if index.doc_coll is None and self.sets_collections:
                index.set_collections(np.array([self._coll_id.get(c, -2) if c is not None else -2
                                                for c in store.collections], dtype=np.int32))

    def _qcoll(self, collection):
        """int32 [1] collection id of a one-query call, None when unfiltered; a name no row
        carries gets an id no row carries (empty result, as the SQL WHERE would give)."""
        if collection is None or self.index.doc_coll is None:
            return None
        return torch.tensor([self._coll_id.get(collection, -3)], dtype=torch.int32, device=self.index.device)

    # ---------------------------------------------------------------- RPCs
    def rpc(self, name: str, params: Dict[str, Any]):
        if self.org_id is not None and params.get("p_org_id") not in (None, self.org_id):
            return _Reply([])  # data isolation: another tenant's index
        if name == "rag2_semantic_search":
            return _Reply(self._semantic(params["p_embedding"], int(params.get("p_limit", 100)),
                                         params.get("p_collection")))
        if name == "rag2_lexical_search":
            return _Reply(self._lexical(params["p_query"], int(params.get("p_limit", 50)),
                                        params.get("p_collection"), defer=bool(params.get("_defer"))))
        if name == "rag2_hybrid_rrf_search":
            return _Reply(self._hybrid_rrf(params))
        if name == "kb_chunks_vector_search":   # legacy RAG 1.0 (20260113_halfvec_4000.sql:70-105)
            rows = self._semantic(params["p_embedding"], int(params.get("p_limit", 50)), None)
            return _Reply([self._legacy_row(r, "similarity") for r in rows])
        if name == "kb_chunks_fts_pt":          # legacy RAG 1.0 (20260113_add_kb_chunks.sql:152-190)
            rows = self._lexical(params["p_query"], int(params.get("p_limit", 50)), None)
            return _Reply([self._legacy_row(r, "rank") for r in rows])
        if name == "kb_chunks_image_search":    # legacy RAG 1.0 (20260113_add_kb_chunks.sql:236-268)
            return _Reply(self._image(params["p_image_embedding"], int(params.get("p_limit", 10))))
        raise ValueError(f"unknown RPC {name!r}")

    def _image(self, embedding, limit: int) -> List[Dict[str, Any]]:
        """Cosine top-``limit`` over the image vectors (exact; the SQL orders by ``<=>``)."""
        if self.image_index is None:
            return []
        q = torch.tensor([list(map(float, embedding))], dtype=torch.float32,
                         device=self.image_index.device)
        if q.shape[1] != self.image_index.dim:
            raise ValueError(f"image embedding has {q.shape[1]} dims, index has {self.image_index.dim}")
        k = min(N.THR_DENSE_MAX_K, limit)
        S, I, cnt, _ = self.image_index.dense_search(q, k)
        out = []
        for j, sc in zip(I[0].tolist()[:int(cnt[0])], S[0].tolist()):
            row = self.store.result_row(self.image_rows[int(j) - self.image_index.doc_base])
            out.append({"id": row["child_id"], "content": row["text"], "modality": row["modality"],
                        "source_document": row["document_id"], "page": row["page"], "alt_text": None,
                        "image_path": None, "similarity": float(np.float32(sc))})   # ::REAL
        return out

    def _legacy_row(self, row: Dict[str, Any], score_key: str) -> Dict[str, Any]:
        i = self.store.row_index(row["child_id"])
        return {"id": row["child_id"], "content": row["text"], "modality": row["modality"],
                "source_document": row["document_id"], "page": row["page"], "chunk_index": i,
                "ocr_confidence": None, "is_table": row["modality"] == "table",
                "table_context": None, "alt_text": None, "category": None, "title": None,
                score_key: row[score_key]}

    def _hybrid_rrf(self, params: Dict[str, Any]) -> List[Dict[str, Any]]:
        """Server-side hybrid variant ``rag2_hybrid_rrf_search`` (rag2_schema.sql:413-496):
        lexical and semantic top ``p_limit*2`` each, FULL OUTER JOIN on the chunk id,
        ``w_l/(k+rank_l) + w_s/(k+rank_s)``, ORDER BY rrf_score DESC LIMIT p_limit -- one call,
        fused on the device by thr_rrf_fuse (ties, unspecified in SQL, follow the RRF kernel's
        sighting order)."""
        limit = int(params.get("p_limit", 50))
        coll = params.get("p_collection")
        wide = min(N.THR_RRF_MAX_PER_CHANNEL, 2 * limit)
        lex = self._lexical(params["p_query"], wide, coll)
        sem = self._semantic(params["p_embedding"], wide, coll)

        def ids(rows):
            t = torch.full((1, max(len(rows), 1)), -1, dtype=torch.int64, device=self.index.device)
            for j, r in enumerate(rows):
                t[0, j] = self.store.doc_base + self.store.row_index(r["child_id"])
            return t

        if not lex and not sem:
            return []
        out_ids, out_sc, out_rk, cnt = N.rrf_fuse(
            ids(lex), ids(sem), None, min(limit, 512), float(params.get("p_lexical_weight", 0.7)),
            float(params.get("p_semantic_weight", 0.8)), 1.0, int(params.get("p_rrf_k", 60)),
            want_ranks=True)
        rows = []
        for gid, sc, rk in zip(out_ids[0, : int(cnt[0])].tolist(), out_sc[0].tolist(), out_rk[0].tolist()):
            row = self.store.result_row(int(gid) - self.store.doc_base)
            row.update(rrf_score=float(np.float32(sc)), lexical_rank=rk[0] or None,
                       semantic_rank=rk[1] or None)
            rows.append(row)
        return rows

    # ---- one query per call: the host side is most of the latency, so every RPC does ONE pinned
    # upload and ONE read-back (scores and ids are the two halves of one [2, 1, k] tile) ----
    def _upload(self, values, dtype, device) -> torch.Tensor:
        """[1, n] device tensor of ``values`` through a reusable pinned buffer."""
        a = np.asarray(values, dtype=np.float32 if dtype == torch.float32 else np.int32).reshape(1, -1)
        key = (dtype, a.shape[1])
        slot = self._pin.get(key)
        if slot is None:
            slot = self._pin[key] = [torch.empty((1, a.shape[1]), dtype=dtype).pin_memory(), None]
        buf, ev = slot
        if ev is not None:
            ev.synchronize()            # the previous copy out of this buffer has left the host
        buf.numpy()[...] = a
        out = buf.to(device, non_blocking=True)
        if device.type == "cuda":
            slot[1] = torch.cuda.Event()
            slot[1].record()
        return out

    @staticmethod
    def _download(S: torch.Tensor, I: torch.Tensor):
        """(scores list, ids list without the -1 padding) in one device-to-host copy."""
        k = I.shape[1]
        if (S.dtype == torch.float64 and S.untyped_storage().data_ptr() == I.untyped_storage().data_ptr()
                and I.data_ptr() - S.data_ptr() == k * 8 and S.shape[0] == 1):
            both = torch.as_strided(S.view(torch.int64), (2, k), (k, 1)).cpu()
            scores, ids = both[0].view(torch.float64).tolist(), both[1].tolist()
        else:
            scores, ids = S[0].tolist(), I[0].tolist()
        n = 0
        while n < k and ids[n] >= 0:
            n += 1
        return scores[:n], ids[:n]

    def _rows(self, ids, scores, count, score_key, limit):
        """RPC result rows (the columns of rag2_schema.sql:350-358 / :386-394), best first."""
        st = self.store
        base, cid, pid, did = st.doc_base, st.child_ids, st.parent_ids, st.document_ids
        txt, pg, md = st.texts, st.pages, st.modalities
        out = []
        for gid, sc in zip(ids[:min(count, limit)], scores[:count]):
            i = gid - base
            out.append({"child_id": cid[i], "parent_id": pid[i], "document_id": did[i], "text": txt[i],
                        "page": pg[i], "modality": md[i], score_key: sc})
        return out

    def _semantic(self, embedding, limit: int, collection):
        if len(embedding) != self.index.dim:
            raise ValueError(f"embedding has {len(embedding)} dims, index has {self.index.dim}")
        q = self._upload(embedding, torch.float32, self.index.device)
        k = min(N.THR_DENSE_MAX_K, limit)
        S, I, _, _ = self.index.dense_search(q, k, collections=self._qcoll(collection), sync=False)
        scores, ids = self._download(S, I)
        return self._rows(ids, scores, len(ids), "similarity", limit)

    def _query_terms(self, query: str) -> Optional[List[int]]:
        """The distinct term ids of the query in order of first appearance (at most
        THR_BM25_MAX_TERMS), or None when nothing can match."""
        terms: List[int] = []
        unknown = False
        for tok in tokenize(query):
            t = self.store.vocab.get(tok)
            if t is None:
                unknown = True
            elif t not in terms:
                terms.append(t)
        if not terms or getattr(self.index, "lex", True) is None:
            return None
        # AND semantics (plainto_tsquery, rag2_schema.sql:365): a token no chunk holds, or a term
        # the kernel's term list would have to drop, makes the conjunction unsatisfiable -- the
        # SQL returns no rows, so does this (the OR form just ignores what it does not know)
        if self.lexical_and and (unknown or len(terms) > N.THR_BM25_MAX_TERMS):
            return None
        return terms[: N.THR_BM25_MAX_TERMS]

    def _lexical(self, query: str, limit: int, collection, defer: bool = False):
        """defer=True (``_defer`` in the RPC's params; RAG2Retriever sets it): the kernels are
        enqueued on the index's side stream and the rows are read back when first looked at --
        the retriever issues its semantic RPC in between, and the two channels overlap."""
        pending = getattr(self, "_pending_lex", None)
        if pending is not None:     # a deferred call still owns the index's lexical workspace:
            self._pending_lex = None
            try:
                pending.materialize()   # read it back before the kernels of this one are enqueued
            except Exception:           # (its own caller sees that failure when it looks at the rows)
                pass
        terms = self._query_terms(query)
        if terms is None:
            return []
        # (fixed width: one pinned buffer, one workspace size, whatever the number of terms)
        qt = self._upload(terms + [-1] * (N.THR_BM25_MAX_TERMS - len(terms)), torch.int32, self.index.device)
        k = min(N.THR_TOPK_MAX, limit)
        if not defer or self.index.device.type != "cuda":
            S, I, _ = self.index.bm25_search(qt, k, collections=self._qcoll(collection),
                                             conjunctive=self.lexical_and)
            scores, ids = self._download(S, I)
            return self._rows(ids, scores, len(ids), "rank", limit)
        side = self.index.side_stream()
        main = torch.cuda.current_stream(self.index.device)
        side.wait_stream(main)              # the upload was enqueued on the main stream
        qc = self._qcoll(collection)
        with torch.cuda.stream(side):
            S, I, _ = self.index.bm25_search(qt, k, collections=qc, conjunctive=self.lexical_and)
        for t in (qt, qc):
            if t is not None:
                t.record_stream(side)

        def fetch():
            with torch.cuda.stream(side):   # the copy is ordered after the kernels on their stream
                scores, ids = self._download(S, I)
            return self._rows(ids, scores, len(ids), "rank", limit)
        self._pending_lex = LazyRows(fetch)
        return self._pending_lex

    # -------------------------------------------------------------- tables
    def table(self, name: str) -> _TableQuery:
        if name == "rag_child_chunks":
            def fetch(ids):
                rows = []
                for cid in ids:
                    i = self.store.row_index(cid)
                    if i is not None:
                        rows.append(self.store.child_row(i))
                return rows
            return _TableQuery(fetch)
        if name == "rag_parent_chunks":
            return _TableQuery(lambda ids: [dict(self.store.parents[p]) for p in ids
                                            if p in self.store.parents])
        # tenant discovery of the tool layer (tools/crm_knowledge.py:89-101 in the reference)
        if name == "rag_documents":
            return _TableQuery(lambda _ids: [{"org_id": self.org_id}] if self.org_id else [])
        if name == "organizations":
            return _TableQuery(lambda _ids: [{"id": self.org_id}] if self.org_id else [])
        raise ValueError(f"table {name!r} is not served by the GPU index")

    # --------------------------------------------------------------- graph
    def _entity_index(self):
        """Trigram index of the lower-cased entity names, built on first use: (sorted trigram
        codes, entity of each) -- a keyword of >= 3 characters is looked up by intersecting the
        entity lists of its trigrams instead of scanning every name (2.5M names at 10M docs)."""
        if getattr(self, "_tri", None) is None:
            names = [nm.lower() for nm in self.store.entity_names]
            cp = np.frombuffer("\x00".join(names).encode("utf-32-le"), dtype=np.uint32).astype(np.int64)
            ent = np.repeat(np.arange(len(names), dtype=np.int64),
                            np.fromiter((len(nm) + 1 for nm in names), dtype=np.int64, count=len(names)))[:len(cp)]
            ok = np.ones(max(len(cp) - 2, 0), dtype=bool)
            for off in range(3):   # a trigram must not straddle the separator between two names
                ok &= cp[off:len(cp) - 2 + off] != 0
            code = (cp[:-2] * 1114112 + cp[1:-1]) * 1114112 + cp[2:] if len(cp) > 2 else cp[:0]
            code, e3 = code[ok], ent[:len(ok)][ok]
            order = np.argsort(code, kind="stable")      # (entities stay ascending inside a trigram)
            self._tri = (code[order], e3[order], names)
        return self._tri

    def find_entities(self, keywords: List[str], limit: int = 20) -> List[int]:
        """Entity ids whose name contains a keyword (ILIKE '%kw%'), at most 5 keywords and
        limit // len(keywords) entities each, in ascending entity order (graph_search.py:161-170)."""
        if not keywords or not self.store.entity_names:
            return []
        codes, ents, names = self._entity_index()
        per = max(1, limit // len(keywords))
        found: List[int] = []
        for kw in keywords[:5]:
            needle, hits = kw.lower(), 0
            if len(needle) >= 3:
                cp = np.frombuffer(needle.encode("utf-32-le"), dtype=np.uint32).astype(np.int64)
                tri = np.unique((cp[:-2] * 1114112 + cp[1:-1]) * 1114112 + cp[2:])
                lo, hi = np.searchsorted(codes, tri, "left"), np.searchsorted(codes, tri, "right")
                cand = None
                for j in np.argsort(hi - lo):            # rarest trigram first
                    part = np.unique(ents[lo[j]:hi[j]])
                    cand = part if cand is None else cand[np.isin(cand, part, assume_unique=True)]
                    if len(cand) == 0:
                        break
                pool = cand.tolist() if cand is not None else []
            else:
                pool = range(len(names))                 # too short for a trigram: the plain scan
            for e in pool:
                if needle in names[e]:
                    if e not in found:
                        found.append(e)
                    hits += 1
                    if hits == per:
                        break
        return found[: N.THR_GRAPH_MAX_SEEDS]

    def entity_name(self, e: int) -> str:
        return self.store.entity_names[e]

    def graph_chunks(self, seeds: List[int], top_k: int, hops: int = 2) -> List[str]:
        qs = self._upload(seeds + [-1] * (N.THR_GRAPH_MAX_SEEDS - len(seeds)), torch.int32, self.index.device)
        S, I, _ = self.index.graph_search(qs, min(N.THR_TOPK_MAX, top_k), hops)
        _, ids = self._download(S, I)
        return [self.store.child_ids[int(g) - self.store.doc_base] for g in ids]

    # -------------------------------------------------------------- rerank
    def maxsim_scores(self, query: str, child_ids: List[str]) -> List[float]:
        """MaxSim of the query's token matrix against the given chunks, divided by the number of
        query tokens so unit-norm tokens give a [-1, 1] relevance like a cross-encoder's."""
        if self.token_embedder is None or self.index.tokens is None:
            raise RuntimeError("no token embedder / token store: late-interaction rerank unavailable")
        qtok = np.asarray(self.token_embedder.embed_query_tokens(query), dtype=np.float16)[None]
        ids = [self.store.row_index(c) for c in child_ids]
        cand = torch.tensor([[self.store.doc_base + i if i is not None else -1 for i in ids]],
                            dtype=torch.int64, device=self.index.device)
        sc = self.index.maxsim(torch.from_numpy(qtok).to(self.index.device), cand)[0]
        return [float(v) / qtok.shape[1] if np.isfinite(v) else 0.5 for v in sc.tolist()]

    def text_to_child_id(self, text: str) -> Optional[str]:
        if not hasattr(self, "_by_text"):
            self._by_text = {t: c for t, c in zip(self.store.texts, self.store.child_ids)}
            for pid, row in self.store.parents.items():
                self._by_text.setdefault(row["text"], None)
        return self._by_text.get(text)


_default_client: Optional[GpuIndexClient] = None


def set_default_client(client: Optional[GpuIndexClient]) -> None:
    global _default_client
    _default_client = client


def get_supabase_client() -> GpuIndexClient:
    """What ``RAG2Retriever.supabase`` resolves to (reference: utils/db.py:372-400)."""
    if _default_client is None:
        raise RuntimeError("no GPU index registered: call backend.set_default_client(...)")
    return _default_client
