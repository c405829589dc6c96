"""Index build / ingest boundary (the step BEFORE the path).

Turns the rows the reference's ingestion writes -- ``rag_child_chunks(id, parent_id,
document_id, text, page, modality, embedding_1024)``, ``rag_parent_chunks(id, text,
section_heading)``, ``rag_entities(id, name)``, ``rag_relations(subject_entity_id,
object_entity_id, confidence)``, ``rag_entity_mentions(entity_id, child_chunk_id[,
confidence])`` (database/migrations/20260114_rag2_schema.sql:104-283;
src/voice_agent/rag2/ingest.py:361-470) -- into the GPU-resident arrays / CSRs and the host
row store, and saves / loads them as a directory of ``.npy`` files + one JSON.

Host-side numpy: this is offline index construction, not the query path.
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass
from typing import Any, Callable, Dict, Iterable, List, Optional, Sequence

import numpy as np

from .backend import CorpusStore, tokenize

K1, B = 1.2, 0.75


@dataclass
class HostIndex:
    """Everything an index holds, on the host (numpy), ready for ``to_gpu``."""
    docs: np.ndarray                      # f32 [n, D]
    rowptr: Optional[np.ndarray] = None   # lexical CSR
    post_doc: Optional[np.ndarray] = None
    post_tf: Optional[np.ndarray] = None
    doclen: Optional[np.ndarray] = None
    idf: Optional[np.ndarray] = None
    avgdl: float = 0.0
    ent_rowptr: Optional[np.ndarray] = None
    ent_col: Optional[np.ndarray] = None
    men_rowptr: Optional[np.ndarray] = None
    men_chunk: Optional[np.ndarray] = None
    men_conf: Optional[np.ndarray] = None
    tokens: Optional[np.ndarray] = None   # f16 [n, T, 128]
    store: Optional[CorpusStore] = None

    def to_gpu(self, doc_base: int = 0):
        from .index import GpuIndex
        idx = GpuIndex(doc_base=doc_base).set_dense(self.docs)
        if self.rowptr is not None:
            idx.set_lexical(self.rowptr, self.post_doc, self.post_tf, self.doclen, self.idf,
                            self.avgdl, K1, B)
        if self.ent_rowptr is not None:
            idx.set_graph(self.ent_rowptr, self.ent_col, self.men_rowptr, self.men_chunk,
                          self.men_conf)
        if self.tokens is not None:
            idx.set_tokens(self.tokens)
        return idx


def build_lexical(texts: Sequence[str], tokenizer: Callable[[str], List[str]] = tokenize,
                  vocab: Optional[Dict[str, int]] = None, n_docs_global: Optional[int] = None,
                  df_global: Optional[np.ndarray] = None, sum_dl_global: Optional[float] = None):
    """Texts -> (vocab, rowptr, post_doc, post_tf, doclen, idf, avgdl).  For a document shard
    pass the GLOBAL ``vocab`` / ``n_docs_global`` / ``df_global`` / ``sum_dl_global`` so that
    idf and avgdl are the whole corpus's."""
    grow = vocab is None
    vocab = {} if vocab is None else vocab
    d_idx, t_idx, tfs = [], [], []
    doclen = np.zeros(len(texts), dtype=np.float32)
    for i, text in enumerate(texts):
        counts: Dict[int, int] = {}
        toks = tokenizer(text)
        doclen[i] = len(toks)
        for tok in toks:
            t = vocab.get(tok)
            if t is None:
                if not grow:
                    continue
                t = vocab[tok] = len(vocab)
            counts[t] = counts.get(t, 0) + 1
        for t, c in counts.items():
            d_idx.append(i)
            t_idx.append(t)
            tfs.append(c)
    v = len(vocab)
    d_idx = np.asarray(d_idx, dtype=np.int32)
    t_idx = np.asarray(t_idx, dtype=np.int32)
    tfs = np.asarray(tfs, dtype=np.int32)
    order = np.lexsort((d_idx, t_idx))
    df_local = np.bincount(t_idx, minlength=v).astype(np.int64)
    rowptr = np.concatenate([[0], np.cumsum(df_local)]).astype(np.int64)
    n_glob = n_docs_global if n_docs_global is not None else len(texts)
    df = (df_global if df_global is not None else df_local).astype(np.float64)
    idf = np.log(1.0 + (float(n_glob) - df + 0.5) / (df + 0.5))
    total_dl = sum_dl_global if sum_dl_global is not None else float(doclen.astype(np.float64).sum())
    avgdl = total_dl / max(n_glob, 1)
    return vocab, rowptr, d_idx[order], tfs[order], doclen, idf, (avgdl if avgdl > 0 else 1.0)


def build_graph(entity_ids: Sequence[Any], relations: Iterable[Dict[str, Any]],
                mentions: Iterable[Dict[str, Any]], chunk_index: Dict[Any, int]):
    """Entity / relation / mention rows -> (ent_rowptr, ent_col, men_rowptr, men_chunk,
    men_conf).  Relations are stored in both directions (the reference walks them
    undirected, graph_search.py:196-226); mention confidence defaults to 1.0;
    ``chunk_index`` maps child-chunk id -> GLOBAL doc index."""
    eidx = {e: i for i, e in enumerate(entity_ids)}
    n = len(eidx)
    src, dst = [], []
    for r in relations:
        a, b = eidx.get(r["subject_entity_id"]), eidx.get(r["object_entity_id"])
        if a is None or b is None or a == b:
            continue
        src += [a, b]
        dst += [b, a]
    src, dst = np.asarray(src, dtype=np.int64), np.asarray(dst, dtype=np.int64)
    if len(src):
        pairs = np.unique(np.stack([src, dst], axis=1), axis=0)  # sorted by (src, dst), deduplicated
        src, dst = pairs[:, 0], pairs[:, 1]
    ent_rowptr = np.concatenate([[0], np.cumsum(np.bincount(src, minlength=n))]).astype(np.int64)
    me, mc, mw = [], [], []
    for m in mentions:
        e, c = eidx.get(m["entity_id"]), chunk_index.get(m["child_chunk_id"])
        if e is None or c is None:
            continue
        me.append(e)
        mc.append(c)
        mw.append(float(m.get("confidence", 1.0)))
    me = np.asarray(me, dtype=np.int64)
    order = np.lexsort((np.asarray(mc, dtype=np.int64), me)) if len(me) else np.zeros(0, dtype=np.int64)
    men_rowptr = np.concatenate([[0], np.cumsum(np.bincount(me, minlength=n))]).astype(np.int64)
    return (ent_rowptr, dst.astype(np.int32), men_rowptr,
            np.asarray(mc, dtype=np.int32)[order], np.asarray(mw, dtype=np.float32)[order])


def from_rows(child_rows: Sequence[Dict[str, Any]], parent_rows: Sequence[Dict[str, Any]] = (),
              entity_rows: Sequence[Dict[str, Any]] = (), relation_rows: Sequence[Dict[str, Any]] = (),
              mention_rows: Sequence[Dict[str, Any]] = (), embedding_key: str = "embedding_1024",
              tokenizer: Callable[[str], List[str]] = tokenize, doc_base: int = 0) -> HostIndex:
    """Reference table rows -> HostIndex (+ CorpusStore).  Rows without an embedding keep a
    zero vector, i.e. are excluded from the dense channel (``embedding_1024 IS NOT NULL``)."""
    n = len(child_rows)
    dim = next((len(r[embedding_key]) for r in child_rows if r.get(embedding_key) is not None), 0)
    docs = np.zeros((n, dim), dtype=np.float32)
    for i, r in enumerate(child_rows):
        if r.get(embedding_key) is not None:
            docs[i] = np.asarray(r[embedding_key], dtype=np.float32)
    texts = [r.get("text", "") for r in child_rows]
    vocab, rowptr, pd, ptf, dl, idf, avgdl = build_lexical(texts, tokenizer)
    store = CorpusStore(
        child_ids=[r["id"] for r in child_rows], parent_ids=[r.get("parent_id") for r in child_rows],
        document_ids=[r.get("document_id") for r in child_rows], texts=texts,
        pages=[r.get("page", 1) for r in child_rows],
        modalities=[r.get("modality", "text") for r in child_rows],
        parents={p["id"]: {"id": p["id"], "text": p.get("text", ""),
                           "section_heading": p.get("section_heading")} for p in parent_rows},
        collections=[r.get("collection") for r in child_rows]
        if any("collection" in r for r in child_rows) else None,
        vocab=vocab, entity_names=[e.get("name", "") for e in entity_rows], doc_base=doc_base)
    hi = HostIndex(docs=docs, rowptr=rowptr, post_doc=pd, post_tf=ptf, doclen=dl, idf=idf,
                   avgdl=avgdl, store=store)
    if entity_rows:
        cidx = {cid: doc_base + i for i, cid in enumerate(store.child_ids)}
        (hi.ent_rowptr, hi.ent_col, hi.men_rowptr, hi.men_chunk, hi.men_conf) = build_graph(
            [e["id"] for e in entity_rows], relation_rows, mention_rows, cidx)
    return hi


_ARRAYS = ("docs", "rowptr", "post_doc", "post_tf", "doclen", "idf", "ent_rowptr", "ent_col",
           "men_rowptr", "men_chunk", "men_conf", "tokens")


def save(hi: HostIndex, path: str) -> None:
    os.makedirs(path, exist_ok=True)
    for name in _ARRAYS:
        arr = getattr(hi, name)
        if arr is not None:
            np.save(os.path.join(path, name + ".npy"), arr, allow_pickle=False)
    meta: Dict[str, Any] = {"avgdl": hi.avgdl, "format": 1}
    if hi.store is not None:
        s = hi.store
        meta["store"] = {k: getattr(s, k) for k in ("child_ids", "parent_ids", "document_ids", "texts",
                                                     "pages", "modalities", "parents", "collections",
                                                     "vocab", "entity_names", "doc_base")}
    with open(os.path.join(path, "meta.json"), "w") as f:
        json.dump(meta, f)


def load(path: str, mmap: bool = True) -> HostIndex:
    with open(os.path.join(path, "meta.json")) as f:
        meta = json.load(f)
    arrays = {}
    for name in _ARRAYS:
        fp = os.path.join(path, name + ".npy")
        arrays[name] = np.load(fp, mmap_mode="r" if mmap else None, allow_pickle=False) \
            if os.path.exists(fp) else None
    store = CorpusStore(**meta["store"]) if "store" in meta else None
    return HostIndex(avgdl=meta["avgdl"], store=store, **arrays)
