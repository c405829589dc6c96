"""Index build / ingest boundary (the step BEFORE the path).

Turns the rows the reference's ingestion writes -- ``rag_child_chunks(id, parent_id,
document_id, text, page, modality, embedding_1024)``, ``rag_parent_chunks(id, text,
section_heading)``, ``rag_entities(id, name)``, ``rag_relations(subject_entity_id,
object_entity_id, confidence)``, ``rag_entity_mentions(entity_id, child_chunk_id[,
confidence])`` (database/migrations/20260114_rag2_schema.sql:104-283;
src/voice_agent/rag2/ingest.py:361-470) -- into the GPU-resident arrays / CSRs and the host
row store, and saves / loads them as a directory of ``.npy`` files + one JSON.

Two routes to the lexical index.  ``build_lexical`` is host numpy (what runs without a GPU:
small corpora, the CPU tests).  ``from_rows(..., device=True)`` / ``lexical_rows`` +
``GpuIndex.set_lexical_rows`` tokenise at C speed on the host (one regex pass per chunk, one
hash factorisation of all tokens) and build the CSR ON THE DEVICE (thr_lexical_build: radix sort
of the (term, doc) pairs, segmented reduce, row pointers by binary search) -- a 1M-chunk shard's
32M postings in well under a second; document shards all-reduce their df / length sums for the
global idf / avgdl.  ``save`` writes, next to the source arrays, what index set-up computed on
the device (``GpuIndex.export_derived``: the float16 image of the rows, the BM25 bounds /
impacts / dense-term rows) and ``load(...).to_gpu()`` reuses them; the row payloads live in
blob + offset files (``StringColumn``), not in ``meta.json``.
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
    derived: Optional[Dict[str, Any]] = None   # GpuIndex.export_derived(): saved with the index, reused by to_gpu()

    def to_gpu(self, doc_base: int = 0):
        from .index import GpuIndex
        idx = GpuIndex(doc_base=doc_base).set_dense(self.docs, derived=self.derived)
        if self.rowptr is not None:
            idx.set_lexical(self.rowptr, self.post_doc, self.post_tf, self.doclen, self.idf,
                            self.avgdl, K1, B, derived=self.derived)
        if self.ent_rowptr is not None:
            idx.set_graph(self.ent_rowptr, self.ent_col, self.men_rowptr, self.men_chunk,
                          self.men_conf)
        if self.tokens is not None:
            idx.set_tokens(self.tokens)
        return idx


class StringColumn(Sequence):
    """A column of strings as ONE utf-8 blob + int64 offsets (memory-mapped when loaded): what a
    10M-row store keeps instead of ten million Python objects in a JSON file.  Decodes on access."""

    def __init__(self, blob, offsets):
        self.blob, self.offsets = blob, offsets

    @classmethod
    def from_strings(cls, values: Sequence[Optional[str]]) -> "StringColumn":
        enc = [b"\x00" if v is None else str(v).encode("utf-8") for v in values]   # (NUL alone = None)
        off = np.zeros(len(enc) + 1, dtype=np.int64)
        np.cumsum(np.fromiter(map(len, enc), dtype=np.int64, count=len(enc)), out=off[1:])
        return cls(np.frombuffer(b"".join(enc), dtype=np.uint8), off)

    def __len__(self):
        return len(self.offsets) - 1

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        raw = bytes(self.blob[int(self.offsets[i]):int(self.offsets[i + 1])])
        return None if raw == b"\x00" else raw.decode("utf-8")

    def __eq__(self, other):
        return len(self) == len(other) and all(a == b for a, b in zip(self, other))

    __hash__ = None


def lexical_rows(texts: Sequence[str], tokenizer: Callable[[str], List[str]] = tokenize,
                 vocab: Optional[Dict[str, int]] = None):
    """Texts -> (vocab, doc int32 [n_tokens], term int32 [n_tokens]): one entry per token
    occurrence, term -1 for a token outside a given ``vocab`` (it still counts toward its chunk's
    length).  One tokenizer call per chunk, then array work: a new vocabulary numbers the tokens in
    order of first appearance (pandas' hash factorisation), exactly as ``build_lexical`` does."""
    import itertools
    per_doc = [tokenizer(t) for t in texts]
    lens = np.fromiter(map(len, per_doc), dtype=np.int64, count=len(per_doc))
    flat = list(itertools.chain.from_iterable(per_doc))
    doc = np.repeat(np.arange(len(per_doc), dtype=np.int32), lens)
    if vocab is None:
        import pandas as pd
        codes, uniques = pd.factorize(np.asarray(flat, dtype=object)) if flat else (np.zeros(0, np.int64), [])
        return {tok: i for i, tok in enumerate(uniques)}, doc, codes.astype(np.int32)
    term = np.fromiter((vocab.get(t, -1) for t in flat), dtype=np.int32, count=len(flat))
    return vocab, doc, term


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
              tokenizer: Callable[[str], List[str]] = tokenize, doc_base: int = 0,
              device: bool = False) -> HostIndex:
    """Reference table rows -> HostIndex (+ CorpusStore).  Rows without an embedding keep a
    zero vector, i.e. are excluded from the dense channel (``embedding_1024 IS NOT NULL``).
    device=True builds the inverted index on the GPU (``lexical_rows`` + thr_lexical_build) and
    brings the arrays back for ``save``; the result equals the host route's."""
    n = len(child_rows)
    dim = next((len(r[embedding_key]) for r in child_rows if r.get(embedding_key) is not None), 0)
    docs = np.zeros((n, dim), dtype=np.float32)
    for i, r in enumerate(child_rows):
        if r.get(embedding_key) is not None:
            docs[i] = np.asarray(r[embedding_key], dtype=np.float32)
    texts = [r.get("text", "") for r in child_rows]
    if device:
        from . import _native as N
        import torch
        vocab, d_tok, t_tok = lexical_rows(texts, tokenizer)
        dev = torch.device("cuda", torch.cuda.current_device())
        rowptr, pd, ptf, dl, df = (a.cpu().numpy() for a in N.lexical_build(
            torch.from_numpy(d_tok).to(dev), torch.from_numpy(t_tok).to(dev), None, n, max(len(vocab), 1)))
        dfh = df.astype(np.float64)
        idf = np.log(1.0 + (float(n) - dfh + 0.5) / (dfh + 0.5))
        avgdl = float(dl.astype(np.float64).sum()) / max(n, 1) or 1.0
    else:
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
_DERIVED_ARRAYS = ("docs16", "term_ub", "block_ub", "post_imp", "dense_slot", "dense_imp", "dense_tf")
_DERIVED_SCALARS = ("doc_rel_err", "f16_layout", "lexical_tag", "dense_stride")
_STRING_COLUMNS = ("child_ids", "parent_ids", "document_ids", "texts", "modalities", "collections",
                   "entity_names")


def save(hi: HostIndex, path: str, gpu_index=None) -> None:
    """``gpu_index``: the GpuIndex built from ``hi`` -- what its set-up computed on the device
    (export_derived) is saved too, so that a later ``load(path).to_gpu()`` does not recompute it."""
    os.makedirs(path, exist_ok=True)
    for name in _ARRAYS:
        arr = getattr(hi, name)
        if arr is not None:
            np.save(os.path.join(path, name + ".npy"), arr, allow_pickle=False)
    meta: Dict[str, Any] = {"avgdl": hi.avgdl, "format": 2}
    derived = gpu_index.export_derived() if gpu_index is not None else hi.derived
    if derived:
        for name in _DERIVED_ARRAYS:
            if derived.get(name) is not None:
                np.save(os.path.join(path, "derived_" + name + ".npy"), np.asarray(derived[name]), allow_pickle=False)
        meta["derived"] = {k: derived[k] for k in _DERIVED_SCALARS if derived.get(k) is not None}
    if hi.store is not None:
        s = hi.store
        columns = []
        for name in _STRING_COLUMNS:   # the big columns: blob + offsets, not JSON
            values = getattr(s, name)
            if values is None:
                continue
            col = values if isinstance(values, StringColumn) else StringColumn.from_strings(values)
            np.save(os.path.join(path, f"store_{name}_blob.npy"), np.asarray(col.blob), allow_pickle=False)
            np.save(os.path.join(path, f"store_{name}_off.npy"), np.asarray(col.offsets), allow_pickle=False)
            columns.append(name)
        np.save(os.path.join(path, "store_pages.npy"), np.asarray(s.pages, dtype=np.int32), allow_pickle=False)
        voc = list(s.vocab.items())
        vcol = StringColumn.from_strings([k for k, _ in voc])
        np.save(os.path.join(path, "store_vocab_blob.npy"), np.asarray(vcol.blob), allow_pickle=False)
        np.save(os.path.join(path, "store_vocab_off.npy"), np.asarray(vcol.offsets), allow_pickle=False)
        np.save(os.path.join(path, "store_vocab_ids.npy"), np.asarray([v for _, v in voc], dtype=np.int64),
                allow_pickle=False)
        meta["store"] = {"columns": columns, "parents": s.parents, "doc_base": s.doc_base}
    with open(os.path.join(path, "meta.json"), "w") as f:
        json.dump(meta, f)


def load(path: str, mmap: bool = True) -> HostIndex:
    with open(os.path.join(path, "meta.json")) as f:
        meta = json.load(f)
    mode = "r" if mmap else None

    def arr(name):
        fp = os.path.join(path, name + ".npy")
        return np.load(fp, mmap_mode=mode, allow_pickle=False) if os.path.exists(fp) else None
    arrays = {name: arr(name) for name in _ARRAYS}
    derived = None
    if "derived" in meta:
        derived = dict(meta["derived"])
        derived.update({name: arr("derived_" + name) for name in _DERIVED_ARRAYS})
    store = None
    if "store" in meta:
        ms = meta["store"]
        if "columns" not in ms:    # format 1: everything in the JSON
            store = CorpusStore(**ms)
        else:
            cols = {name: StringColumn(arr(f"store_{name}_blob"), arr(f"store_{name}_off")) for name in ms["columns"]}
            vcol = StringColumn(arr("store_vocab_blob"), arr("store_vocab_off"))
            vocab = dict(zip(vcol, (int(v) for v in arr("store_vocab_ids"))))
            store = CorpusStore(child_ids=cols["child_ids"], parent_ids=cols["parent_ids"],
                                document_ids=cols["document_ids"], texts=cols["texts"],
                                pages=arr("store_pages").tolist(), modalities=cols["modalities"], parents=ms["parents"],
                                collections=cols.get("collections"), vocab=vocab,
                                entity_names=cols.get("entity_names", []), doc_base=ms["doc_base"])
    return HostIndex(avgdl=meta["avgdl"], store=store, derived=derived, **arrays)
