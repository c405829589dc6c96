"""Deterministic synthetic corpora (SURVEY.md section 8d / BASELINE.md section 2).

Everything is generated in fixed blocks of 65 536 documents from block-seeded
PCG64 streams, so a shard of any size holds exactly the rows the unsharded
corpus holds at the same global indices, and the CPU oracle and the GPU path
are fed identical bytes.  Host-side numpy only (this is input generation, not
the product path).
"""
from __future__ import annotations

import os
from concurrent.futures import ThreadPoolExecutor
from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np

BLOCK = 65536
SEED_DOCS = 1234
SEED_QUERIES = 4321
TERMS_PER_DOC = 32        # distinct terms per doc (L-bar)
POSTINGS_PER_TERM = 64    # average df  => V = N * 32 / 64
ZIPF_S = 1.07
ENT_PER_DOC = 0.25        # E = N / 4
ENT_DEGREE = 8
MENTIONS_PER_ENT = 4


def _rng(*seed) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64(list(seed)))


def _normalize_rows_f32(x: np.ndarray) -> np.ndarray:
    n = np.sqrt(np.einsum("ij,ij->i", x, x, dtype=np.float32)).astype(np.float32)
    n[n == 0] = 1
    x /= n[:, None]
    return x


# ------------------------------------------------------------------ dense
def dense_block(b: int, dim: int) -> np.ndarray:
    x = _rng(SEED_DOCS, b).standard_normal((BLOCK, dim), dtype=np.float32)
    return _normalize_rows_f32(x)


def dense_rows(start: int, count: int, dim: int) -> np.ndarray:
    """Rows [start, start+count) of the global corpus."""
    out = np.empty((count, dim), dtype=np.float32)
    b0, b1 = start // BLOCK, (start + count - 1) // BLOCK

    def fill(b):
        blk = dense_block(b, dim)
        lo = max(start, b * BLOCK)
        hi = min(start + count, (b + 1) * BLOCK)
        out[lo - start:hi - start] = blk[lo - b * BLOCK:hi - b * BLOCK]

    # numpy's Generator releases the GIL while filling: blocks generate in parallel
    with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1, b1 - b0 + 1)) as ex:
        list(ex.map(fill, range(b0, b1 + 1)))
    return out


def dense_queries(n_queries: int, dim: int, n_docs: int, noise: float = 0.5) -> np.ndarray:
    """Half planted neighbours normalize(doc[j] + noise*g) with g a unit-norm noise vector
    (cosine to the planted row ~ 1/sqrt(1+noise^2) = 0.894), half pure noise."""
    r = _rng(SEED_QUERIES)
    g = _normalize_rows_f32(r.standard_normal((n_queries, dim), dtype=np.float32))
    planted = np.arange(n_queries) % 2 == 0
    j = r.integers(0, n_docs, size=n_queries)
    q = g.copy()
    blocks = j // BLOCK
    for b in np.unique(blocks[planted]):  # each needed block is generated once
        blk = dense_block(int(b), dim)
        sel = np.nonzero(planted & (blocks == b))[0]
        q[sel] = blk[j[sel] - int(b) * BLOCK] + np.float32(noise) * g[sel]
    return _normalize_rows_f32(q)


# ---------------------------------------------------------------- lexical
def vocab_size(n_docs_global: int) -> int:
    return max(64, n_docs_global * TERMS_PER_DOC // POSTINGS_PER_TERM)


def _zipf_cdf(v: int) -> np.ndarray:
    w = 1.0 / np.power(np.arange(1, v + 1, dtype=np.float64), ZIPF_S)
    c = np.cumsum(w)
    return c / c[-1]


def _lexical_block(b: int, v: int, cdf: np.ndarray):
    """Block b of the global corpus: (term [BLOCK, draws] sorted per row, tf, keep mask)."""
    draws = TERMS_PER_DOC + 8
    r = _rng(SEED_DOCS, 1, b)
    t = np.searchsorted(cdf, r.random((BLOCK, draws))).astype(np.int32)
    t = np.minimum(t, v - 1)
    tf = r.geometric(0.5, size=(BLOCK, draws)).astype(np.int32)  # = 1 + Geometric0(0.5)
    t.sort(axis=1)
    keep = np.ones_like(t, dtype=bool)
    keep[:, 1:] = t[:, 1:] != t[:, :-1]
    # at most 32 distinct terms per doc
    keep &= np.cumsum(keep, axis=1) <= TERMS_PER_DOC
    return t, tf, keep


def _workers(n_tasks: int) -> int:
    return max(1, min(16, os.cpu_count() or 1, n_tasks))


def lexical_rows(start: int, count: int, n_docs_global: int
                 ) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Per-doc bags for rows [start, start+count): (doc_local i32[nnz], term i32[nnz],
    tf i32[nnz]) -- up to 32 distinct Zipf-distributed terms per doc,
    tf ~ 1 + Geometric(0.5)."""
    v = vocab_size(n_docs_global)
    cdf = _zipf_cdf(v)
    b0, b1 = start // BLOCK, (start + count - 1) // BLOCK
    draws = TERMS_PER_DOC + 8

    def block(b):
        t, tf, keep = _lexical_block(b, v, cdf)
        lo = max(start, b * BLOCK) - b * BLOCK
        hi = min(start + count, (b + 1) * BLOCK) - b * BLOCK
        rows = np.repeat(np.arange(lo, hi, dtype=np.int64)[:, None], draws, axis=1)
        sel = keep[lo:hi]
        return ((rows[sel] + b * BLOCK - start).astype(np.int32), t[lo:hi][sel], tf[lo:hi][sel])

    # (numpy releases the GIL in the generators, searchsorted and sort: blocks run in parallel)
    with ThreadPoolExecutor(max_workers=_workers(b1 - b0 + 1)) as ex:
        parts = list(ex.map(block, range(b0, b1 + 1)))
    return (np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts]),
            np.concatenate([p[2] for p in parts]))


def lexical_global_stats(n_docs_global: int) -> Tuple[np.ndarray, float]:
    """(df int64 [V], sum of doc lengths) of the WHOLE corpus without keeping its rows: what a
    shard needs for the global idf / avgdl when no other rank is there to all-reduce with."""
    v = vocab_size(n_docs_global)
    cdf = _zipf_cdf(v)
    nb = (n_docs_global + BLOCK - 1) // BLOCK

    def block(b):
        t, tf, keep = _lexical_block(b, v, cdf)
        hi = min(n_docs_global, (b + 1) * BLOCK) - b * BLOCK
        sel = keep[:hi]
        return t[:hi][sel], int(tf[:hi][sel].sum(dtype=np.int64))

    df = np.zeros(v, dtype=np.int64)
    sum_dl = 0
    with ThreadPoolExecutor(max_workers=_workers(nb)) as ex:
        for terms, s in ex.map(block, range(nb)):
            df += np.bincount(terms, minlength=v)
            sum_dl += s
    return df, float(sum_dl)


@dataclass
class LexicalCSR:
    rowptr: np.ndarray    # int64 [V+1]
    post_doc: np.ndarray  # int32 [nnz] local doc index, ascending inside a term
    post_tf: np.ndarray   # int32 [nnz]
    doclen: np.ndarray    # float32 [n_local]
    df_local: np.ndarray  # int64 [V]
    sum_dl_local: float


def build_lexical_csr(doc: np.ndarray, term: np.ndarray, tf: np.ndarray, n_local: int,
                      v: int) -> LexicalCSR:
    order = np.lexsort((doc, term))
    term_s, doc_s, tf_s = term[order], doc[order], tf[order]
    df = np.bincount(term_s, minlength=v).astype(np.int64)
    rowptr = np.concatenate([[0], np.cumsum(df)]).astype(np.int64)
    doclen = np.bincount(doc, weights=tf, minlength=n_local).astype(np.float32)
    return LexicalCSR(rowptr, doc_s.astype(np.int32), tf_s.astype(np.int32), doclen, df,
                      float(doclen.astype(np.float64).sum()))


def lexical_queries(n_queries: int, df_global: np.ndarray, n_terms: int = 4) -> np.ndarray:
    """T term ids per query sampled proportionally to df (without replacement)."""
    r = _rng(SEED_QUERIES, 1)
    p = df_global.astype(np.float64)
    cdf = np.cumsum(p) / p.sum()
    out = np.full((n_queries, n_terms), -1, dtype=np.int32)
    t = np.searchsorted(cdf, r.random((n_queries, n_terms * 2))).astype(np.int32)
    t = np.minimum(t, len(df_global) - 1)
    for i in range(n_queries):
        seen = []
        for x in t[i]:
            if x not in seen and df_global[x] > 0:
                seen.append(int(x))
            if len(seen) == n_terms:
                break
        out[i, :len(seen)] = seen
    return out


# ------------------------------------------------------------------ graph
@dataclass
class GraphCSR:
    ent_rowptr: np.ndarray  # int64 [E+1]
    ent_col: np.ndarray     # int32, both directions stored
    men_rowptr: np.ndarray  # int64 [E+1]
    men_chunk: np.ndarray   # int32 GLOBAL chunk ids
    men_conf: np.ndarray    # float32


def n_entities(n_docs_global: int) -> int:
    return max(8, int(n_docs_global * ENT_PER_DOC))


def build_graph(n_docs_global: int, chunk_lo: int = 0, chunk_hi: Optional[int] = None) -> GraphCSR:
    """Entity graph (replicated) + mentions restricted to chunks [chunk_lo, chunk_hi)."""
    e = n_entities(n_docs_global)
    chunk_hi = n_docs_global if chunk_hi is None else chunk_hi
    r = _rng(SEED_DOCS, 2)
    m = e * ENT_DEGREE // 2  # undirected edges; stored both ways => avg degree 8
    src = r.integers(0, e, size=m).astype(np.int32)
    dst = r.integers(0, e, size=m).astype(np.int32)
    ok = src != dst
    a = np.concatenate([src[ok], dst[ok]])
    bb = np.concatenate([dst[ok], src[ok]])
    order = np.lexsort((bb, a))
    a, bb = a[order], bb[order]
    ent_rowptr = np.concatenate([[0], np.cumsum(np.bincount(a, minlength=e))]).astype(np.int64)
    nm = e * MENTIONS_PER_ENT
    me = r.integers(0, e, size=nm).astype(np.int32)
    mc = r.integers(0, n_docs_global, size=nm).astype(np.int64)
    conf = (0.5 + 0.5 * r.random(nm)).astype(np.float32)
    order = np.lexsort((mc, me))
    me, mc, conf = me[order], mc[order], conf[order]
    keep = (mc >= chunk_lo) & (mc < chunk_hi)
    me, mc, conf = me[keep], mc[keep], conf[keep]
    men_rowptr = np.concatenate([[0], np.cumsum(np.bincount(me, minlength=e))]).astype(np.int64)
    return GraphCSR(ent_rowptr, bb.astype(np.int32), men_rowptr, mc.astype(np.int32), conf)


def graph_queries(n_queries: int, n_docs_global: int, n_seeds: int = 3) -> np.ndarray:
    r = _rng(SEED_QUERIES, 2)
    return r.integers(0, n_entities(n_docs_global), size=(n_queries, n_seeds)).astype(np.int32)


# ----------------------------------------------------------------- tokens
def doc_tokens(start: int, count: int, d_tokens: int = 128, tok_dim: int = 128) -> np.ndarray:
    out = np.empty((count, d_tokens, tok_dim), dtype=np.float16)
    tb = 4096  # token blocks are smaller: 4096 docs * 128 * 128 * 2 B = 128 MiB
    b0, b1 = start // tb, (start + count - 1) // tb
    for b in range(b0, b1 + 1):
        x = _rng(SEED_DOCS, 3, b).standard_normal((tb, d_tokens, tok_dim)).astype(np.float32)
        x /= np.linalg.norm(x, axis=2, keepdims=True)
        lo, hi = max(start, b * tb), min(start + count, (b + 1) * tb)
        out[lo - start:hi - start] = x[lo - b * tb:hi - b * tb].astype(np.float16)
    return out


def query_tokens(n_queries: int, q_tokens: int = 32, tok_dim: int = 128) -> np.ndarray:
    x = _rng(SEED_QUERIES, 3).standard_normal((n_queries, q_tokens, tok_dim)).astype(np.float32)
    x /= np.linalg.norm(x, axis=2, keepdims=True)
    return x.astype(np.float16)


def device_tokens(start, count, d_tokens=128, tok_dim=128, seed=1234 + 3, block=8192):
    """Unit-norm float16 token matrices [count, d_tokens, tok_dim] for global docs
    [start, start+count), generated ON THE DEVICE in blocks seeded by the global block index
    (a shard holds what the unsharded store holds; 32.8 GB per 1M docs never touches the host)."""
    import torch
    out = torch.empty((count, d_tokens, tok_dim), dtype=torch.float16, device="cuda")
    g = torch.Generator(device="cuda")
    for b in range(start // block, (start + count - 1) // block + 1):
        g.manual_seed(seed * 1_000_003 + b)
        x = torch.randn((block, d_tokens, tok_dim), generator=g, device="cuda", dtype=torch.float32)
        x = torch.nn.functional.normalize(x, dim=2).to(torch.float16)
        lo, hi = max(start, b * block), min(start + count, (b + 1) * block)
        out[lo - start:hi - start] = x[lo - b * block:hi - b * block]
    return out
