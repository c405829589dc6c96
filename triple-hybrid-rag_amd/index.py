"""HBM-resident retrieval index and the batched device pipeline.

``GpuIndex`` owns the tensors the HIP scorers read (one document shard per
GPU) and exposes one method per channel plus ``retrieve_batch`` -- dense ->
lexical -> graph -> weighted RRF (-> MaxSim rerank) with every stage a kernel
behind the C ABI and no host round trip between stages.  The reference has no
batch entry point (its ``retrieve()`` is one query per call,
src/voice_agent/rag2/retrieval.py:118-201); the per-query drop-in sits on top
of this in ``rag2/retrieval.py`` + ``backend.py``.

Layout in HBM (per shard of n documents, D dims):
    docs      float32 [n, D]   row-major, the only large array of the dense path
    dnorm     float64 [n]      ||d||  (sequential float64, oracle contract)
    inv_norm  float32 [n]      1/||d|| for the fp32 scan, 0 = no embedding
    lexical   CSR by term: rowptr int64 [V+1], post_doc int32, post_tf int32,
              doclen float32 [n], idf float64 [V] (global), avgdl (global)
    graph     entity CSR (replicated) + entity->chunk mention CSR (this shard's chunks)
    tokens    float16 [n, T_d, 128] late-interaction token matrices
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, Optional

import logging
import os
import warnings

import numpy as np
import torch

from . import _native as N

log = logging.getLogger(__name__)


def floor_width(k: int, n_shards: int) -> int:
    """Lower bounds a shard sends per query for the common floor (thr_dense_floor): twice its fair
    share of the k best, at least 16 -- the k-th largest of the n_shards * m values is then close
    to the true k-th score unless one shard holds most of the k best."""
    m = max(16, 2 * -(-k // max(1, n_shards)))
    return max(1, min(m, N.THR_DENSE_MAX_K, 4096 // max(1, n_shards)))   # (n_shards * m values fit the band kernel's LDS)


@dataclass
class BatchResult:
    ids: torch.Tensor            # int64 [nq, top_k] fused (or reranked) global doc ids, -1 pad
    scores: torch.Tensor         # float64 [nq, top_k] RRF scores (rerank: MaxSim scores)
    counts: torch.Tensor         # int32 [nq]
    channels: Dict[str, tuple] = field(default_factory=dict)  # name -> (scores, ids, counts)
    # queries that needed the exhaustive float64 path: an int, or (batch path, so that a batch
    # needs no host synchronisation) a device int32[1] -- int(result.rescued) reads it back
    rescued: object = 0


class GpuIndex:
    def __init__(self, device: Optional[torch.device] = None, doc_base: int = 0):
        if not torch.cuda.is_available():
            raise N.NativeError("GpuIndex needs a HIP device (no CPU fallback exists)")
        N.load()
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        self.doc_base = int(doc_base)
        self.n_docs = 0
        self.dim = 0
        self.docs = self.dnorm = self.inv_norm = self.docs16 = None
        self.doc_rel_err = 0.0
        self.shortlist = "f32"
        self.lex = None
        self.graph = None
        self.tokens = None
        self.doc_coll = None
        self.tokens_packed = False
        self._ws: Optional[torch.Tensor] = None
        self._ws_rescue: Optional[torch.Tensor] = None
        self._ws_lex: Optional[torch.Tensor] = None
        self._ws_graph: Optional[torch.Tensor] = None
        self._lex_done = None   # event: the last bm25_search's kernels have left the lexical workspace

    # ------------------------------------------------------------ builders
    def _t(self, a, dtype):
        if isinstance(a, torch.Tensor):
            return a.to(device=self.device, dtype=dtype).contiguous()
        a = np.ascontiguousarray(a)
        if not a.flags.writeable:   # a memory-mapped index (index_build.load): read-only is fine,
            with warnings.catch_warnings():   # the tensor is only the source of the device copy
                warnings.simplefilter("ignore", UserWarning)
                return torch.from_numpy(a).to(device=self.device, dtype=dtype)
        return torch.from_numpy(a).to(device=self.device, dtype=dtype)

    SHORTLISTS = ("auto", "f32", "f16", "f16-inline", "exact")
    F16_DIMS = (512, 768, 1024)
    SCAN_DIMS = (256, 512, 768, 1024)   # row lengths the streaming shortlist scans are built for
    AUTO_COPY_FRACTION = 0.25    # of the device's memory: every shard the f16 scans can index (2^25 rows)
    F16_MAX_ROWS = 1 << 25       # per shard: the f16 scans pack (query-in-tile, row) in 32 bits
    F16_MAX_REL_ERR = 2e-3       # ~8x the rounding error of rows in float16's normal range
    DENSE_SHARE = 0.01           # lexical terms held by this share of the docs get per-doc rows

    def set_dense(self, docs, shortlist: str = "auto", derived: Optional[dict] = None) -> "GpuIndex":
        """How the streaming pass picks its shortlist (the returned scores are ALWAYS the float64
        rescoring of the float32 rows, and the per-query certificate covers the scan's error):
          "f32"        float32 rows on the fp32 matrix cores (32 queries per pass);
          "f16-inline" float32 rows rounded to float16 in registers, f16 matrix cores
                       (64 queries per pass, no extra memory);
          "f16"        additionally keeps a float16 copy of the rows and streams that;
          "exact"      no shortlist pass: every row scored in float64 (thr_dense_topk_exact) -- any
                       row length that is a multiple of 4;
          ``derived``: the saved float16 image of a loaded index (export_derived), reused when
                       its layout is the one this build's scan reads.
          "auto"       an f16 flavour when the dimension has an f16 kernel and every row fits the
                       float16 range -- "f16" while the copy is small next to the device's memory
                       (<= AUTO_COPY_FRACTION of it: a quarter, which covers every shard size the f16 scans
                       index), "f16-inline" beyond -- else "f32"; a row length none of the scans is
                       built for (the reference's legacy RAG 1.0 store keeps 4000-d halfvec rows,
                       src/voice_agent/config.py:216, 20260113_halfvec_4000.sql:70-105) goes to
                       "exact" with a logged warning: same bits, O(n * dim) float64 work per query."""
        if shortlist not in self.SHORTLISTS:
            raise ValueError(f"shortlist must be one of {self.SHORTLISTS}")
        self.docs = self._t(docs, torch.float32)
        self.n_docs, self.dim = self.docs.shape
        if self.dim % 4:
            raise N.NativeError(f"row length {self.dim} is not a multiple of 4")
        self.dnorm, self.inv_norm = N.doc_norms(self.docs)
        self.docs16, self.doc_rel_err = (None, 0.0)
        auto = shortlist == "auto"
        if self.dim not in self.SCAN_DIMS and shortlist != "exact":
            if not auto:
                raise N.NativeError(f"shortlist={shortlist!r} needs a row length in {self.SCAN_DIMS}, got "
                                    f"{self.dim}: use shortlist='exact' (or 'auto')")
            log.warning("dense rows of %d dims: no streaming scan is built for that length, every search "
                        "scores all %d rows in float64 (thr_dense_topk_exact)", self.dim, self.n_docs)
            shortlist, auto = "exact", False
        if shortlist == "exact":
            self.shortlist = shortlist
            return self
        if auto:
            shortlist = "f32"
            if self.dim in self.F16_DIMS and self.n_docs < self.F16_MAX_ROWS:
                total = torch.cuda.get_device_properties(self.device).total_memory
                copy_bytes = 2 * self.n_docs * self.dim
                shortlist = "f16" if copy_bytes <= self.AUTO_COPY_FRACTION * total else "f16-inline"
        if shortlist != "f32":
            have = derived and derived.get("docs16") is not None and shortlist == "f16" \
                and derived.get("f16_layout") == N.dense_f16_layout(self.dim)
            if have:   # (a saved index: the float16 image and its measured error come with it)
                self.docs16 = self._t(derived["docs16"], torch.float16)
                self.doc_rel_err = float(derived["doc_rel_err"])
            else:
                self.docs16, self.doc_rel_err = N.dense_quantize_f16(self.docs,
                                                                     keep_copy=shortlist == "f16")
            # float16 holds the rows when no value overflows (|v| < 65504: else the measured error
            # is +inf) and few underflow (rows scaled to ~1e-6 are all subnormals: the error bound
            # would exceed F16_MAX_REL_ERR and no query could be certified)
            if not np.isfinite(self.doc_rel_err) or self.doc_rel_err > self.F16_MAX_REL_ERR:
                if not auto:
                    raise N.NativeError("rows do not fit float16 (values >= 65504 or mostly below "
                                        "6e-5 in magnitude): use shortlist='f32'")
                shortlist, self.docs16, self.doc_rel_err = "f32", None, 0.0
        self.shortlist = shortlist
        return self

    def set_lexical(self, rowptr, post_doc, post_tf, doclen, idf, avgdl: float,
                    k1: float = 1.2, b: float = 0.75, dense_share: Optional[float] = None,
                    derived: Optional[dict] = None) -> "GpuIndex":
        """``dense_share``: a term held by at least this share of the shard's docs also gets
        per-doc rows of impacts / term frequencies (3 bytes per doc and term; 0 = none;
        default DENSE_SHARE, or the A/B knob THR_BM25_DENSE_SHARE).
        ``derived``: the saved bounds / impacts / dense rows of a loaded index (export_derived):
        reused when they were computed for the same k1, b, avgdl and dense share."""
        if dense_share is None:
            dense_share = float(os.environ.get("THR_BM25_DENSE_SHARE", self.DENSE_SHARE))
        self.lex = dict(rowptr=self._t(rowptr, torch.int64), post_doc=self._t(post_doc, torch.int32),
                        post_tf=self._t(post_tf, torch.int32), doclen=self._t(doclen, torch.float32),
                        idf=self._t(idf, torch.float64), avgdl=float(avgdl), k1=float(k1), b=float(b),
                        dense_share=float(dense_share))
        L = self.lex
        tag = [L["avgdl"], L["k1"], L["b"], L["dense_share"]]
        if derived and derived.get("term_ub") is not None and list(derived.get("lexical_tag", [])) == tag:
            L["bounds"] = (self._t(derived["term_ub"], torch.float64), self._t(derived["block_ub"], torch.float64),
                           self._t(derived["post_imp"], torch.uint8))
            L["dense"] = None
            if derived.get("dense_slot") is not None:
                L["dense"] = (self._t(derived["dense_slot"], torch.int32), self._t(derived["dense_imp"], torch.uint8),
                              self._t(derived["dense_tf"], torch.int16), int(derived["dense_stride"]))
        else:
            # per-term / per-128-posting score bounds for the WAND-style pruning of thr_bm25_topk
            L["bounds"] = N.bm25_bounds(L["rowptr"], L["post_doc"], L["post_tf"], L["doclen"], L["idf"],
                                        L["avgdl"], L["k1"], L["b"])
            L["dense"] = N.bm25_dense_terms(L["rowptr"], L["post_doc"], L["post_tf"], L["bounds"][2],
                                            int(L["doclen"].shape[0]), dense_share) if dense_share > 0 else None
        if self.n_docs == 0:
            self.n_docs = int(self.lex["doclen"].shape[0])
        return self

    def set_lexical_rows(self, doc, term, tf, n_vocab: int, n_docs: Optional[int] = None,
                         n_docs_global: Optional[int] = None, group=None, k1: float = 1.2,
                         b: float = 0.75, dense_share: Optional[float] = None) -> "GpuIndex":
        """The lexical side from tokenised ROWS, built on the device (thr_lexical_build): ``doc`` /
        ``term`` / ``tf`` int32 [n_pairs] -- one entry per distinct (doc, term) of a chunk, or one
        per token occurrence with ``tf`` None; doc ids LOCAL to this shard.  A document shard
        passes the corpus' ``n_docs_global`` and its process ``group``: the per-term document
        frequencies and the length total are all-reduced over the shards, so idf and avgdl are the
        whole corpus'  -- idf itself is float64 numpy on the host, the oracle's formula
        to the bit.  The reference leaves this step to PostgreSQL's generated tsvector column +
        GIN index (rag2_schema.sql:146-148, 171-172; rows from rag2/ingest.py:361-470)."""
        import torch.distributed as dist
        n = int(n_docs if n_docs is not None else self.n_docs)
        if n <= 0:
            raise N.NativeError("set_lexical_rows: the shard's doc count is unknown (n_docs)")
        rowptr, post_doc, post_tf, doclen, df = N.lexical_build(
            self._t(doc, torch.int32), self._t(term, torch.int32),
            None if tf is None else self._t(tf, torch.int32), n, int(n_vocab))
        sum_dl = doclen.sum(dtype=torch.float64).reshape(1)
        df_glob = df.clone()
        if group is not None or (dist.is_initialized() and n_docs_global is not None and n_docs_global != n):
            if dist.get_backend(group) == "gloo":   # (CPU rendezvous: rehearsals and tests)
                df_c, sd_c = df_glob.cpu(), sum_dl.cpu()
                dist.all_reduce(df_c, group=group)
                dist.all_reduce(sd_c, group=group)
                df_glob, sum_dl = df_c.to(self.device), sd_c.to(self.device)
            else:
                dist.all_reduce(df_glob, group=group)
                dist.all_reduce(sum_dl, group=group)
        n_glob = int(n_docs_global if n_docs_global is not None else n)
        dfh = df_glob.cpu().numpy().astype(np.float64)
        idf = np.log(1.0 + (float(n_glob) - dfh + 0.5) / (dfh + 0.5))
        avgdl = float(sum_dl.item()) / max(n_glob, 1)
        self.df_local, self.df_global = df, df_glob
        return self.set_lexical(rowptr, post_doc, post_tf, doclen, idf, avgdl if avgdl > 0 else 1.0, k1, b,
                                dense_share)

    def export_derived(self) -> dict:
        """What index set-up computed on the device and a saved index can carry along, as host
        arrays: the float16 image of the rows (+ its layout tag and measured error), the BM25
        bounds / per-posting impacts and the dense-term rows (+ the parameters they hold for)."""
        out: dict = {}
        if self.docs16 is not None and self.shortlist == "f16":
            out.update(docs16=self.docs16.cpu().numpy(), doc_rel_err=float(self.doc_rel_err),
                       f16_layout=N.dense_f16_layout(self.dim))
        if self.lex is not None:
            L = self.lex
            out.update(term_ub=L["bounds"][0].cpu().numpy(), block_ub=L["bounds"][1].cpu().numpy(),
                       post_imp=L["bounds"][2].cpu().numpy(),
                       lexical_tag=[L["avgdl"], L["k1"], L["b"], L["dense_share"]])
            if L["dense"] is not None:
                out.update(dense_slot=L["dense"][0].cpu().numpy(), dense_imp=L["dense"][1].cpu().numpy(),
                           dense_tf=L["dense"][2].cpu().numpy(), dense_stride=int(L["dense"][3]))
        return out

    def set_collections(self, doc_coll) -> "GpuIndex":
        """Per-document collection id (int32 [n], any non-negative labelling): the
        ``p_collection`` filter of the two RPCs (rag2_schema.sql:368-373, 404-408) is applied
        on the device BEFORE the ranking -- per query, -1 = unfiltered."""
        self.doc_coll = self._t(doc_coll, torch.int32)
        return self

    def set_graph(self, ent_rowptr, ent_col, men_rowptr, men_chunk, men_conf) -> "GpuIndex":
        self.graph = dict(ent_rowptr=self._t(ent_rowptr, torch.int64),
                          ent_col=self._t(ent_col, torch.int32),
                          men_rowptr=self._t(men_rowptr, torch.int64),
                          men_chunk=self._t(men_chunk, torch.int32),
                          men_conf=self._t(men_conf, torch.float32))
        return self

    def _graph_transposed(self):
        """Chunk-major copy of this shard's mentions (the capacity-free third tier of
        thr_graph_topk), built on first use: the shard's chunk count must be known."""
        G = self.graph
        if G.get("n_chunks") != self.n_docs:
            G["n_chunks"] = self.n_docs
            G["transposed"] = N.graph_transpose_mentions(G["men_rowptr"], G["men_chunk"],
                                                         G["men_conf"], self.doc_base, self.n_docs)
        return G["transposed"]

    def set_tokens(self, dtok, pack: bool = True) -> "GpuIndex":
        """Late-interaction token store f16 [n, d_tokens, tok_dim].  pack=True keeps it in the
        fragment-major layout of thr_maxsim_pack (the row-major copy is dropped)."""
        tok = self._t(dtok, torch.float16)
        self.tokens_packed = bool(pack)
        self.tokens = N.maxsim_pack(tok) if pack else tok
        return self

    # ------------------------------------------------------------ channels
    def _workspace(self, nbytes: int) -> torch.Tensor:
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        return self._ws

    def reserve(self, n_queries: int, k: int, kprime: Optional[int] = None) -> "GpuIndex":
        """Allocate the dense workspaces for batches of ``n_queries`` up front (index set-up), so
        that no search pays a device allocation."""
        if self.shortlist == "exact":
            return self
        if self.shortlist != "f32":
            kp = min(N.THR_DENSE_MAX_K, max(k, kprime or (k + 92)))
            self._workspace(N.dense_f16_workspace_bytes(self.n_docs, self.dim, n_queries, kp))
        else:
            kp = min(N.THR_DENSE_MAX_K, max(k, kprime or (k + 28)))
            self._workspace(N.dense_workspace_bytes(self.n_docs, self.dim, n_queries, kp))
        need = N.dense_rescue_workspace_bytes(n_queries, k)
        if self._ws_rescue is None or self._ws_rescue.numel() < need:
            self._ws_rescue = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self

    def max_batch(self) -> int:
        """Queries one dense_search call hands to the scan at once (larger batches are split)."""
        if self.shortlist == "f16":
            return N.dense_f16_max_queries(self.dim, True)
        return 1 << 30

    def dense_search(self, queries: torch.Tensor, k: int, kprime: Optional[int] = None,
                     rescue: bool = True, sync: bool = True, collections=None, floor_exchange=None):
        """Exact cosine top-k -> (scores f64, ids i64, counts i32, n_rescued).  Queries the
        error-bound certificate cannot prove exact (massive ties / duplicates) are redone on the
        exhaustive float64 path, on the device (thr_dense_rescue: no host read-back).
        n_rescued is an int, or with sync=False the device int32[1] it would be read from.
        collections: int32 [nq] collection id per query (-1 = unfiltered), applied before the
        ranking (set_collections).
        floor_exchange: (callable, n_shards) of a DOCUMENT-SHARDED index -- the callable maps this
        shard's float32 [nq, m] lower bounds to all the shards' [n_shards, nq, m] (an all-gather).
        The search is then split around that one exchange (thr_dense_shortlist_f16 / thr_dense_floor
        / thr_dense_finish_f16): rows that cannot be among the k best of ALL shards are not
        rescored, the returned list may hold fewer than k rows and is this shard's part of the
        global top-k (merge the shards' lists with thr_merge_topk).  Every shard must make the same
        sequence of calls.  Used by the f16 scans only."""
        queries = self._t(queries, torch.float32)
        nq = queries.shape[0]
        self._shortlist_of = None     # (the workspace's candidate lists are about to be overwritten)
        if self.shortlist == "f16":
            # one call of the copy scan takes at most dense_f16_max_queries queries (its
            # candidate-segment offsets are 32 bits): larger batches go through in pieces
            step = self.max_batch()
            if nq > step:
                S, I, cnt, _ = N._alloc_out(nq, k, self.device)
                n_rescued = 0 if sync else torch.zeros(1, dtype=torch.int32, device=self.device)
                coll = None if collections is None else self._t(collections, torch.int32)
                for lo in range(0, nq, step):
                    hi = min(nq, lo + step)
                    s_, i_, c_, r_ = self.dense_search(queries[lo:hi], k, kprime, rescue, sync,
                                                       None if coll is None else coll[lo:hi],
                                                       floor_exchange)
                    S[lo:hi], I[lo:hi], cnt[lo:hi] = s_, i_, c_
                    n_rescued = n_rescued + r_
                return S, I, cnt, n_rescued
        dc, qc = self._qcoll(collections, nq)
        if self.shortlist == "exact":
            S, I, cnt, _ = N.dense_topk_exact(self.docs, self.dnorm, queries, k, self.doc_base, dc, qc)
            return S, I, cnt, (0 if sync else torch.zeros(1, dtype=torch.int32, device=self.device))
        if self.shortlist != "f32":
            # tau must sit clearly below the k-th score for the quantisation-aware certificate:
            # k' = 192 puts it ~6e-3 below on a 1M-row corpus, ~6x the f16 error bound
            kp = min(N.THR_DENSE_MAX_K, max(k, kprime or (k + 92)))
            ws = self._workspace(N.dense_f16_workspace_bytes(self.n_docs, self.dim,
                                                             queries.shape[0], kp))
            if floor_exchange is not None:
                exchange, n_shards = floor_exchange
                lb = N.dense_shortlist_f16(self.docs, self.docs16, self.doc_rel_err, self.inv_norm,
                                           queries, kp, floor_width(k, n_shards), ws,
                                           doc_coll=dc, query_coll=qc)
                S, I, cnt, flg = N.dense_finish_f16(self.docs, self.docs16, self.doc_rel_err, self.dnorm,
                                                    self.inv_norm, queries, k, kp, None, self.doc_base,
                                                    ws, doc_coll=dc, query_coll=qc, lb_all=exchange(lb))
            else:
                S, I, cnt, flg = N.dense_topk_f16(self.docs, self.docs16, self.doc_rel_err, self.dnorm,
                                                  self.inv_norm, queries, k, kp, self.doc_base, ws,
                                                  doc_coll=dc, query_coll=qc)
        else:
            kp = min(N.THR_DENSE_MAX_K, max(k, kprime or (k + 28)))
            ws = self._workspace(N.dense_workspace_bytes(self.n_docs, self.dim, queries.shape[0], kp))
            S, I, cnt, flg = N.dense_topk(self.docs, self.dnorm, self.inv_norm, queries, k, kp,
                                          self.doc_base, ws, doc_coll=dc, query_coll=qc)
        n_rescued = 0
        if rescue:
            need = N.dense_rescue_workspace_bytes(queries.shape[0], k)
            if self._ws_rescue is None or self._ws_rescue.numel() < need:
                self._ws_rescue = torch.empty(need, dtype=torch.uint8, device=self.device)
            n_rescued = N.dense_rescue(self.docs, self.dnorm, queries, S, I, cnt, flg,
                                       self.doc_base, self._ws_rescue, doc_coll=dc, query_coll=qc)
            if sync:
                n_rescued = int(n_rescued)
        return S, I, cnt, n_rescued

    # The two halves of dense_search(floor_exchange=...) on their own, for a caller that holds
    # several shards in ONE process (tests, bench.py's shard proxy): shortlist on every shard,
    # stack the results, thr_dense_floor, finish on every shard.
    def _f16_call(self, queries, k, kprime, collections):
        if self.shortlist not in ("f16", "f16-inline"):
            raise N.NativeError("the shard floor is built for the f16 scans")
        queries = self._t(queries, torch.float32)
        if queries.shape[0] > self.max_batch():
            raise N.NativeError("dense_shortlist/finish: one scan batch at a time (max_batch())")
        kp = min(N.THR_DENSE_MAX_K, max(k, kprime or (k + 92)))
        ws = self._workspace(N.dense_f16_workspace_bytes(self.n_docs, self.dim, queries.shape[0], kp))
        dc, qc = self._qcoll(collections, queries.shape[0])
        return queries, kp, ws, dc, qc

    def dense_shortlist(self, queries: torch.Tensor, k: int, n_shards: int, kprime: Optional[int] = None,
                        collections=None) -> torch.Tensor:
        queries, kp, ws, dc, qc = self._f16_call(queries, k, kprime, collections)
        self._shortlist_of = None
        lb = N.dense_shortlist_f16(self.docs, self.docs16, self.doc_rel_err, self.inv_norm, queries, kp,
                                   floor_width(k, n_shards), ws, doc_coll=dc, query_coll=qc)
        # what the candidate lists in the workspace belong to: dense_finish refuses anything else
        self._shortlist_of = (queries.shape[0], kp, collections is not None, ws.data_ptr())
        return lb

    def dense_finish(self, queries: torch.Tensor, k: int, gfloor: Optional[torch.Tensor] = None,
                     kprime: Optional[int] = None, rescue: bool = True, collections=None,
                     lb_all: Optional[torch.Tensor] = None):
        """-> (scores, ids, counts, flags BEFORE the rescue, n_rescued device int32[1] or 0).
        The floor: gfloor [nq], or the gathered bounds lb_all [n_shards, nq, m] themselves."""
        queries, kp, ws, dc, qc = self._f16_call(queries, k, kprime, collections)
        if getattr(self, "_shortlist_of", None) != (queries.shape[0], kp, collections is not None, ws.data_ptr()):
            raise N.NativeError("dense_finish: the workspace does not hold the candidate lists of a matching "
                                "dense_shortlist call (same batch size, k, collections; no other dense "
                                "search on this index in between)")
        S, I, cnt, flg = N.dense_finish_f16(self.docs, self.docs16, self.doc_rel_err, self.dnorm,
                                            self.inv_norm, queries, k, kp, gfloor, self.doc_base, ws,
                                            doc_coll=dc, query_coll=qc, lb_all=lb_all)
        flags0 = flg.clone()
        n_rescued = 0
        if rescue:
            need = N.dense_rescue_workspace_bytes(queries.shape[0], k)
            if self._ws_rescue is None or self._ws_rescue.numel() < need:
                self._ws_rescue = torch.empty(need, dtype=torch.uint8, device=self.device)
            n_rescued = N.dense_rescue(self.docs, self.dnorm, queries, S, I, cnt, flg, self.doc_base,
                                       self._ws_rescue, doc_coll=dc, query_coll=qc)
        return S, I, cnt, flags0, n_rescued

    def scan_probe(self, queries: torch.Tensor) -> None:
        """Launch ONLY the streaming scan kernel of the last dense_search (same workspace, so the
        thresholds tau are the ones that search computed): the timing/roofline probe."""
        if self._ws is None or self.shortlist == "exact":
            raise N.NativeError("scan_probe needs a preceding dense_search on this index (and a shortlist scan)")
        queries = self._t(queries, torch.float32)
        if self.shortlist != "f32":
            N.dense_scan_probe_f16(self.docs, self.docs16, self.inv_norm, queries, self._ws)
        else:
            N.dense_scan_probe(self.docs, self.inv_norm, queries, self._ws)

    def _qcoll(self, collections, nq: int):
        if collections is None:
            return None, None
        if self.doc_coll is None:
            raise N.NativeError("collection filter without set_collections()")
        qc = self._t(collections, torch.int32)
        if qc.shape != (nq,):
            raise N.NativeError("collections: one id per query")
        return self.doc_coll, qc

    def bm25_search(self, query_terms: torch.Tensor, k: int, collections=None,
                    conjunctive: bool = False, prune: bool = True, dense_rows: bool = True):
        """collections: int32 [nq] collection id per query (-1 = unfiltered) or None."""
        L = self.lex
        qt = self._t(query_terms, torch.int32)
        dc, qc = self._qcoll(collections, qt.shape[0])
        need = N.bm25_workspace_bytes(qt.shape[0], qt.shape[1], k)
        # ONE lexical workspace per index, used from whichever stream the caller is on (the main
        # one, or the side stream of side_channels / GpuIndexClient's deferred RPC): the stream of
        # this call waits for the previous call's kernels before its memset touches the workspace
        cur = torch.cuda.current_stream(self.device)
        if self._lex_done is not None:
            cur.wait_event(self._lex_done)
        if self._ws_lex is None or self._ws_lex.numel() < need:   # kept: no allocation per search
            self._ws_lex = torch.empty(need, dtype=torch.uint8, device=self.device)
        try:
            return self._bm25_call(L, qt, k, dc, qc, conjunctive, prune, dense_rows)
        finally:
            if self._lex_done is None:
                self._lex_done = torch.cuda.Event()
            self._lex_done.record(cur)

    def _bm25_call(self, L, qt, k, dc, qc, conjunctive, prune, dense_rows):
        return N.bm25_topk(L["rowptr"], L["post_doc"], L["post_tf"], L["doclen"], L["idf"],
                           L["avgdl"], qt, k, self.doc_base, L["k1"], L["b"],
                           bounds=L["bounds"] if prune else None, conjunctive=conjunctive,
                           doc_coll=dc, query_coll=qc, workspace=self._ws_lex,
                           dense=L["dense"] if prune and dense_rows else None)

    def graph_search(self, query_seeds: torch.Tensor, k: int, hops: int = 2):
        G = self.graph
        # three tiers on the device (small / full on-chip capacities, then a capacity-free walk
        # in global memory): no flag to read back, nothing to raise in the middle of a batch
        seeds = self._t(query_seeds, torch.int32)
        need = N.graph_workspace_bytes(seeds.shape[0], G["ent_rowptr"].shape[0] - 1)
        if self._ws_graph is None or self._ws_graph.numel() < need:   # kept: no allocation per search
            self._ws_graph = torch.empty(need, dtype=torch.uint8, device=self.device)
        S, I, cnt, _ = N.graph_topk(G["ent_rowptr"], G["ent_col"], G["men_rowptr"],
                                    G["men_chunk"], G["men_conf"], seeds, hops, k, self.doc_base,
                                    self.n_docs, transposed=self._graph_transposed(),
                                    workspace=self._ws_graph)
        return S, I, cnt

    def maxsim(self, qtok: torch.Tensor, cand_global_ids: torch.Tensor) -> torch.Tensor:
        """MaxSim of each query against its candidate docs (global ids; ids outside this
        shard or negative score -inf)."""
        return N.maxsim_ids(self._t(qtok, torch.float16), self.tokens, self._t(cand_global_ids, torch.int64),
                            self.doc_base, packed=self.tokens_packed)

    # ------------------------------------------------------------ pipeline
    def side_channels(self, query_terms, lexical_top_k: int, query_seeds, graph_top_k: int, hops: int):
        """The lexical and graph channels of a batch on a second HIP stream, so that they run
        beside the dense channel instead of after it: they do not depend on it before the fusion,
        they are latency-bound (one workgroup per query, a few per CU), and the dense pipeline has
        stretches that leave CUs idle (sample pass, shortlist, the gather-bound rescoring, the
        tail of the scan).  Returns (lexical result or None, graph result or None, join): call
        ``join()`` on the main stream before the results are read there.  THR_SIDE_STREAM=0 keeps
        everything on one stream."""
        want_lex = query_terms is not None and self.lex is not None
        want_gra = query_seeds is not None and self.graph is not None
        if not (want_lex or want_gra):
            return None, None, (lambda: None)
        if os.environ.get("THR_SIDE_STREAM") == "0" or self.device.type != "cuda":
            lex = self.bm25_search(query_terms, lexical_top_k) if want_lex else None
            gra = self.graph_search(query_seeds, graph_top_k, hops) if want_gra else None
            return lex, gra, (lambda: None)
        self.side_stream()
        main = torch.cuda.current_stream(self.device)
        self._side.wait_stream(main)            # the inputs were produced on the main stream
        for t in (query_terms, query_seeds):
            if isinstance(t, torch.Tensor) and t.is_cuda:
                t.record_stream(self._side)
        with torch.cuda.stream(self._side):
            lex = self.bm25_search(query_terms, lexical_top_k) if want_lex else None
            gra = self.graph_search(query_seeds, graph_top_k, hops) if want_gra else None
        for res in (lex, gra):
            if res is not None:
                for t in res:
                    t.record_stream(main)       # allocated on the side stream, read on the main one

        def join():
            torch.cuda.current_stream(self.device).wait_stream(self._side)
        return lex, gra, join

    def side_stream(self) -> "torch.cuda.Stream":
        if getattr(self, "_side", None) is None:
            # a high-priority queue: its short kernels get their CUs first and are gone before
            # the scan's one-workgroup-per-CU launch needs them (triple + rerank step 4.57 ms on
            # one stream, 4.42 with an equal-priority side stream -- THR_SIDE_STREAM=eq --, 4.32 so)
            eq = os.environ.get("THR_SIDE_STREAM") == "eq"
            self._side = torch.cuda.Stream(device=self.device, priority=0 if eq else -1)
        return self._side

    def retrieve_batch(self, queries: torch.Tensor, query_terms: Optional[torch.Tensor] = None,
                       query_seeds: Optional[torch.Tensor] = None, top_k: int = 10,
                       semantic_top_k: int = 100, lexical_top_k: int = 50, graph_top_k: int = 50,
                       weights: Optional[Dict[str, float]] = None, hops: int = 2,
                       qtok: Optional[torch.Tensor] = None, rerank_top_k: int = 100,
                       rescue: bool = True) -> BatchResult:
        """plan.semantic/lexical/graph_top_k = 100/50/50 and weights 0.7/0.8/1.0 are the
        reference's QueryPlan defaults (src/voice_agent/rag2/query_planner.py:23-50)."""
        w = {"lexical": 0.7, "semantic": 0.8, "graph": 1.0}
        w.update(weights or {})
        ch = {}
        lex, gra, join = self.side_channels(query_terms, lexical_top_k, query_seeds, graph_top_k, hops)
        Ss, Is, Cs, nres = self.dense_search(queries, semantic_top_k, rescue=rescue, sync=False)
        ch["semantic"] = (Ss, Is, Cs)
        join()
        Il = Ig = None
        if lex is not None:
            ch["lexical"] = lex
            Il = lex[1]
        if gra is not None:
            ch["graph"] = gra
            Ig = gra[1]
        rerank = qtok is not None and self.tokens is not None
        n_fused = max(rerank_top_k, top_k) if rerank else top_k
        ids, sc, _, cnt = N.rrf_fuse(Il, Is, Ig, n_fused, w["lexical"], w["semantic"], w["graph"])
        if rerank:
            # MaxSim of the fused top rerank_top_k, then the reference's stable descending sort
            # on ``rerank_score or 0`` (retrieval.py:449-455), on the device
            ids, sc, cnt = N.rerank_order(self.maxsim(qtok, ids), ids, cnt, top_k)
        return BatchResult(ids, sc, cnt, ch, nres)
