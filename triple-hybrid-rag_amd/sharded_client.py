"""The drop-in backend seam over a DOCUMENT-SHARDED index (SURVEY.md section 8e + 8b).

``GpuIndexClient`` answers the retriever's Supabase calls from ONE shard.  A 10M-doc deployment
over the 8 GPUs of a node has one process -- one ``GpuIndex`` -- per GPU; this module gives that
deployment the same duck type:

* the FRONT rank (0 of its group) owns a ``ShardedIndexClient``: ``rpc()`` / ``table()`` /
  ``find_entities()`` / ``graph_chunks()`` / ``maxsim_scores()`` exactly as ``GpuIndexClient``
  (reference seams: src/voice_agent/rag2/retrieval.py:103-108, 282-312, 339-341, 390-392), so
  ``RAG2Retriever.retrieve()``, ``Reranker.rerank()`` and ``search_knowledge_base()`` run over it
  unchanged.  It holds the host-side row payloads of the WHOLE corpus (what the SQL rows carry:
  ids, text, page, parents) and the global vocabulary / entity names; the device arrays it holds
  are its own shard's;
* every other rank sits in ``ShardWorker.serve()``.

One request = one broadcast + one all-gather:

    front: tokenise / look up -> ONE fixed-size int32 message (header + payload: the query vector's
           bits, or the term ids, or the seed entities, or candidate ids + query token matrix)
           -> ``broadcast`` to the group
    every rank (front included): the channel's kernel on its own shard (thr_dense_topk_f16 /
           thr_bm25_topk / thr_graph_topk / thr_maxsim_ids) -> fixed-shape (score, global id) tile
    ``all_gather`` of the tiles (RCCL over xGMI; distributed.gather_topk / gather_rows)
    front: thr_merge_topk under (score desc, id asc) -> rows from the host store.

The collective sequence of a request is fixed by its op code, so the ranks cannot disagree on
what comes next; a worker that fails inside a kernel still takes part in the all-gather (with
an empty tile) and reports the failure in its tile's header slot, so the front raises and the
group does not hang.  ``close()`` stops the workers.

Collection filter: the name -> id mapping must be the same on every rank, so the collection
names are fixed at construction (``collection_names``), not derived from each shard's rows.
"""
from __future__ import annotations

import logging
from typing import Any, Callable, Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.distributed as dist

from . import _native as N
from .backend import CorpusStore, GpuIndexClient
from .distributed import gather_rows, gather_topk

log = logging.getLogger(__name__)

OP_STOP, OP_SEMANTIC, OP_LEXICAL, OP_GRAPH, OP_MAXSIM = 0, 1, 2, 3, 4
HEADER = 8   # int32 words: op, k, aux (hops | conjunctive | n candidates), collection id, payload words, token rows, token dim, sequence


def _empty_store() -> CorpusStore:
    return CorpusStore(child_ids=[], parent_ids=[], document_ids=[], texts=[], pages=[], modalities=[])


class _ShardEndpoint(GpuIndexClient):
    """What the front and the workers share: the message buffer, the per-op shard step, the
    exchange.  ``index`` needs the GpuIndex search methods (dense_search / bm25_search /
    graph_search / maxsim), ``dim``, ``device`` and ``doc_coll``."""

    defers_readback = False   # the lexical rows come back with the exchange: nothing to defer
    sets_collections = False  # the store is the WHOLE corpus (front) or empty (workers): each rank's
    #                           GpuIndex gets its shard's ids from set_collections() at index set-up

    def __init__(self, index, store: CorpusStore, group=None, front: int = 0,
                 collection_names: Optional[Sequence[str]] = None, org_id: Optional[str] = None,
                 token_embedder: Any = None, lexical_and: bool = False,
                 max_token_words: int = 32 * 128 // 2, merge_fn: Optional[Callable] = None):
        """front: rank (within ``group``) that receives the requests.
        max_token_words: int32 words reserved for a query's float16 token matrix (32 x 128).
        merge_fn(scores [W, 1, k], ids [W, 1, k], k) -> (scores [1, k], ids [1, k]): the per-query
        merge of the gathered shard lists; default thr_merge_topk (tests of the control flow on a
        CPU-only box inject their own)."""
        super().__init__(index, store, org_id=org_id, token_embedder=token_embedder,
                         lexical_and=lexical_and, collection_names=collection_names)
        if not dist.is_initialized():
            raise RuntimeError("ShardedIndexClient needs an initialised torch.distributed group")
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.front = int(front)
        self._front_global = dist.get_global_rank(group, self.front) if group is not None else self.front
        self._merge_fn = merge_fn or (lambda S, I, k: N.merge_topk(S, I, k)[:2])
        # messages travel on the collective backend's device: HBM for RCCL, host memory for gloo
        self._msg_dev = torch.device("cpu") if dist.get_backend(group) == "gloo" else index.device
        self._words = HEADER + max(int(index.dim), N.THR_BM25_MAX_TERMS, N.THR_GRAPH_MAX_SEEDS,
                                   2 * N.THR_RRF_MAX_PER_CHANNEL + int(max_token_words))
        self._seq = 0
        self._closed = False

    # ---------------------------------------------------------------- messages
    def _broadcast(self, msg: Optional[torch.Tensor]) -> torch.Tensor:
        if msg is None:
            msg = torch.empty(self._words, dtype=torch.int32, device=self._msg_dev)
        dist.broadcast(msg, src=self._front_global, group=self.group)
        return msg

    def _message(self, op: int, k: int, aux: int, coll: int, payload: np.ndarray,
                 tok_rows: int = 0, tok_dim: int = 0) -> torch.Tensor:
        words = np.ascontiguousarray(payload).view(np.int32).reshape(-1)
        if HEADER + words.size > self._words:
            raise ValueError(f"request of {words.size} payload words exceeds the message buffer "
                             f"({self._words - HEADER}): raise max_token_words")
        self._seq += 1
        buf = np.zeros(self._words, dtype=np.int32)
        buf[:HEADER] = (op, k, aux, coll, words.size, tok_rows, tok_dim, self._seq & 0x7FFFFFFF)
        buf[HEADER:HEADER + words.size] = words
        return torch.from_numpy(buf).to(self._msg_dev)

    # ---------------------------------------------------------------- the shard step
    def _step(self, msg: torch.Tensor):
        """Run the request in ``msg`` on this rank's shard and take part in its exchange.
        -> what the front needs: merged (scores, ids) lists, or the [W, n] MaxSim rows."""
        head = msg[:HEADER].tolist()
        op, k, aux, coll, n_words, tok_rows, tok_dim = head[:7]
        dev = self.index.device
        body = msg[HEADER:HEADER + n_words]
        err = None
        S = I = ms = None
        try:
            qc = None
            if coll != -1 and self.index.doc_coll is not None:
                qc = torch.tensor([coll], dtype=torch.int32, device=dev)
            if op == OP_SEMANTIC:
                q = body.view(torch.float32).reshape(1, -1).to(dev)
                S, I, _, _ = self.index.dense_search(q, k, collections=qc, sync=False)
            elif op == OP_LEXICAL:
                qt = body.reshape(1, -1).to(dev)
                S, I, _ = self.index.bm25_search(qt, k, collections=qc, conjunctive=bool(aux))
            elif op == OP_GRAPH:
                S, I, _ = self.index.graph_search(body.reshape(1, -1).to(dev), k, aux)
            elif op == OP_MAXSIM:
                cand = body[:2 * aux].view(torch.int64).reshape(1, aux).to(dev)
                qtok = body[2 * aux:].view(torch.float16).reshape(1, tok_rows, tok_dim).to(dev)
                ms = self.index.maxsim(qtok, cand)
            else:
                raise ValueError(f"unknown op {op}")
        except Exception as exc:  # noqa: BLE001 -- the exchange below must still happen on every rank
            err = exc
            log.error("shard %d: request %d failed: %s", self.rank, op, exc)
        if op == OP_MAXSIM:
            if ms is None:   # NaN marks the failure (a shard that does not own a doc sends -inf)
                ms = torch.full((1, aux), float("nan"), dtype=torch.float32, device=dev)
            rows = gather_rows(ms.to(torch.float32), self.group)          # [W, 1, n]
            if err is not None:
                raise err
            return rows
        if S is None:
            S = torch.full((1, k), float("nan"), dtype=torch.float64, device=dev)
            I = torch.full((1, k), -1, dtype=torch.int64, device=dev)
        Sg, Ig = gather_topk(S, I, self.group)                            # [W, 1, k] each
        if err is not None:
            raise err
        return Sg, Ig


class ShardWorker(_ShardEndpoint):
    """A non-front rank: ``serve()`` answers the front's requests until it closes."""

    def __init__(self, index, group=None, front: int = 0, collection_names=None, **kw):
        super().__init__(index, _empty_store(), group=group, front=front,
                         collection_names=collection_names, **kw)
        if self.rank == self.front:
            raise ValueError("the front rank owns a ShardedIndexClient, not a ShardWorker")

    def serve(self) -> int:
        """-> the number of requests served.  A failing request is logged and answered with a
        NaN tile (the front raises); the loop goes on."""
        served = 0
        while True:
            msg = self._broadcast(None)
            if int(msg[0]) == OP_STOP:
                return served
            try:
                self._step(msg)
            except Exception:  # noqa: BLE001 -- logged in _step; the front sees the NaN tile
                pass
            served += 1


class ShardedIndexClient(_ShardEndpoint):
    """The ``GpuIndexClient`` duck type on the front rank of a document-sharded index."""

    def __init__(self, index, store: CorpusStore, group=None, front: int = 0, **kw):
        """store: the row payloads of the WHOLE corpus (``doc_base`` 0), the global vocabulary and
        the entity names -- what the front needs to tokenise, look up entities and build rows."""
        super().__init__(index, store, group=group, front=front, **kw)
        if self.rank != self.front:
            raise ValueError("only the front rank answers requests; the others run ShardWorker.serve()")

    def close(self) -> None:
        if not self._closed:
            self._closed = True
            self._broadcast(self._message(OP_STOP, 0, 0, -1, np.zeros(0, dtype=np.int32)))

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ---------------------------------------------------------------- requests
    def _request_topk(self, op, k, aux, collection, payload):
        coll = -1
        if collection is not None and self.index.doc_coll is not None:
            coll = self._coll_id.get(collection, -3)
        Sg, Ig = self._step(self._broadcast(self._message(op, k, aux, coll, payload)))
        if bool(torch.isnan(Sg).any()):
            raise N.NativeError("a shard failed to answer the request (see its log)")
        Sm, Im = self._merge_fn(Sg, Ig, k)
        scores, ids = Sm[0].tolist(), Im[0].tolist()
        n = 0
        while n < len(ids) and ids[n] >= 0:
            n += 1
        return scores[:n], ids[:n]

    def _semantic(self, embedding, limit: int, collection):
        if len(embedding) != self.index.dim:
            raise ValueError(f"embedding has {len(embedding)} dims, index has {self.index.dim}")
        k = min(N.THR_DENSE_MAX_K, limit)
        scores, ids = self._request_topk(OP_SEMANTIC, k, 0, collection,
                                         np.asarray(embedding, dtype=np.float32))
        return self._rows(ids, scores, len(ids), "similarity", limit)

    def _lexical(self, query: str, limit: int, collection, defer: bool = False):
        terms = self._query_terms(query)
        if terms is None:
            return []
        k = min(N.THR_TOPK_MAX, limit)
        padded = np.array(terms + [-1] * (N.THR_BM25_MAX_TERMS - len(terms)), dtype=np.int32)
        scores, ids = self._request_topk(OP_LEXICAL, k, int(self.lexical_and), collection, padded)
        return self._rows(ids, scores, len(ids), "rank", limit)

    def _image(self, embedding, limit: int):
        return []   # (the legacy image channel is not sharded: a separate small index)

    def graph_chunks(self, seeds: List[int], top_k: int, hops: int = 2) -> List[str]:
        padded = np.array(list(seeds) + [-1] * (N.THR_GRAPH_MAX_SEEDS - len(seeds)), dtype=np.int32)
        _, ids = self._request_topk(OP_GRAPH, min(N.THR_TOPK_MAX, top_k), hops, None, padded)
        return [self.store.child_ids[int(g) - self.store.doc_base] for g in ids]

    def maxsim_scores(self, query: str, child_ids: List[str]) -> List[float]:
        """Every shard scores the candidates it owns (-inf for the others), one all-gather, the
        maximum per candidate -- SURVEY 8e's rerank leg, one query at a time."""
        if self.token_embedder is None:
            raise RuntimeError("no token embedder: late-interaction rerank unavailable")
        qtok = np.ascontiguousarray(self.token_embedder.embed_query_tokens(query), dtype=np.float16)
        rows = [self.store.row_index(c) for c in child_ids]
        cand = np.array([self.store.doc_base + i if i is not None else -1 for i in rows], dtype=np.int64)
        if len(cand) > N.THR_RRF_MAX_PER_CHANNEL:
            raise ValueError(f"at most {N.THR_RRF_MAX_PER_CHANNEL} candidates per rerank request")
        payload = np.concatenate([cand.view(np.int32), qtok.reshape(-1).view(np.int32)])
        got = self._step(self._broadcast(self._message(OP_MAXSIM, 0, len(cand), -1, payload,
                                                        qtok.shape[0], qtok.shape[1])))
        if bool(torch.isnan(got).any()):
            raise N.NativeError("a shard failed to score the rerank candidates (see its log)")
        best = got.max(dim=0).values[0].tolist()
        return [float(v) / qtok.shape[0] if np.isfinite(v) else 0.5 for v in best]
