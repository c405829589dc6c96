"""Document-sharded retrieval across the GPUs of one node.

One process per GPU.  Every channel's score for a document depends only on that
document's row / postings / mentions plus query-global scalars, so the corpus is
split into contiguous doc ranges [g*N/G, (g+1)*N/G); each rank produces its
exact per-query top-k (global ids) and ONE exchange step per channel merges
them: an all-gather of fixed-shape (score f64, id i64) tiles over RCCL/xGMI,
then a merge kernel under (score desc, id asc).  Ranks (not scores) feed RRF,
so the merge completes per channel before fusion.  The payload is KBs..MBs per
batch, i.e. latency-bound on xGMI; the >=6x scaling comes from the 8x smaller
per-GPU scan, not from the collective.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch
import torch.distributed as dist

from . import _native as N
from .index import BatchResult, GpuIndex


def shard_range(n_docs: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous doc range of ``rank``: [lo, hi)."""
    per = (n_docs + world - 1) // world
    lo = min(n_docs, rank * per)
    return lo, min(n_docs, lo + per)


def layout_2d(rank: int, world: int, doc_shards: int):
    """Doc-shards x query-replicas layout of ``world`` ranks: the corpus is split into
    ``doc_shards`` document ranges only as far as memory asks for (288 GB per MI355X hold
    ~90M x 768 float32 rows), and the world // doc_shards groups of that many ranks are
    REPLICAS that serve different query batches -- the per-batch costs that do not shrink with
    the shard (threshold, shortlist, rescoring, exchange) are then paid once per replica instead
    of once per GPU.  -> (shard index, replica index, ranks of this rank's replica).
    doc_shards == world is the pure document-sharded layout (one replica)."""
    if doc_shards < 1 or world % doc_shards:
        raise ValueError("doc_shards must divide the world size")
    replica, shard = divmod(rank, doc_shards)
    return shard, replica, list(range(replica * doc_shards, (replica + 1) * doc_shards))


def auto_doc_shards(world: int, n_docs: int, min_shard_docs: int = 500_000) -> int:
    """How many document shards ``world`` GPUs should cut an ``n_docs`` corpus into: the largest
    divisor of the world that leaves every shard >= ``min_shard_docs`` rows (a shard's step is the
    scan, which shrinks with the shard, plus per-batch work that does not; below ~500 K rows the
    fixed part dominates and more replicas beat more shards) -- but at least 2 when there is more
    than one GPU, so that the exchange stays in the path.  The other world / shards groups are
    replicas that serve their own query batches (layout_2d)."""
    best = max(d for d in range(1, world + 1)
               if world % d == 0 and (d == 1 or n_docs // d >= min_shard_docs))
    if world > 1 and best == 1 and n_docs >= 2:
        best = min(d for d in range(2, world + 1) if world % d == 0)
    return best


def replica_groups(world: int, doc_shards: int):
    """One process group per replica (every rank must create all of them, in this order)."""
    return [dist.new_group(list(range(r * doc_shards, (r + 1) * doc_shards)))
            for r in range(world // doc_shards)]


def _as_tile(scores: torch.Tensor, ids: torch.Tensor) -> torch.Tensor:
    """The [2, nq, k] int64 tile whose halves ``scores`` (as bit patterns) and ``ids`` are.  The
    kernels' outputs already are the two halves of one allocation (_native._alloc_out): then
    this is a view, otherwise one packing copy."""
    nq, k = ids.shape
    if (scores.is_contiguous() and ids.is_contiguous() and scores.dtype == torch.float64
            and scores.untyped_storage().data_ptr() == ids.untyped_storage().data_ptr()
            and ids.data_ptr() - scores.data_ptr() == nq * k * 8):
        return torch.as_strided(scores.view(torch.int64), (2, nq, k), (nq * k, k, 1))
    return torch.stack([scores.contiguous().view(torch.int64), ids.contiguous()])


def gather_topk(scores: torch.Tensor, ids: torch.Tensor, group=None
                ) -> Tuple[torch.Tensor, torch.Tensor]:
    """All-gather each rank's [nq, k] (scores f64, ids i64) -> [world, nq, k] on every rank.
    ONE collective per channel: the float64 scores travel as their int64 bit patterns next to
    the ids in a single [2, nq, k] int64 tile (the exchange is latency-bound, not
    bandwidth-bound).  The results are strided views of the gathered [world, 2, nq, k] tile,
    which thr_merge_topk reads in place.  Backend-agnostic (RCCL on GPUs; gloo in the CPU tests)."""
    world = dist.get_world_size(group)
    tile = _as_tile(scores, ids)
    out = torch.empty((world,) + tuple(tile.shape), dtype=torch.int64, device=tile.device)
    if dist.get_backend(group) == "gloo":
        # CPU rendezvous (tests / rehearsals): stage through host memory if the tiles are on a GPU
        host = [torch.empty(tile.shape, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(host, tile.cpu(), group=group)
        out = torch.stack(host).to(tile.device)
    else:
        dist.all_gather_into_tensor(out, tile, group=group)
    return out[:, 0].view(torch.float64), out[:, 1]


def gather_topk_many(pairs, group=None):
    """The same exchange for SEVERAL channels in ONE collective: ``pairs`` = [(scores [nq, k_c],
    ids [nq, k_c]), ...] -> [(scores [world, nq, k_c], ids [world, nq, k_c]), ...].  The channels'
    [2, nq, k_c] tiles travel back to back in one flat int64 buffer (one small packing copy: the
    exchange is latency-bound -- a collective per channel would pay its latency three times); the
    results are strided views of the gathered [world, total] buffer, which thr_merge_topk reads
    in place."""
    world = dist.get_world_size(group)
    tiles = [_as_tile(s, i).reshape(-1) for s, i in pairs]
    flat = tiles[0] if len(tiles) == 1 else torch.cat(tiles)
    if dist.get_backend(group) == "gloo":
        host = [torch.empty(flat.shape, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(host, flat.cpu(), group=group)
        out = torch.stack(host).to(flat.device)
    else:
        out = torch.empty((world, flat.numel()), dtype=torch.int64, device=flat.device)
        dist.all_gather_into_tensor(out, flat, group=group)
    res, off = [], 0
    for s, i in pairs:
        nq, k = i.shape
        t = out[:, off:off + 2 * nq * k].view(world, 2, nq, k)
        res.append((t[:, 0].view(torch.float64), t[:, 1]))
        off += 2 * nq * k
    return res


def gather_rows(x: torch.Tensor, group=None) -> torch.Tensor:
    """All-gather one fixed-shape tensor per rank -> [world, *x.shape] (the second exchange of
    the rerank leg: each rank's [nq, n] MaxSim scores, -inf where it does not own the doc)."""
    world = dist.get_world_size(group)
    x = x.contiguous()
    if dist.get_backend(group) == "gloo":
        host = [torch.empty(x.shape, dtype=x.dtype) for _ in range(world)]
        dist.all_gather(host, x.cpu(), group=group)
        return torch.stack(host).to(x.device)
    out = torch.empty((world,) + tuple(x.shape), dtype=x.dtype, device=x.device)
    dist.all_gather_into_tensor(out, x, group=group)
    return out


class ShardedIndex:
    """A GpuIndex holding this rank's document shard + the cross-rank merge."""

    def __init__(self, local: GpuIndex, group=None, floor: bool = True):
        self.local = local
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # the dense channel split around one more (small) exchange, so that a shard rescores its
        # share of the global top-k instead of a top-k of its own (GpuIndex.dense_search).  Every
        # rank must make the same collective calls, and which scan a shard runs depends on ITS rows
        # (set_dense falls back to the float32 scan for rows outside the float16 range): the
        # shards agree once, here, and the floor is used only if all of them can
        self.floor = floor
        if self.world > 1 and floor:
            ok = torch.tensor([1 if getattr(local, "shortlist", None) in ("f16", "f16-inline") else 0],
                              dtype=torch.int32,
                              device="cpu" if dist.get_backend(group) == "gloo" else local.device)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
            self.floor = bool(int(ok.item()))

    def _floor_exchange(self):
        if self.world == 1 or not self.floor or self.local.shortlist not in ("f16", "f16-inline"):
            return None
        if self.world > 4096:   # (the band kernel holds the shards' G * m bounds in 4096 LDS words)
            return None
        return (lambda lb: gather_rows(lb, self.group)), self.world

    def _merge(self, S, I, k):
        if self.world == 1:
            return S, I
        Sg, Ig = gather_topk(S, I, self.group)
        Sm, Im, _ = N.merge_topk(Sg, Ig, k)
        return Sm, Im

    def retrieve_batch(self, queries, query_terms=None, query_seeds=None, top_k: int = 10,
                       semantic_top_k: int = 100, lexical_top_k: int = 50, graph_top_k: int = 50,
                       weights: Optional[Dict[str, float]] = None, hops: int = 2,
                       qtok: Optional[torch.Tensor] = None, rerank_top_k: int = 100) -> BatchResult:
        """Per channel: this shard's exact top-k -> all-gather -> merge; fusion on every rank (the
        merged lists are identical everywhere); rerank: every rank scores the fused
        candidates it OWNS with thr_maxsim (-inf for the others), a second all-gather of the
        [nq, rerank_top_k] float32 scores, then thr_rerank_order takes the maximum per candidate
        and applies the reference's stable sort."""
        w = {"lexical": 0.7, "semantic": 0.8, "graph": 1.0}
        w.update(weights or {})
        L = self.local
        ch = {}
        # the lexical and graph kernels run on a side stream beside the dense channel; the
        # exchanges stay on the main stream, after the join
        lex, gra, join = L.side_channels(query_terms, lexical_top_k, query_seeds, graph_top_k, hops)
        Ss, Is, _, nres = L.dense_search(queries, semantic_top_k, sync=False,
                                         floor_exchange=self._floor_exchange())
        join()
        # ONE exchange for all the channels of the batch, then a merge per channel
        names, locals_, ks = ["semantic"], [(Ss, Is)], [semantic_top_k]
        if lex is not None:
            names.append("lexical"); locals_.append((lex[0], lex[1])); ks.append(lexical_top_k)
        if gra is not None:
            names.append("graph"); locals_.append((gra[0], gra[1])); ks.append(graph_top_k)
        if self.world > 1:
            merged = []
            for (Sg, Ig_), k in zip(gather_topk_many(locals_, self.group), ks):
                Sm, Im, _ = N.merge_topk(Sg, Ig_, k)
                merged.append((Sm, Im))
        else:
            merged = locals_
        for name, (Sm, Im) in zip(names, merged):
            ch[name] = (Sm, Im, None)
        Is = ch["semantic"][1]
        Il = ch["lexical"][1] if "lexical" in ch else None
        Ig = ch["graph"][1] if "graph" in ch else None
        rerank = qtok is not None and L.tokens is not None
        n_fused = max(rerank_top_k, top_k) if rerank else top_k
        ids, sc, _, cnt = N.rrf_fuse(Il, Is, Ig, n_fused, w["lexical"], w["semantic"], w["graph"])
        if rerank:
            ms = L.maxsim(qtok, ids)
            if self.world > 1:
                ms = gather_rows(ms, self.group)
            ids, sc, cnt = N.rerank_order(ms, ids, cnt, top_k)
        return BatchResult(ids, sc, cnt, ch, nres)
