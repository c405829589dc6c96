"""Row e2 on CPU: the drop-in surface over a DOCUMENT-SHARDED index -- control flow only.

Two gloo ranks; each holds one shard behind the GpuIndex search methods (the CPU oracle stands in
for the kernels, which need a GPU -- this is a test of sharded_client.py's request / broadcast /
all-gather / merge / serve-loop logic, not of the scorers).  Rank 0 runs ``RAG2Retriever.retrieve()``
and the tool layer over a ``ShardedIndexClient``; rank 1 sits in ``ShardWorker.serve()``.  The
contexts must equal those of the same retriever over ONE oracle-backed shard holding the whole
corpus."""
import asyncio
import json
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_DOCS, DIM = 1200, 64


class OracleShard:
    """The search methods of GpuIndex over rows [lo, hi) of the synthetic corpus, on the CPU."""

    def __init__(self, lo, hi, n, dim, collections=None):
        sys.path.insert(0, ROOT)
        from oracle import thr_oracle as O
        from triple_hybrid_rag_amd import synth
        self.O, self.lo, self.hi, self.dim = O, lo, hi, dim
        self.device = torch.device("cpu")
        self.doc_base = lo
        self.docs = synth.dense_rows(lo, hi - lo, dim)
        doc, term, tf = synth.lexical_rows(lo, hi - lo, n)
        self.csr = synth.build_lexical_csr(doc, term, tf, hi - lo, synth.vocab_size(n))
        df, sum_dl = synth.lexical_global_stats(n)       # global statistics, as every shard uses
        self.idf, self.avgdl = O.bm25_idf(n, df), sum_dl / n
        self.lex = True
        self.g = synth.build_graph(n, lo, hi)
        self.tokens = synth.doc_tokens(lo, hi - lo, 8, 16)
        self.doc_coll = None if collections is None else torch.from_numpy(collections[lo:hi].copy())
        self.calls = []

    @staticmethod
    def _tile(S, I, k):
        s = torch.full((1, k), -np.inf, dtype=torch.float64)
        i = torch.full((1, k), -1, dtype=torch.int64)
        s[0, :len(S[0])] = torch.from_numpy(np.asarray(S[0], dtype=np.float64))
        i[0, :len(I[0])] = torch.from_numpy(np.asarray(I[0], dtype=np.int64))
        return s, i

    def dense_search(self, q, k, collections=None, sync=True):
        self.calls.append("dense")
        docs = self.docs
        if collections is not None:   # WHERE collection = ..., before the ranking
            docs = docs.copy()
            docs[self.doc_coll.numpy() != int(collections[0])] = 0.0   # zero row = no embedding
        S, I = self.O.dense_topk_exact(docs, q.numpy(), k, doc_id_base=self.lo)
        return (*self._tile(S, I, k), None, 0)

    def bm25_search(self, qt, k, collections=None, conjunctive=False):
        self.calls.append("bm25")
        c = self.csr
        S, I = self.O.bm25_topk(c.rowptr, c.post_doc, c.post_tf, c.doclen, self.idf, self.avgdl,
                                qt.numpy(), self.hi - self.lo, k, doc_id_base=self.lo,
                                conjunctive=conjunctive,
                                doc_coll=None if collections is None else self.doc_coll.numpy(),
                                query_coll=None if collections is None else collections.numpy())
        return (*self._tile(S, I, k), None)

    def graph_search(self, seeds, k, hops):
        self.calls.append("graph")
        g = self.g
        S, I = self.O.graph_topk(g.ent_rowptr, g.ent_col, g.men_rowptr, g.men_chunk, g.men_conf,
                                 seeds.numpy(), hops, self.hi - self.lo, k, chunk_base=self.lo)
        return (*self._tile(S, I, k), None)

    def maxsim(self, qtok, cand):
        self.calls.append("maxsim")
        c = cand.numpy()
        local = np.where((c >= self.lo) & (c < self.hi), c - self.lo, -1)
        return torch.from_numpy(self.O.maxsim_scores(qtok.numpy(), self.tokens, local).astype(np.float32))


def cpu_merge(Sg, Ig, k):
    """(score desc, id asc) over the gathered [W, 1, k] lists -- what thr_merge_topk does on the GPU."""
    s, i = Sg[:, 0].reshape(-1).numpy(), Ig[:, 0].reshape(-1).numpy()
    keep = i >= 0
    s, i = s[keep], i[keep]
    order = np.lexsort((i, -s))[:k]
    S = torch.full((1, k), -np.inf, dtype=torch.float64)
    I = torch.full((1, k), -1, dtype=torch.int64)
    S[0, :len(order)] = torch.from_numpy(s[order])
    I[0, :len(order)] = torch.from_numpy(i[order])
    return S, I


class TokenEmbedder:
    def embed_query_tokens(self, text):
        rng = np.random.default_rng(sum(map(ord, text)))
        t = rng.standard_normal((4, 16)).astype(np.float32)
        return (t / np.linalg.norm(t, axis=1, keepdims=True)).astype(np.float16)


class Embedder:
    def __init__(self, n, dim):
        from triple_hybrid_rag_amd import synth
        self.q = synth.dense_queries(8, dim, n)

    def embed_query(self, text):
        return self.q[sum(map(ord, text)) % 8].tolist()


def _store(n):
    from triple_hybrid_rag_amd import synth
    from triple_hybrid_rag_amd.backend import CorpusStore
    st = CorpusStore.synthetic(n, vocab_size=synth.vocab_size(n), n_entities=synth.build_graph(n).ent_rowptr.shape[0] - 1)
    st.collections = [("manuals" if i % 3 == 0 else "faq") for i in range(n)]
    return st


def _collection_ids(st):
    return np.array([{"faq": 0, "manuals": 1}[c] for c in st.collections], dtype=np.int32)


QUERIES = [("t3 t17 t40", None), ("t5 t9", "manuals"), ("t2 t300 t8 zzz", "faq"), ("entity7 t11", None)]


def _run_queries(client, n, want_graph=True):
    """retrieve() + the tool layer + a rerank over ``client``; JSON-able summary."""
    from triple_hybrid_rag_amd.config import SETTINGS
    from triple_hybrid_rag_amd.rag2.query_planner import QueryPlan
    from triple_hybrid_rag_amd.rag2.retrieval import RAG2Retriever
    SETTINGS.rag2_safety_threshold, SETTINGS.rag2_denoise_alpha = 0.0, 0.0
    SETTINGS.rag2_graph_enabled = True

    class Planner:
        async def plan_async(self, query, collection=None):
            return QueryPlan(original_query=query, keywords=query.split(), semantic_query_text=query,
                             requires_graph=True, cypher_query="MATCH (e) RETURN e")

    out = []
    loop = asyncio.new_event_loop()
    for text, coll in QUERIES:
        r = RAG2Retriever(org_id="org", embedder=Embedder(n, DIM), query_planner=Planner(), graph_enabled=want_graph)
        r._supabase = client
        res = loop.run_until_complete(r.retrieve(text, collection=coll, top_k=7, skip_rerank=True))
        out.append([(c.child_id, c.lexical_rank, c.semantic_rank, c.graph_rank, c.rrf_score, c.parent_text)
                    for c in res.contexts])
    loop.close()
    out.append(client.maxsim_scores("some question", [f"c{i}" for i in (3, 700, 1199, 64, 601)]))
    out.append(client.rpc("rag2_lexical_search", {"p_org_id": "other-tenant", "p_query": "t3", "p_limit": 5}).execute().data)
    return out


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from triple_hybrid_rag_amd.distributed import shard_range
    from triple_hybrid_rag_amd.sharded_client import ShardedIndexClient, ShardWorker
    st = _store(N_DOCS)
    lo, hi = shard_range(N_DOCS, rank, world)
    shard = OracleShard(lo, hi, N_DOCS, DIM, _collection_ids(st))
    names = ["faq", "manuals"]
    if rank != 0:
        w = ShardWorker(shard, collection_names=names, merge_fn=cpu_merge)
        served = w.serve()
        with open(os.path.join(out_dir, f"served{rank}.json"), "w") as f:
            json.dump({"served": served, "calls": shard.calls}, f)
    else:
        with ShardedIndexClient(shard, st, collection_names=names, org_id="org",
                                token_embedder=TokenEmbedder(), merge_fn=cpu_merge) as client:
            got = _run_queries(client, N_DOCS)
            # a request the message buffer cannot hold fails on the front, before any collective
            with pytest.raises(ValueError):
                client._message(1, 10, 0, -1, np.zeros(1 << 20, dtype=np.float32))
        with open(os.path.join(out_dir, "got.json"), "w") as f:
            json.dump(got, f)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_client_two_ranks_equal_one_shard(tmp_path):
    world = 2
    port = 34100 + os.getpid() % 1500
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    got = json.load(open(tmp_path / "got.json"))
    served = json.load(open(tmp_path / "served1.json"))
    # the one-shard reference: the same retriever over a GpuIndexClient-shaped client on ONE
    # oracle shard that holds the whole corpus (a 1-rank group: same code path, no partner)
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port + 1))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        from triple_hybrid_rag_amd.sharded_client import ShardedIndexClient
        st = _store(N_DOCS)
        whole = OracleShard(0, N_DOCS, N_DOCS, DIM, _collection_ids(st))
        with ShardedIndexClient(whole, st, collection_names=["faq", "manuals"], org_id="org",
                                token_embedder=TokenEmbedder(), merge_fn=cpu_merge) as one:
            exp = _run_queries(one, N_DOCS)
    finally:
        dist.destroy_process_group()
    exp = json.loads(json.dumps(exp))
    assert len(got) == len(exp) == len(QUERIES) + 2
    for g, e in zip(got[:len(QUERIES)], exp[:len(QUERIES)]):
        assert g == e and len(g) > 0
    assert any(c[3] is not None for ctx in got[:len(QUERIES)] for c in ctx)      # the graph channel answered
    assert any(c[1] is not None and c[2] is not None for ctx in got[:len(QUERIES)] for c in ctx)
    assert got[len(QUERIES)] == pytest.approx(exp[len(QUERIES)], abs=1e-6) and got[-1] == []
    # the worker answered every request of the front and only those: one per RPC of every query
    assert served["served"] == len(served["calls"]) == len(whole.calls)
    assert set(served["calls"]) == {"dense", "bm25", "graph", "maxsim"}
