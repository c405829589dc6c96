#!/usr/bin/env python3
"""Generate golden vectors by RUNNING the reference's own Python.

Build-container only: imports /root/reference (never present on the GPU box)
through the stub modules in ``tests/golden/_stubs`` and
writes inputs + expected outputs as JSON next to this script.  Only data is
written -- no reference source text.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
"""
from __future__ import annotations

import asyncio
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path[:0] = [os.path.join(HERE, "_stubs"), os.path.join(REF, "src"),
                os.path.join(REF, "triple-hybrid-rag", "src")]

from voice_agent.config import SETTINGS  # noqa: E402
from voice_agent.rag2 import embedder as ref_embedder  # noqa: E402
from voice_agent.rag2 import retrieval as ref_retrieval  # noqa: E402
from voice_agent.rag2.query_planner import QueryPlan  # noqa: E402
from voice_agent.retrieval import hybrid_search as ref_hs  # noqa: E402
from voice_agent.retrieval import reranker as ref_rr  # noqa: E402
from triple_hybrid_rag.config import RAGConfig  # noqa: E402
from triple_hybrid_rag.core import fusion as ref_fusion  # noqa: E402
from triple_hybrid_rag import types as ref_types  # noqa: E402
from triple_hybrid_rag.core import embedder as ref_core_embedder  # noqa: E402
from voice_agent.tools import crm_knowledge as ref_crm  # noqa: E402


def dump(name, obj):
    path = os.path.join(HERE, name)
    with open(path, "w") as f:
        json.dump(obj, f, sort_keys=True, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes")


class _Exec:
    def __init__(self, data):
        self.data = data

    def execute(self):
        return self


class _Table:
    def __init__(self, rows):
        self.rows = rows

    def select(self, *_a, **_k):
        return self

    def in_(self, col, ids):
        ids = list(ids)
        # "DB order" = table order, not request order
        return _Exec([r for r in self.rows if r[col] in ids])


class FakeBackend:
    """Answers the four calls the retriever makes."""

    def __init__(self, lexical, semantic, children, parents):
        self.lexical, self.semantic = lexical, semantic
        self.children, self.parents = children, parents
        self.calls = []

    def rpc(self, name, params):
        self.calls.append((name, {k: v for k, v in params.items() if k != "p_embedding"}))
        rows = self.lexical if name == "rag2_lexical_search" else self.semantic
        return _Exec(rows[: params["p_limit"]])

    def table(self, name):
        return _Table(self.children if name == "rag_child_chunks" else self.parents)


class FakeEmbedder:
    def embed_query(self, text):
        return [0.0] * 8


def make_retriever(graph_enabled=False):
    return ref_retrieval.RAG2Retriever(org_id="org", embedder=FakeEmbedder(),
                                       query_planner=object(), graph_enabled=graph_enabled)


def cand_to_dict(c):
    return {k: getattr(c, k) for k in ("child_id", "parent_id", "document_id", "text", "page",
                                       "modality", "lexical_rank", "semantic_rank", "graph_rank",
                                       "rrf_score", "parent_text", "section_heading",
                                       "rerank_score")}


def gen_embed(rng):
    cases = []
    vecs = [[3.0, 4.0], [0.0, 0.0, 0.0], [1e-20, 0.0], [1.0], [-2.5, 0.5, 7.25, 1e-3]]
    for _ in range(6):
        vecs.append([rng.gauss(0, 1) for _ in range(rng.choice([5, 16, 33]))])
    for v in vecs:
        cases.append({"fn": "normalize_l2", "in": v, "out": ref_embedder.normalize_l2(v)})
    for v in vecs:
        for dim in (1, 4, 16, 64):
            for norm in (True, False):
                cases.append({"fn": "truncate_matryoshka", "in": v, "target_dim": dim,
                              "normalize": norm,
                              "out": ref_embedder.truncate_matryoshka(v, dim, norm)})
    return cases


def gen_rrf(rng):
    r = make_retriever()
    cases = []
    fixed = [(1, 1, 1), (1, 2, 3), (50, 100, 50), (None, 1, None), (7, None, 3), (None, None, None)]
    weight_sets = [{"lexical": 0.7, "semantic": 0.8, "graph": 1.0}, {},
                   {"lexical": 2.0, "semantic": 0.5, "graph": 0.0}, {"semantic": 0.8, "lexical": 0.7}]
    for w in weight_sets:
        for (l, s, g) in fixed:
            c = ref_retrieval.RetrievalCandidate("x", "p", "d", "t", 1, "text",
                                                 lexical_rank=l, semantic_rank=s, graph_rank=g)
            out = r._fuse_rrf([c], dict(w))
            cases.append({"weights": w, "ranks": [[l, s, g]], "ids": ["x"],
                          "scores": [out[0].rrf_score], "order": ["x"]})
    for _ in range(40):
        n = rng.randint(2, 60)
        w = rng.choice(weight_sets)
        ranks, ids, cs = [], [], []
        for i in range(n):
            l = rng.choice([None, rng.randint(1, 50)])
            s = rng.choice([None, rng.randint(1, 100)])
            g = rng.choice([None, None, rng.randint(1, 50)])
            ranks.append([l, s, g])
            ids.append(f"c{i}")
            cs.append(ref_retrieval.RetrievalCandidate(f"c{i}", "p", "d", "t", 1, "text",
                                                       lexical_rank=l, semantic_rank=s,
                                                       graph_rank=g))
        out = r._fuse_rrf(cs, dict(w))
        by_id = {c.child_id: c.rrf_score for c in out}
        cases.append({"weights": w, "ranks": ranks, "ids": ids,
                      "scores": [by_id[i] for i in ids], "order": [c.child_id for c in out]})
    return cases


def rand_rows(rng, n, pool):
    ids = [rng.choice(pool) for _ in range(n)] if rng.random() < 0.3 else rng.sample(pool, n)
    return [{"child_id": i, "parent_id": "p" + i[1:], "document_id": "d" + str(int(i[1:]) % 3),
             "text": "text of " + i, **({"page": rng.randint(1, 9)} if rng.random() < 0.7 else {}),
             **({"modality": rng.choice(["text", "table", "image"])} if rng.random() < 0.5 else {})}
            for i in ids]


def gen_merge(rng):
    cases = []
    for case in range(30):
        pool = [f"c{i}" for i in range(rng.randint(5, 120))]
        lex = rand_rows(rng, min(len(pool), rng.randint(0, 50)), pool)
        sem = rand_rows(rng, min(len(pool), rng.randint(0, 100)), pool)
        gra = rand_rows(rng, min(len(pool), rng.randint(0, 50)), pool)
        graph_on = rng.random() < 0.6
        keywords = ["kw"] if (lex and rng.random() < 0.9) else []
        r = make_retriever(graph_enabled=True)
        r.graph_enabled = graph_on

        async def lex_fn(keywords, collection, limit, _l=lex):
            return _l

        async def sem_fn(query_text, collection, limit, _s=sem):
            return _s

        async def gra_fn(cypher, keywords, collection, limit, _g=gra):
            return _g

        r._lexical_search, r._semantic_search, r._graph_search = lex_fn, sem_fn, gra_fn
        plan = QueryPlan(original_query="q", keywords=keywords, semantic_query_text="q",
                         cypher_query="MATCH (n) RETURN n" if graph_on else None,
                         requires_graph=graph_on)
        out = asyncio.run(r._retrieve_candidates(plan, None))
        cases.append({"lexical": lex if keywords else None, "semantic": sem,
                      "graph": gra if graph_on else None,
                      "out": [cand_to_dict(c) for c in out]})
    return cases


def gen_safety(rng):
    cases = []
    r = make_retriever()
    saved = (SETTINGS.rag2_safety_threshold, SETTINGS.rag2_denoise_alpha)
    settings = [(0.6, 0.6), (0.0, 0.0), (0.6, 0.5), (0.3, 0.9), (0.02, 0.6)]
    for thr, alpha in settings:
        SETTINGS.rag2_safety_threshold, SETTINGS.rag2_denoise_alpha = thr, alpha
        for _ in range(8):
            n = rng.randint(0, 12)
            cs = []
            for i in range(n):
                rs = rng.choice([None, 0.0, round(rng.random(), 3), 0.54, 0.9])
                cs.append(ref_retrieval.RetrievalCandidate(
                    f"c{i}", "p", "d", "t", 1, "text", rrf_score=rng.choice(
                        [0.03, 0.7, round(rng.random() * 0.05, 5), 0.539]), rerank_score=rs))
            top_k = rng.randint(1, 6)
            final, refused, reason, mx = r._apply_safety(cs, top_k)
            cases.append({"threshold": thr, "alpha": alpha, "top_k": top_k,
                          "cands": [{"child_id": c.child_id, "rrf_score": c.rrf_score,
                                     "rerank_score": c.rerank_score} for c in cs],
                          "final": [c.child_id for c in final], "refused": refused,
                          "reason": reason, "max_score": mx})
    SETTINGS.rag2_safety_threshold, SETTINGS.rag2_denoise_alpha = saved
    return cases


def gen_retrieve(rng):
    """End-to-end retrieve() traces over the fake backend (skip_planning and
    planner-stub variants, rerank skipped or native scores injected)."""
    cases = []
    saved = {k: getattr(SETTINGS, k) for k in ("rag2_safety_threshold", "rag2_denoise_alpha",
                                               "rag2_rerank_top_k", "rag2_rerank_enabled",
                                               "rag2_final_top_k", "rag2_graph_enabled")}
    for case in range(24):
        pool = [f"c{i}" for i in range(rng.randint(20, 200))]
        children = [{"id": i, "parent_id": "p" + str(int(i[1:]) // 4), "document_id": "d0",
                     "text": "child " + i, "page": int(i[1:]) % 7 + 1, "modality": "text"}
                    for i in pool]
        rng.shuffle(children)
        parents = [{"id": "p" + str(j), "text": "parent text " + str(j),
                    "section_heading": None if j % 3 == 0 else "Section " + str(j)}
                   for j in range(0, len(pool) // 4 + 1) if j % 5 != 4]
        row = lambda i: {"child_id": i, "parent_id": "p" + str(int(i[1:]) // 4),
                         "document_id": "d0", "text": "child " + i, "page": int(i[1:]) % 7 + 1,
                         "modality": "text"}
        lex = [row(i) for i in rng.sample(pool, min(len(pool), rng.randint(0, 50)))]
        sem = [row(i) for i in rng.sample(pool, min(len(pool), rng.randint(0, 100)))]
        gids = rng.sample(pool, min(len(pool), rng.randint(0, 60)))
        if case < 2:  # "No candidates found" path (retrieval.py:160-168)
            lex, sem, gids = [], [], []
        cfg = {"rag2_safety_threshold": rng.choice([0.0, 0.6, 0.01]),
               "rag2_denoise_alpha": rng.choice([0.0, 0.6, 0.9]),
               "rag2_rerank_top_k": rng.choice([20, 100, 5]),
               "rag2_rerank_enabled": True, "rag2_final_top_k": 5,
               "rag2_graph_enabled": True}
        for k, v in cfg.items():
            setattr(SETTINGS, k, v)
        use_graph = rng.random() < 0.5
        skip_planning = not use_graph and rng.random() < 0.5
        skip_rerank = rng.random() < 0.5
        top_k = rng.choice([None, 3, 10])
        query = rng.choice(["refund policy terms", "  ", "single", "a b c d e f"])
        weights = rng.choice([{"lexical": 0.7, "semantic": 0.8, "graph": 1.0},
                              {"lexical": 0.7, "semantic": 0.8}, {}])
        plan = QueryPlan(original_query=query, keywords=query.split(), semantic_query_text=query,
                         cypher_query="MATCH (e) RETURN e" if use_graph else None,
                         requires_graph=use_graph, weights=dict(weights))

        class Planner:
            async def plan_async(self, q, collection=None, _p=plan):
                return _p

        r = ref_retrieval.RAG2Retriever(org_id="org", embedder=FakeEmbedder(),
                                        query_planner=Planner(), graph_enabled=use_graph)
        be = FakeBackend(lex, sem, children, parents)
        r._supabase = be

        class GS:
            def __init__(self, ids):
                self.ids = ids

            async def search(self, keywords, cypher_query, org_id, top_k):
                class R:
                    chunk_ids = self.ids
                return R()

        import voice_agent.rag2.graph_search as gs_mod
        gs_mod.get_graph_searcher = lambda client, _g=gids: GS(_g)
        rerank_scores = [round(rng.random(), 4) if rng.random() < 0.9 else 0.0 for _ in range(200)]

        async def native(self, q, docs, _s=rerank_scores):
            return _s[: len(docs)]

        ref_rr.Qwen3VLReranker._rerank_batch_native = native
        res = asyncio.run(r.retrieve(query, collection=None, top_k=top_k,
                                     skip_planning=skip_planning, skip_rerank=skip_rerank))
        cases.append({
            "settings": cfg, "query": query, "top_k": top_k, "skip_planning": skip_planning,
            "skip_rerank": skip_rerank, "use_graph": use_graph, "weights": weights,
            "lexical": lex, "semantic": sem, "graph_chunk_ids": gids, "children": children,
            "parents": parents, "rerank_scores": rerank_scores,
            "out": {"success": res.success, "refused": res.refused,
                    "refusal_reason": res.refusal_reason, "max_rerank_score": res.max_rerank_score,
                    "contexts": [cand_to_dict(c) for c in res.contexts],
                    "timing_keys": sorted(res.timings),
                    "plan_keywords": res.query_plan.keywords if res.query_plan else None},
            "backend_calls": be.calls,
        })
    for k, v in saved.items():
        setattr(SETTINGS, k, v)
    return cases


def sr_legacy(d):
    return ref_hs.SearchResult(chunk_id=d["chunk_id"], content=d.get("content", ""),
                               modality="text", source_document="doc", page=1, chunk_index=0,
                               similarity_score=d.get("similarity_score", 0.0),
                               bm25_score=d.get("bm25_score", 0.0),
                               rrf_score=d.get("rrf_score", 0.0), is_table=d.get("is_table", False),
                               title=d.get("title"), table_context=d.get("table_context"),
                               alt_text=d.get("alt_text"))


def gen_legacy(rng):
    out = {"rrf": [], "lightweight": [], "qwen_rerank": []}
    searcher = ref_hs.HybridSearcher.__new__(ref_hs.HybridSearcher)
    searcher.config = ref_hs.SearchConfig()
    for _ in range(12):
        pool = [f"k{i}" for i in range(rng.randint(3, 40))]
        lists = []
        for _l in range(rng.randint(1, 3)):
            ids = rng.sample(pool, rng.randint(1, len(pool)))
            lists.append([{"chunk_id": i, "similarity_score": round(rng.random(), 4),
                           "bm25_score": round(rng.random() * 3, 4)} for i in ids])
        res = searcher._rrf_fusion([[sr_legacy(d) for d in l] for l in lists])
        out["rrf"].append({"lists": lists, "k": searcher.config.rrf_k,
                           "out": [{"chunk_id": r.chunk_id, "rrf_score": r.rrf_score,
                                    "similarity_score": r.similarity_score,
                                    "bm25_score": r.bm25_score,
                                    "retrieval_method": r.retrieval_method} for r in res]})
    words = ["refund", "policy", "table", "data", "dados", "price", "terms", "the", "of", "Tabela"]
    lw = ref_rr.LightweightReranker()
    for _ in range(12):
        n = rng.randint(0, 9)
        rows = [{"chunk_id": f"r{i}", "content": " ".join(rng.choice(words)
                                                          for _ in range(rng.randint(1, 8))),
                 "rrf_score": round(rng.random() * 0.05, 5),
                 "similarity_score": round(rng.random(), 4), "is_table": rng.random() < 0.3}
                for i in range(n)]
        query = " ".join(rng.choice(words) for _ in range(rng.randint(1, 4)))
        top_k = rng.choice([None, 2, 5])
        objs = [sr_legacy(d) for d in rows]
        res = asyncio.run(lw.rerank(query, objs, top_k))
        out["lightweight"].append({"query": query, "rows": rows, "top_k": top_k,
                                   "out": [{"chunk_id": r.chunk_id, "rerank_score": r.rerank_score}
                                           for r in res],
                                   "inplace_order": [r.chunk_id for r in objs]})
    # Qwen3VLReranker.rerank ordering/truncation with injected native scores
    for _ in range(10):
        n = rng.randint(0, 70)
        rows = [{"chunk_id": f"r{i}", "content": f"content {i}",
                 "title": "T" if rng.random() < 0.3 else None,
                 "is_table": rng.random() < 0.2, "table_context": "ctx",
                 "alt_text": "alt" if rng.random() < 0.2 else None} for i in range(n)]
        scores = [round(rng.random(), 3) for _ in range(n)]
        top_k = rng.choice([None, 3, 10, 60])
        rr = ref_rr.Qwen3VLReranker()
        enabled = rng.random() < 0.8
        rr.enabled, rr.use_local = enabled, True
        seen = {}

        async def native(self, q, docs, _s=scores, _seen=seen):
            _seen["docs"] = list(docs)
            return _s[: len(docs)]

        ref_rr.Qwen3VLReranker._rerank_batch_native = native
        objs = [sr_legacy(d) for d in rows]
        res = asyncio.run(rr.rerank("the query", objs, top_k))
        out["qwen_rerank"].append({"rows": rows, "scores": scores, "top_k": top_k,
                                   "enabled": enabled, "default_top_k": rr.top_k,
                                   "documents_sent": seen.get("docs"),
                                   "out": [{"chunk_id": r.chunk_id, "rerank_score": r.rerank_score}
                                           for r in res]})
    return out


def gen_standalone(rng):
    out = {"fuse": [], "two": [], "normalize": []}
    import uuid
    for _ in range(16):
        cfg = RAGConfig(rag_safety_threshold=rng.choice([0.0, 0.6, 0.3]),
                        rag_denoise_enabled=rng.random() < 0.7,
                        rag_denoise_alpha=rng.choice([0.6, 0.3, 0.9]))
        fus = ref_fusion.RRFFusion(cfg)
        pool = [str(uuid.UUID(int=rng.getrandbits(128))) for _ in range(rng.randint(3, 40))]

        def rows(field, n):
            ids = [rng.choice(pool) for _ in range(n)] if rng.random() < 0.25 else \
                rng.sample(pool, min(n, len(pool)))
            return [{"chunk_id": i, field: round(rng.random(), 4)} for i in ids]

        lex, sem, gra = rows("lexical_score", rng.randint(0, 20)), \
            rows("semantic_score", rng.randint(0, 30)), rows("graph_score", rng.randint(0, 10))
        mk = lambda d: ref_types.SearchResult(chunk_id=uuid.UUID(d["chunk_id"]),
                                              lexical_score=d.get("lexical_score", 0.0),
                                              semantic_score=d.get("semantic_score", 0.0),
                                              graph_score=d.get("graph_score", 0.0))
        weights = rng.choice([None, {"lexical": 0.7, "semantic": 0.8, "graph": 1.0},
                              {"lexical": 1.0, "semantic": 1.0}])
        plan = ref_types.QueryPlan(weights=dict(weights)) if weights is not None else None
        top_k = rng.choice([None, 5, 10])
        res = fus.fuse([mk(d) for d in lex], [mk(d) for d in sem], [mk(d) for d in gra],
                       query_plan=plan, top_k=top_k)
        out["fuse"].append({"safety_threshold": cfg.rag_safety_threshold,
                            "denoise_enabled": cfg.rag_denoise_enabled,
                            "denoise_alpha": cfg.rag_denoise_alpha,
                            "default_weights": fus.default_weights,
                            "lexical": lex, "semantic": sem, "graph": gra, "weights": weights,
                            "top_k": top_k,
                            "out": [{"chunk_id": str(r.chunk_id), "rrf_score": r.rrf_score,
                                     "lexical_score": r.lexical_score,
                                     "semantic_score": r.semantic_score,
                                     "graph_score": r.graph_score, "final_score": r.final_score,
                                     "source_channels": sorted(r.metadata["source_channels"])}
                                    for r in res]})
        a, b = rows("lexical_score", rng.randint(0, 15)), rows("semantic_score", rng.randint(0, 15))
        wa, wb = rng.choice([1.0, 0.7]), rng.choice([1.0, 0.8])
        res2 = fus.fuse_two_channels([mk(d) for d in a], [mk(d) for d in b], wa, wb, top_k)
        out["two"].append({"a": a, "b": b, "wa": wa, "wb": wb, "top_k": top_k,
                           "out": [{"chunk_id": str(r.chunk_id), "rrf_score": r.rrf_score}
                                   for r in res2]})
        vals = [round(rng.random() * 3, 4) for _ in range(rng.randint(0, 8))]
        if rng.random() < 0.3 and vals:
            vals = [vals[0]] * len(vals)
        objs = [ref_types.SearchResult(final_score=v) for v in vals]
        fus.normalize_scores(objs)
        out["normalize"].append({"in": vals, "out": [o.final_score for o in objs]})
    return out


def gen_planner(rng):
    """Rule-based planner of the standalone package (no GPT call)."""
    from triple_hybrid_rag.core.query_planner import QueryPlanner
    qp = QueryPlanner.__new__(QueryPlanner)
    qp.config = RAGConfig()
    queries = ["What is the refund policy?", "how do I reset my password", "Compare plan A and plan B",
               "Who works for Acme Corp?", "the of and", "Refund refund REFUND policy.",
               "difference between (premium) and [basic] tiers!", "define SLA", "",
               "which organization is related to Bob's company", "a b cd efg"]
    out = []
    for q in queries:
        plan = qp._simple_plan(q)
        out.append({"query": q, "keywords": plan.keywords, "requires_graph": plan.requires_graph,
                    "intent": plan.intent, "weights": plan.weights,
                    "semantic_query_text": plan.semantic_query_text,
                    "top_ks": [plan.lexical_top_k, plan.semantic_top_k, plan.graph_top_k],
                    "cypher_query": plan.cypher_query})
    return out


def b64(arr):
    """An array as data: little-endian bytes, base64 (tests/conftest.py decode_array)."""
    import base64
    import numpy as np
    a = np.ascontiguousarray(arr)
    return {"dtype": a.dtype.newbyteorder("<").str, "shape": list(a.shape),
            "b64": base64.b64encode(a.astype(a.dtype.newbyteorder("<")).tobytes()).decode()}


class _KBQuery:
    """knowledge_base_chunks query builder of the fallback scorer (hybrid_search.py:268-284)."""

    def __init__(self, rows, log):
        self.rows, self.log, self.not_ = rows, log, self

    def select(self, cols):
        self.log.append(["select", cols])
        return self

    def eq(self, col, val):
        self.log.append(["eq", col, val])
        self.rows = [r for r in self.rows if r.get(col) == val]
        return self

    def is_(self, col, val):          # reached through `.not_.is_(col, "null")`
        self.log.append(["not_is", col, val])
        self.rows = [r for r in self.rows if r.get(col) is not None]
        return self

    def limit(self, n):
        self.log.append(["limit", n])
        self.rows = self.rows[:n]
        return self

    def execute(self):
        return _Exec(self.rows)


class _KBClient:
    def __init__(self, rows):
        self.rows, self.log = rows, []

    def table(self, name):
        self.log.append(["table", name])
        return _KBQuery(list(self.rows), self.log)


def gen_dense():
    """Pins the dense score: the reference's two Python cosine scorers, run on
    float32-valued vectors.  (i) MultimodalEmbedder.cosine_similarity (core/embedder.py:316-331)
    for every (query, row) pair of a [24] x [120] set with zero, unnormalised and duplicate
    vectors; (ii) HybridSearcher._vector_search_fallback (hybrid_search.py:260-320) over a fake
    knowledge_base_chunks table holding the unit-normalised rows (NULL embedding for zero rows):
    returned ids, order and similarity scores."""
    import numpy as np
    nrng = np.random.default_rng(20260204)
    n, nq, d = 120, 24, 512
    rows = nrng.standard_normal((n, d)).astype(np.float32)
    rows *= nrng.choice([0.01, 1.0, 30.0], size=(n, 1)).astype(np.float32)
    rows[5] = 0
    rows[17] = 0
    rows[40:44] = rows[39]                      # exact duplicates: ties
    rows[60] = rows[59] * np.float32(2.0)       # same direction, other norm
    rows[70] = rows[69] + np.float32(1e-4) * nrng.standard_normal(d).astype(np.float32)
    q = nrng.standard_normal((nq, d)).astype(np.float32)
    q[0] = rows[39]
    q[1] = rows[59] * np.float32(0.25)
    q[2] = 0
    q[3] = rows[69]
    q[4:12] = rows[nrng.integers(0, n, 8)] + np.float32(0.5) * q[4:12]
    emb = ref_core_embedder.MultimodalEmbedder.__new__(ref_core_embedder.MultimodalEmbedder)
    cos = [[emb.cosine_similarity(q[i].tolist(), rows[j].tolist()) for j in range(n)]
           for i in range(nq)]
    # (ii) the table stores what a1 produces: float32 L2-normalised rows (rag2/embedder.py:31-37)
    unit = np.array([ref_embedder.normalize_l2(r.tolist()) for r in rows], dtype=np.float32)
    qunit = np.array([ref_embedder.normalize_l2(v.tolist()) for v in q], dtype=np.float32)
    table = [{"id": f"c{j}", "org_id": "org", "content": f"text {j}", "modality": "text",
              "source_document": "doc", "page": 1 + j % 7, "chunk_index": j,
              "category": "faq" if j % 3 == 0 else "pricing",
              "vector_embedding": unit[j].tolist() if rows[j].any() else None}
             for j in range(n)]
    searcher = ref_hs.HybridSearcher.__new__(ref_hs.HybridSearcher)
    searcher.org_id = "org"
    searcher.config = ref_hs.SearchConfig(top_k_retrieve=60)
    fallback = []
    for i in range(nq):
        for category in (None, "faq") if i < 6 else (None,):
            client = _KBClient(table)
            searcher._supabase = client
            res = asyncio.run(searcher._vector_search_fallback(qunit[i].tolist(), category))
            fallback.append({"query": i, "category": category, "top_k_retrieve": 60,
                             "calls": client.log,
                             "out": [{"chunk_id": r.chunk_id, "similarity_score": r.similarity_score,
                                      "retrieval_method": r.retrieval_method} for r in res]})
    return {"dim": d, "rows": b64(rows), "queries": b64(q), "cosine_similarity": cos,
            "unit_rows": b64(unit), "unit_queries": b64(qunit),
            "null_rows": [j for j in range(n) if not rows[j].any()],
            "categories": [t["category"] for t in table], "fallback": fallback}


def gen_tool(rng):
    """_search_knowledge_base_rag2 (tools/crm_knowledge.py:69-182) over a fake retriever: the dict
    schema, rounding, falsy-score handling, refusal mapping and millisecond timings."""
    C = ref_retrieval.RetrievalCandidate
    cases = []
    fixed_score = [None, 0.0, 0.85, 0.01639344262295082, 0.92, 0.00004, 1.0]
    for ci in range(14):
        refused = ci in (3, 9)
        n = 0 if ci == 5 else rng.randint(1, 6)
        ctxs = []
        for j in range(n):
            ctxs.append(dict(
                child_id=f"child-{ci}-{j}", parent_id=f"parent-{ci}-{j // 2}",
                document_id=f"doc-{ci}", text=f"child text {j}", page=rng.randint(1, 30),
                modality=rng.choice(["text", "table", "image"]),
                lexical_rank=rng.choice([None, 1, 2, 17]), semantic_rank=rng.choice([None, 1, 3, 99]),
                graph_rank=rng.choice([None, 1, 4]),
                rrf_score=rng.choice(fixed_score[1:] + [round(rng.random() * 0.05, 6)]),
                parent_text=rng.choice([None, "", f"parent text {j}"]),
                section_heading=rng.choice([None, "", "Section Title"]),
                rerank_score=rng.choice(fixed_score + [rng.random()])))
        res = dict(success=True, contexts=ctxs,
                   max_rerank_score=rng.choice([0.0, 0.92, rng.random()]),
                   refused=refused,
                   refusal_reason="Max score 0.01 below threshold 0.6" if refused else None,
                   timings={"planning": rng.random() * 0.2, "retrieval": rng.random(),
                            "fusion": 1.23456e-4, "total": 0.33335})
        call = dict(query=rng.choice(["What is the policy?", "preço do plano", ""]),
                    category=rng.choice([None, "faq", "pricing"]), limit=rng.choice([1, 5, 10]),
                    org_id=rng.choice([None, "org-explicit"]))
        seen = {}

        class FakeRetriever:
            def __init__(self, org_id, graph_enabled=False, _seen=seen):
                _seen["init"] = {"org_id": org_id, "graph_enabled": graph_enabled}

            async def retrieve(self, query, collection=None, top_k=None, _seen=seen, _res=res):
                _seen["retrieve"] = {"query": query, "collection": collection, "top_k": top_k}
                return ref_retrieval.RetrievalResult(
                    success=_res["success"], contexts=[C(**c) for c in _res["contexts"]],
                    max_rerank_score=_res["max_rerank_score"], refused=_res["refused"],
                    refusal_reason=_res["refusal_reason"], timings=dict(_res["timings"]))

        class FakeDB:
            def table(self, name, _seen=seen):
                _seen.setdefault("tables", []).append(name)
                return self

            def select(self, *_a):
                return self

            def limit(self, _n):
                return self

            def execute(self):
                return _Exec([{"org_id": "org-123", "id": "org-first"}])

        keep = (ref_retrieval.RAG2Retriever, ref_crm.get_supabase_client,
                SETTINGS.rag2_graph_enabled)
        ref_retrieval.RAG2Retriever = FakeRetriever
        ref_crm.get_supabase_client = lambda: FakeDB()
        graph_flag = ci % 2 == 0
        object.__setattr__(SETTINGS, "rag2_graph_enabled", graph_flag)
        try:
            out = ref_crm._search_knowledge_base_rag2(call["query"], call["category"],
                                                      call["limit"], call["org_id"])
        finally:
            ref_retrieval.RAG2Retriever, ref_crm.get_supabase_client = keep[0], keep[1]
            object.__setattr__(SETTINGS, "rag2_graph_enabled", keep[2])
        cases.append({"call": call, "result": res, "rag2_graph_enabled": graph_flag,
                      "seen": seen, "out": out})
    return cases


def main():
    rng = random.Random(20260130)
    dump("embed_postproc.json", gen_embed(rng))
    dump("rrf_fuse.json", gen_rrf(rng))
    dump("merge_candidates.json", gen_merge(rng))
    dump("safety.json", gen_safety(rng))
    dump("retrieve_traces.json", gen_retrieve(rng))
    dump("legacy_rerank.json", gen_legacy(rng))
    dump("standalone_fusion.json", gen_standalone(rng))
    dump("simple_planner.json", gen_planner(rng))
    dump("dense_cosine.json", gen_dense())
    dump("tool_layer.json", gen_tool(random.Random(20260205)))
    dump("defaults.json", {
        "settings": {k: getattr(SETTINGS, k) for k in (
            "rag2_enabled", "rag2_graph_enabled", "rag2_rerank_enabled", "rag2_denoise_enabled",
            "rag2_embed_dim_store", "rag2_embed_dim_model", "rag2_safety_threshold",
            "rag2_denoise_alpha", "rag2_lexical_weight", "rag2_semantic_weight",
            "rag2_graph_weight", "rag2_lexical_top_k", "rag2_semantic_top_k", "rag2_graph_top_k",
            "rag2_rerank_top_k", "rag2_final_top_k")},
        "query_plan": QueryPlan(original_query="q").__dict__,
    })


if __name__ == "__main__":
    main()
