"""The product's Python surface (retrieve()/rerank()/fusion mirrors) replayed against the
golden vectors the reference's own code produced (tests/golden/*.json).  CPU only: the
channel seams are faked exactly as the reference's tests fake them
(tests/test_rag2_triple_hybrid.py:44-69), so no kernel runs here."""
import asyncio
import copy
import uuid

import pytest

import triple_hybrid_rag_amd as T
from triple_hybrid_rag_amd.config import SETTINGS, RAGConfig, Settings
from triple_hybrid_rag_amd.core.fusion import RRFFusion
from triple_hybrid_rag_amd.core.types import QueryPlan as SQueryPlan
from triple_hybrid_rag_amd.core.types import SearchResult as SSearchResult
from triple_hybrid_rag_amd.rag2 import embedder as E
from triple_hybrid_rag_amd.rag2 import graph_search as GS
from triple_hybrid_rag_amd.rag2.query_planner import QueryPlan
from triple_hybrid_rag_amd.rag2.retrieval import RAG2Retriever, RetrievalCandidate
from triple_hybrid_rag_amd.retrieval import reranker as RR
from triple_hybrid_rag_amd.retrieval.hybrid_search import SearchResult, rrf_fusion

CAND_KEYS = ("child_id", "parent_id", "document_id", "text", "page", "modality", "lexical_rank",
             "semantic_rank", "graph_rank", "rrf_score", "parent_text", "section_heading",
             "rerank_score")


class _Reply:
    def __init__(self, data):
        self.data = data

    def execute(self):
        return self


class _Table:
    def __init__(self, rows):
        self.rows = rows

    def select(self, *_a):
        return self

    def in_(self, col, ids):
        ids = list(ids)
        return _Reply([r for r in self.rows if r[col] in ids])


class FakeBackend:
    def __init__(self, c):
        self.c, self.calls = c, []

    def rpc(self, name, params):
        self.calls.append([name, {k: v for k, v in params.items() if k != "p_embedding"}])
        rows = self.c["lexical"] if name == "rag2_lexical_search" else self.c["semantic"]
        return _Reply(rows[: params["p_limit"]])

    def table(self, name):
        return _Table(self.c["children"] if name == "rag_child_chunks" else self.c["parents"])


class FakeEmbedder:
    def embed_query(self, text):
        return [0.0] * 8


@pytest.fixture
def settings_guard():
    saved = copy.deepcopy(SETTINGS.__dict__)
    yield
    SETTINGS.__dict__.update(saved)


def test_defaults_match_reference(golden):
    d = golden("defaults.json")
    fresh = Settings()
    for k, v in d["settings"].items():
        assert getattr(fresh, k) == v, k
    assert QueryPlan(original_query="q").__dict__ == d["query_plan"]


def test_embed_postproc_host_functions(golden):
    for c in golden("embed_postproc.json"):
        if c["fn"] == "normalize_l2":
            assert E.normalize_l2(c["in"]) == c["out"]
        else:
            assert E.truncate_matryoshka(c["in"], c["target_dim"], c["normalize"]) == c["out"]


def test_fuse_rrf_and_merge_and_safety(golden, settings_guard):
    r = RAG2Retriever(org_id="t", embedder=FakeEmbedder(), query_planner=object())
    for c in golden("rrf_fuse.json"):
        cands = [RetrievalCandidate(i, "p", "d", "t", 1, "text", lexical_rank=k[0],
                                    semantic_rank=k[1], graph_rank=k[2])
                 for i, k in zip(c["ids"], c["ranks"])]
        out = r._fuse_rrf(cands, dict(c["weights"]))
        assert [x.child_id for x in out] == c["order"]
        by = {x.child_id: x.rrf_score for x in out}
        assert [by[i] for i in c["ids"]] == c["scores"]
    for c in golden("merge_candidates.json"):
        rr = RAG2Retriever(org_id="t", embedder=FakeEmbedder(), query_planner=object())
        rr.graph_enabled = c["graph"] is not None

        async def lex(keywords, collection, limit, _r=c["lexical"]):
            return _r

        async def sem(query_text, collection, limit, _r=c["semantic"]):
            return _r

        async def gra(cypher, keywords, collection, limit, _r=c["graph"]):
            return _r

        rr._lexical_search, rr._semantic_search, rr._graph_search = lex, sem, gra
        plan = QueryPlan(original_query="q", keywords=["kw"] if c["lexical"] is not None else [],
                         semantic_query_text="q", cypher_query="MATCH" if c["graph"] is not None else None,
                         requires_graph=c["graph"] is not None)
        out = asyncio.run(rr._retrieve_candidates(plan, None))
        keys = CAND_KEYS[:9]
        assert [{k: getattr(x, k) for k in keys} for x in out] == [{k: e[k] for k in keys} for e in c["out"]]
    for c in golden("safety.json"):
        SETTINGS.rag2_safety_threshold, SETTINGS.rag2_denoise_alpha = c["threshold"], c["alpha"]
        cands = [RetrievalCandidate(d["child_id"], "p", "d", "t", 1, "text", rrf_score=d["rrf_score"],
                                    rerank_score=d["rerank_score"]) for d in c["cands"]]
        final, refused, reason, mx = r._apply_safety(cands, c["top_k"])
        assert [x.child_id for x in final] == c["final"]
        assert (refused, reason, mx) == (c["refused"], c["reason"], c["max_score"])


def test_retrieve_traces_match_reference(golden, settings_guard, monkeypatch):
    for c in golden("retrieve_traces.json"):
        for k, v in c["settings"].items():
            setattr(SETTINGS, k, v)
        plan = QueryPlan(original_query=c["query"], keywords=c["query"].split(),
                         semantic_query_text=c["query"],
                         cypher_query="MATCH (e) RETURN e" if c["use_graph"] else None,
                         requires_graph=c["use_graph"], weights=dict(c["weights"]))

        class Planner:
            async def plan_async(self, q, collection=None, _p=plan):
                return _p

        r = RAG2Retriever(org_id="org", embedder=FakeEmbedder(), query_planner=Planner(),
                          graph_enabled=c["use_graph"])
        be = FakeBackend(c)
        r._supabase = be

        class Searcher:
            async def search(self, keywords, cypher_query, org_id, top_k, _ids=c["graph_chunk_ids"]):
                return GS.GraphSearchResult([], [], [], list(_ids), "test")

        monkeypatch.setattr(GS, "get_graph_searcher", lambda client: Searcher())

        async def native(self, q, docs, _s=c["rerank_scores"]):
            return _s[: len(docs)]

        monkeypatch.setattr(RR.Qwen3VLReranker, "_rerank_batch_native", native)
        monkeypatch.setattr(RR.Qwen3VLReranker, "bind_candidates", lambda self, ids, cl=None: None)
        res = asyncio.run(r.retrieve(c["query"], collection=None, top_k=c["top_k"],
                                     skip_planning=c["skip_planning"], skip_rerank=c["skip_rerank"]))
        o = c["out"]
        assert (res.success, res.refused, res.refusal_reason, res.max_rerank_score) == \
            (o["success"], o["refused"], o["refusal_reason"], o["max_rerank_score"])
        assert [{k: getattr(x, k) for k in CAND_KEYS} for x in res.contexts] == o["contexts"]
        assert sorted(res.timings) == o["timing_keys"]
        assert res.query_plan.keywords == o["plan_keywords"]
        assert be.calls == c["backend_calls"]


def test_module_level_retrieve_builds_a_fresh_retriever(monkeypatch):
    # reference tests/test_rag2_triple_hybrid.py:910-918
    made = []

    class Probe(RAG2Retriever):
        def __init__(self, org_id, **kw):
            made.append(org_id)

        async def retrieve(self, query, **kw):
            return ("ok", query, kw)

    import triple_hybrid_rag_amd.rag2.retrieval as mod
    monkeypatch.setattr(mod, "RAG2Retriever", Probe)
    assert asyncio.run(mod.retrieve("org9", "q", top_k=3)) == ("ok", "q", {"top_k": 3})
    assert made == ["org9"]


def test_graph_failure_degrades_to_empty_channel(monkeypatch):
    r = RAG2Retriever(org_id="o", embedder=FakeEmbedder(), query_planner=object())
    r._supabase = object()

    def boom(client):
        raise RuntimeError("puppygraph down")

    monkeypatch.setattr(GS, "get_graph_searcher", boom)
    assert asyncio.run(r._graph_search("MATCH", ["a"], None, 5)) == []


def test_embed_failure_raises_value_error():
    emb = E.PrecomputedEmbedder(store_dim=4)
    emb.register("known", [3.0, 4.0, 0.0, 0.0, 9.0])
    assert emb.embed_query("known") == [0.6000000238418579, 0.800000011920929, 0.0, 0.0]
    with pytest.raises(ValueError):
        emb.embed_query("unknown")


def legacy(d):
    return SearchResult(chunk_id=d["chunk_id"], content=d.get("content", ""), modality="text",
                        source_document="doc", page=1, chunk_index=0,
                        similarity_score=d.get("similarity_score", 0.0),
                        bm25_score=d.get("bm25_score", 0.0), rrf_score=d.get("rrf_score", 0.0),
                        is_table=d.get("is_table", False), title=d.get("title"),
                        table_context=d.get("table_context"), alt_text=d.get("alt_text"))


def test_legacy_rrf_and_rerankers(golden, monkeypatch):
    g = golden("legacy_rerank.json")
    for c in g["rrf"]:
        out = rrf_fusion([[legacy(d) for d in l] for l in c["lists"]], c["k"])
        assert [{"chunk_id": r.chunk_id, "rrf_score": r.rrf_score,
                 "similarity_score": r.similarity_score, "bm25_score": r.bm25_score,
                 "retrieval_method": r.retrieval_method} for r in out] == c["out"]
    lw = RR.LightweightReranker()
    for c in g["lightweight"]:
        objs = [legacy(d) for d in c["rows"]]
        out = asyncio.run(lw.rerank(c["query"], objs, c["top_k"]))
        assert [{"chunk_id": r.chunk_id, "rerank_score": r.rerank_score} for r in out] == c["out"]
        assert [r.chunk_id for r in objs] == c["inplace_order"]
    for c in g["qwen_rerank"]:
        rr = RR.Qwen3VLReranker(client=object())
        rr.enabled, rr.use_local = c["enabled"], True
        assert rr.top_k == c["default_top_k"]
        seen = {}

        async def native(self, q, docs, _s=c["scores"], _seen=seen):
            _seen["docs"] = list(docs)
            return _s[: len(docs)]

        monkeypatch.setattr(RR.Qwen3VLReranker, "_rerank_batch_native", native)
        out = asyncio.run(rr.rerank("the query", [legacy(d) for d in c["rows"]], c["top_k"]))
        assert [{"chunk_id": r.chunk_id, "rerank_score": r.rerank_score} for r in out] == c["out"]
        assert seen.get("docs") == c["documents_sent"]


def test_reranker_failure_keeps_order():
    class Broken:
        def maxsim_scores(self, q, ids):
            raise RuntimeError("no GPU")

        def text_to_child_id(self, t):
            raise RuntimeError("no GPU")

    rr = RR.Qwen3VLReranker(client=Broken())
    rows = [legacy({"chunk_id": f"r{i}", "content": f"c{i}"}) for i in range(7)]
    out = asyncio.run(rr.rerank("q", rows, 3))
    # native fails -> per-pair fallback answers 0.5 for everything -> stable sort keeps order
    assert [r.chunk_id for r in out] == ["r0", "r1", "r2"]
    assert RR.get_reranker(use_local=False).__class__ is RR.LightweightReranker
    assert RR.get_reranker(use_local=True, client=Broken()).__class__ is RR.Qwen3VLReranker


def test_standalone_fusion(golden):
    g = golden("standalone_fusion.json")
    mk = lambda d: SSearchResult(chunk_id=uuid.UUID(d["chunk_id"]),
                                 lexical_score=d.get("lexical_score", 0.0),
                                 semantic_score=d.get("semantic_score", 0.0),
                                 graph_score=d.get("graph_score", 0.0))
    for c in g["fuse"]:
        fus = RRFFusion(RAGConfig(rag_safety_threshold=c["safety_threshold"],
                                  rag_denoise_enabled=c["denoise_enabled"],
                                  rag_denoise_alpha=c["denoise_alpha"]))
        assert fus.default_weights == c["default_weights"]
        plan = SQueryPlan(weights=dict(c["weights"])) if c["weights"] is not None else None
        out = fus.fuse([mk(d) for d in c["lexical"]], [mk(d) for d in c["semantic"]],
                       [mk(d) for d in c["graph"]], query_plan=plan, top_k=c["top_k"])
        got = [{"chunk_id": str(r.chunk_id), "rrf_score": r.rrf_score,
                "lexical_score": r.lexical_score, "semantic_score": r.semantic_score,
                "graph_score": r.graph_score, "final_score": r.final_score,
                "source_channels": sorted(r.metadata["source_channels"])} for r in out]
        assert got == c["out"]
    fus = RRFFusion(RAGConfig())
    for c in g["two"]:
        out = fus.fuse_two_channels([mk(d) for d in c["a"]], [mk(d) for d in c["b"]], c["wa"],
                                    c["wb"], c["top_k"])
        assert [{"chunk_id": str(r.chunk_id), "rrf_score": r.rrf_score} for r in out] == c["out"]
    for c in g["normalize"]:
        objs = [SSearchResult(final_score=v) for v in c["in"]]
        fus.normalize_scores(objs)
        assert [o.final_score for o in objs] == c["out"]


def test_rule_based_planner_matches_reference(golden):
    from triple_hybrid_rag_amd.core.query_planner import QueryPlanner
    qp = QueryPlanner(RAGConfig())
    for c in golden("simple_planner.json"):
        plan = qp._simple_plan(c["query"])
        assert plan.keywords == c["keywords"], c["query"]
        assert (plan.requires_graph, plan.intent, plan.weights, plan.semantic_query_text,
                plan.cypher_query) == (c["requires_graph"], c["intent"], c["weights"],
                                       c["semantic_query_text"], c["cypher_query"])
        assert [plan.lexical_top_k, plan.semantic_top_k, plan.graph_top_k] == c["top_ks"]
    assert asyncio.run(qp.plan_async("Who works for Acme?")).requires_graph


def test_legacy_hybrid_searcher_flow():
    from triple_hybrid_rag_amd.retrieval.hybrid_search import HybridSearcher, SearchConfig

    def row(i, **kw):
        return {"id": f"k{i}", "content": f"c{i}", "modality": "text", "source_document": "docA" if i % 2 else "docB",
                "page": None, "chunk_index": None, **kw}

    class Backend:
        def __init__(self):
            self.calls = []

        def rpc(self, name, params):
            self.calls.append((name, params.get("p_limit")))
            if name == "kb_chunks_vector_search":
                rows = [row(i, similarity=1 - i / 10) for i in (1, 2, 3)]
            else:
                rows = [row(i, rank=5.0 - i) for i in (3, 1, 4)]
            return _Reply(rows)

    class Emb:
        async def embed_query(self, q):
            return [0.1, 0.2], None

    hs = HybridSearcher("org", embedder=Emb(), config=SearchConfig(top_k_retrieve=7))
    hs._supabase = be = Backend()
    out = asyncio.run(hs.search("q", top_k=3))
    assert be.calls == [("kb_chunks_vector_search", 7), ("kb_chunks_fts_pt", 7)]
    # unweighted RRF, 0-based ranks: k1 = 1/61 + 1/62, k3 = 1/63 + 1/61; first-seen order on ties
    assert [r.chunk_id for r in out] == ["k1", "k3", "k2"]
    assert out[0].rrf_score == 1 / 61 + 1 / 62 and out[0].retrieval_method == "hybrid"
    assert out[1].similarity_score == 0.7 and out[1].bm25_score == 2.0 and out[0].page == 1
    only = asyncio.run(HybridSearcher("org", Emb(), SearchConfig(use_bm25=False))._with(be).search("q", source_document="docA"))
    assert [r.chunk_id for r in only] == ["k1", "k3"] and only[0].retrieval_method == "vector"


def test_tool_layer_matches_reference(golden, settings_guard, monkeypatch):
    """8f.2: _search_knowledge_base_rag2 against the reference's own function run over a fake
    retriever (tests/golden/tool_layer.json <- tools/crm_knowledge.py:69-182): constructor and
    retrieve() arguments, tenant discovery, dict schema, rounding of falsy scores, refusal
    mapping, millisecond timings.  Known answers of tests/test_rag2_tool_connection.py:78-330
    (chunk_id / parent_id / rounding / refusal keys) are covered by the same cases."""
    from triple_hybrid_rag_amd.rag2.retrieval import RetrievalResult
    from triple_hybrid_rag_amd.tools import crm_knowledge as crm
    for c in golden("tool_layer.json"):
        seen = {}

        class FakeRetriever:
            def __init__(self, org_id, graph_enabled=False):
                seen["init"] = {"org_id": org_id, "graph_enabled": graph_enabled}

            async def retrieve(self, query, collection=None, top_k=None, _r=c["result"]):
                seen["retrieve"] = {"query": query, "collection": collection, "top_k": top_k}
                return RetrievalResult(success=_r["success"],
                                       contexts=[RetrievalCandidate(**x) for x in _r["contexts"]],
                                       max_rerank_score=_r["max_rerank_score"], refused=_r["refused"],
                                       refusal_reason=_r["refusal_reason"], timings=dict(_r["timings"]))

        class FakeDB:
            def table(self, name):
                seen.setdefault("tables", []).append(name)
                return self

            def select(self, *_a):
                return self

            def limit(self, _n):
                return self

            def execute(self):
                return _Reply([{"org_id": "org-123", "id": "org-first"}])

        monkeypatch.setattr(crm, "RAG2Retriever", FakeRetriever)
        monkeypatch.setattr(crm, "get_supabase_client", lambda: FakeDB())
        SETTINGS.rag2_graph_enabled = c["rag2_graph_enabled"]
        call = c["call"]
        out = crm._search_knowledge_base_rag2(call["query"], call["category"], call["limit"],
                                              call["org_id"])
        assert out == c["out"]
        assert seen == c["seen"]
    # the dispatcher: RAG 2.0 when enabled, never raises into the agent
    SETTINGS.rag2_enabled = True
    c = golden("tool_layer.json")[0]
    assert crm.search_knowledge_base(c["call"]["query"], c["call"]["category"],
                                     c["call"]["limit"])["search_type"] == "rag2_triple_hybrid"
    monkeypatch.setattr(crm, "get_supabase_client", lambda: (_ for _ in ()).throw(RuntimeError("db down")))
    err = crm.search_knowledge_base("q", "faq")
    assert err == {"error": "Database error: db down", "query": "q", "category": "faq"}


def test_entity_lookup_index_equals_the_plain_scan():
    """GpuIndexClient.find_entities (graph_search.py:161-170: name ILIKE %kw%, first 5 keywords,
    limit // len(keywords) entities each, ascending entity order) through the trigram index must
    return what the O(E) scan over the names returns -- accents, case, keywords shorter than a
    trigram, keywords nobody matches, duplicates across keywords."""
    from triple_hybrid_rag_amd import _native as N
    from triple_hybrid_rag_amd.backend import CorpusStore, GpuIndexClient
    names = [f"entity{e}" for e in range(3000)] + ["Ação Ltda", "açúcar união", "Acme Corp", "acme holdings",
                                                   "xy", "x", "ACME", "São Paulo S.A.", ""]
    store = CorpusStore.synthetic(8)
    store.entity_names = names

    class Client(GpuIndexClient):
        def __init__(self):
            self.store = store

    client = Client()

    def scan(keywords, limit):
        per, found = max(1, limit // len(keywords)), []
        for kw in keywords[:5]:
            needle, hits = kw.lower(), 0
            for e, name in enumerate(names):
                if needle in name.lower():
                    if e not in found:
                        found.append(e)
                    hits += 1
                    if hits == per:
                        break
        return found[: N.THR_GRAPH_MAX_SEEDS]

    for kws in (["entity12"], ["entity4", "acme"], ["AÇÃO", "união", "zzz"], ["xy", "x", "ent"], ["São", "s.a."],
                ["tity299", "e", "ac", "corp", "hold", "ignored"], ["entity"], ["ACME", "acme", "Acme"]):
        for limit in (20, 50, 3, 1):
            assert client.find_entities(kws, limit) == scan(kws, limit), (kws, limit)
    assert client.find_entities([], 20) == []


def test_lazy_rows_and_the_layout_rule():
    from triple_hybrid_rag_amd.backend import LazyRows
    from triple_hybrid_rag_amd.distributed import auto_doc_shards, layout_2d
    calls = []
    rows = LazyRows(lambda: calls.append(1) or [{"child_id": "a"}, {"child_id": "b"}])
    assert calls == []                       # nothing is read back before the rows are looked at
    assert [r["child_id"] for r in rows] == ["a", "b"] and len(rows) == 2 and rows and calls == [1]
    assert rows[1]["child_id"] == "b" and rows == [{"child_id": "a"}, {"child_id": "b"}] and calls == [1]
    assert rows + [1] == [{"child_id": "a"}, {"child_id": "b"}, 1] and {"child_id": "b"} in rows
    assert list(reversed(rows))[0]["child_id"] == "b" and not isinstance(rows, list)
    import json
    assert json.loads(json.dumps(rows.materialize())) == [{"child_id": "a"}, {"child_id": "b"}]
    # a failing deferred part RAISES at the first look and at every later one (the reference's
    # _lexical_search has no try/except, retrieval.py:273-292): a GPU fault is never an empty channel
    boom = []
    broken = LazyRows(lambda: boom.append(1) or 1 / 0)
    for _ in range(2):
        with pytest.raises(ZeroDivisionError):
            list(broken)
        with pytest.raises(ZeroDivisionError):
            len(broken)
    assert boom == [1]                       # the fetch itself ran once
    # one GPU: no split; the 1M-doc headline corpus: 2 shards x N/2 replicas; 10M docs: N shards
    assert [auto_doc_shards(w, 1_000_000) for w in (1, 2, 4, 8)] == [1, 2, 2, 2]
    assert [auto_doc_shards(w, 10_000_000) for w in (1, 2, 4, 8)] == [1, 2, 4, 8]
    assert [auto_doc_shards(w, 40_000) for w in (2, 8)] == [2, 2]      # the exchange stays in the path
    assert auto_doc_shards(6, 2_000_000) == 3 and auto_doc_shards(8, 2_000_000, 250_000) == 8
    assert layout_2d(5, 8, 2) == (1, 2, [4, 5])


def test_synthetic_global_lexical_statistics():
    """synth.lexical_global_stats (what ONE shard needs for the global idf / avgdl) equals the
    statistics of the rows themselves, and a shard's rows are the unsharded corpus' rows."""
    import numpy as np
    from triple_hybrid_rag_amd import synth
    n = 150_000
    df, sum_dl = synth.lexical_global_stats(n)
    doc, term, tf = synth.lexical_rows(0, n, n)
    assert np.array_equal(df, np.bincount(term, minlength=synth.vocab_size(n))) and sum_dl == float(tf.sum())
    d2, t2, f2 = synth.lexical_rows(70_000, 50_000, n)
    sel = (doc >= 70_000) & (doc < 120_000)
    assert np.array_equal(d2 + 70_000, doc[sel]) and np.array_equal(t2, term[sel]) and np.array_equal(f2, tf[sel])
