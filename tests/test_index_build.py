"""Index build and tool-layer mapping (8f.2): host logic, CPU only."""
import asyncio
import math

import numpy as np

from oracle import thr_oracle as O
from triple_hybrid_rag_amd import index_build as IB
from triple_hybrid_rag_amd.rag2.retrieval import RetrievalCandidate, RetrievalResult
from triple_hybrid_rag_amd.tools.crm_knowledge import search_knowledge_base_rag2

TEXTS = ["refund policy for annual plans", "the refund is processed in five days",
         "Politica de reembolso: cinco dias", "shipping and returns policy", ""]


def test_build_lexical_matches_hand_bm25():
    vocab, rowptr, pd, ptf, dl, idf, avgdl = IB.build_lexical(TEXTS)
    n = len(TEXTS)
    assert list(dl) == [5, 7, 5, 4, 0] and abs(avgdl - 21 / 5) < 1e-12
    t = vocab["refund"]
    lo, hi = rowptr[t], rowptr[t + 1]
    assert list(pd[lo:hi]) == [0, 1] and list(ptf[lo:hi]) == [1, 1]
    assert np.all(np.diff(pd[rowptr[vocab["policy"]]:rowptr[vocab["policy"] + 1]]) > 0)
    assert idf[t] == math.log(1.0 + (n - 2 + 0.5) / (2 + 0.5))
    s = O.bm25_scores(rowptr, pd, ptf, dl, idf, avgdl, [vocab["refund"], vocab["policy"]], n)
    nrm0 = 1.2 * (0.25 + 0.75 * (5 / avgdl))
    exp0 = idf[vocab["refund"]] * (2.2 / (1 + nrm0)) + idf[vocab["policy"]] * (2.2 / (1 + nrm0))
    assert s[0] == exp0 and s[4] == -np.inf
    # a shard built with the global vocabulary / statistics scores like the whole corpus
    df_glob = np.diff(rowptr)
    _, rp2, pd2, tf2, dl2, idf2, avg2 = IB.build_lexical(TEXTS[2:], vocab=dict(vocab), n_docs_global=n,
                                                         df_global=df_glob, sum_dl_global=21.0)
    s2 = O.bm25_scores(rp2, pd2, tf2, dl2, idf2, avg2, [vocab["policy"]], 3)
    sfull = O.bm25_scores(rowptr, pd, ptf, dl, idf, avgdl, [vocab["policy"]], n)
    assert np.array_equal(s2, sfull[2:])


def test_build_graph_and_rows_roundtrip(tmp_path):
    children = [{"id": f"c{i}", "parent_id": f"p{i // 2}", "document_id": "d", "text": TEXTS[i],
                 "page": i + 1, "modality": "text",
                 "embedding_1024": None if i == 4 else list(np.eye(8, dtype=np.float32)[i])}
                for i in range(5)]
    parents = [{"id": f"p{j}", "text": f"parent {j}", "section_heading": None} for j in range(3)]
    ents = [{"id": "e_a", "name": "Acme Corp"}, {"id": "e_b", "name": "Refund Desk"},
            {"id": "e_c", "name": "Bob"}]
    rels = [{"subject_entity_id": "e_a", "object_entity_id": "e_b", "confidence": 0.9},
            {"subject_entity_id": "e_b", "object_entity_id": "e_a"},
            {"subject_entity_id": "e_c", "object_entity_id": "e_zzz"}]
    mens = [{"entity_id": "e_b", "child_chunk_id": "c1"}, {"entity_id": "e_a", "child_chunk_id": "c3",
                                                          "confidence": 0.5},
            {"entity_id": "e_a", "child_chunk_id": "c0"}]
    hi = IB.from_rows(children, parents, ents, rels, mens)
    assert hi.docs.shape == (5, 8) and not hi.docs[4].any()
    assert list(hi.ent_rowptr) == [0, 1, 2, 2] and list(hi.ent_col) == [1, 0]   # symmetric, deduplicated
    assert list(hi.men_rowptr) == [0, 2, 3, 3] and list(hi.men_chunk) == [0, 3, 1]
    g = O.graph_scores(hi.ent_rowptr, hi.ent_col, hi.men_rowptr, hi.men_chunk, hi.men_conf, [0], 1, 5)
    assert g[0] == 1.0 and g[3] == 0.5 and g[1] == 0.5 and g[2] == -np.inf
    IB.save(hi, str(tmp_path / "idx"))
    back = IB.load(str(tmp_path / "idx"))
    for name in IB._ARRAYS:
        a, b = getattr(hi, name), getattr(back, name)
        assert (a is None and b is None) or np.array_equal(a, b)
    assert back.store.child_ids == hi.store.child_ids and back.store.vocab == hi.store.vocab
    assert back.store.row_index("c3") == 3 and back.avgdl == hi.avgdl


def test_tool_layer_mapping():
    class Fake:
        def __init__(self, result):
            self.result = result

        async def retrieve(self, query, collection=None, top_k=None):
            self.args = (query, collection, top_k)
            return self.result

    ctx = RetrievalCandidate("c1", "p1", "d1", "child text", 3, "table", lexical_rank=2, semantic_rank=None,
                             graph_rank=1, rrf_score=0.0281234, parent_text="parent text",
                             section_heading=None, rerank_score=0.91239)
    ok = RetrievalResult(True, [ctx], max_rerank_score=0.91239,
                         timings={"planning": 0.0012345, "retrieval": 0.02})
    fk = Fake(ok)
    out = search_knowledge_base_rag2("q", "pricing", 3, retriever=fk)
    assert fk.args == ("q", "pricing", 3)
    assert out["search_type"] == "rag2_triple_hybrid" and out["result_count"] == 1
    assert out["max_rerank_score"] == 0.9124 and out["timings_ms"] == {"planning": 1.23, "retrieval": 20.0}
    r = out["results"][0]
    assert (r["content"], r["title"], r["is_table"], r["relevance_rank"]) == ("parent text", "", True, 1)
    assert (r["similarity_score"], r["rerank_score"], r["lexical_rank"], r["graph_rank"]) == (0.0281, 0.9124, 2, 1)
    refused = RetrievalResult(True, [], refused=True, refusal_reason="Max score 0.01 below threshold 0.6")
    out = search_knowledge_base_rag2("q", None, 5, retriever=Fake(refused))
    assert out == {"success": True, "query": "q", "category": None, "result_count": 0,
                   "search_type": "rag2_triple_hybrid", "refused": True,
                   "refusal_reason": "Max score 0.01 below threshold 0.6", "results": []}
