"""BASELINE.json configs[0] in the container (no GPU): the scripts/test_rag2.py command line over
the oracle-backed client -- 10k-doc / 768-d synthetic corpus, 100 queries, every channel on."""
import numpy as np

from oracle import c_oracle as CO
from oracle import thr_oracle as O
from tests import oracle_cli


def test_config0_cpu_reference_retriever_via_the_cli():
    from triple_hybrid_rag_amd import synth
    from triple_hybrid_rag_amd.rag2.embedder import HashEmbedder
    n, d, top_k = 10000, 768, 5
    rc, res = oracle_cli.run(["--batch", "100", "--org-id", "org_1", "--top-k", str(top_k), "--graph", "--json",
                              "--docs", str(n)])
    assert rc == 0 and len(res) == 100
    # the reference's --json keys (scripts/test_rag2.py:214-235)
    assert set(res[0]) == {"query", "success", "refused", "refusal_reason", "max_score", "contexts", "timings"}
    assert set(res[0]["contexts"][0]) == {"child_id", "document_id", "page", "rrf_score", "rerank_score",
                                          "text", "section"}
    assert all(r["success"] and not r["refused"] and len(r["contexts"]) == top_k for r in res)
    # a few of them against the channels called directly (the plumbing adds nothing, loses nothing)
    x = synth.dense_rows(0, n, d)
    v = synth.vocab_size(n)
    doc, term, tf = synth.lexical_rows(0, n, n)
    csr = synth.build_lexical_csr(doc, term, tf, n, v)
    idf, avgdl = O.bm25_idf(n, csr.df_local), csr.sum_dl_local / n
    emb = HashEmbedder(model_dim=4096, store_dim=d)
    graph_hits = 0
    for r in res[::10]:
        text = r["query"]
        terms = []
        for kw in text.split():
            if kw.startswith("t") and kw[1:].isdigit() and int(kw[1:]) < v and int(kw[1:]) not in terms:
                terms.append(int(kw[1:]))
        _, Il = O.bm25_topk(csr.rowptr, csr.post_doc, csr.post_tf, csr.doclen, idf, avgdl, [terms], n, 50)
        _, Id, _ = CO.dense_topk_exact(x, np.array([emb.embed_query(text)], dtype=np.float32), 100)
        two, _ = O.fused_topk_ids(list(Il[0]), list(Id[0]), None, top_k)
        got = [int(c["child_id"][1:]) for c in r["contexts"]]
        graph_hits += got != two          # (the graph channel moves some of them: it is on)
        assert all(c["rrf_score"] > 0 and c["text"] == f"chunk {c['child_id'][1:]}" for c in r["contexts"])
        assert [c["rrf_score"] for c in r["contexts"]] == sorted((c["rrf_score"] for c in r["contexts"]), reverse=True)
        assert {"planning", "retrieval", "fusion", "expansion", "safety"} <= set(r["timings"])
    assert graph_hits > 0
