#!/usr/bin/env python3
"""RAG 2.0 retrieval CLI over the GPU index -- counterpart of the reference's
scripts/test_rag2.py:151-243 (same flags, same --json keys).  The reference needs a live
Supabase + embedding server; this one builds the synthetic in-process index (BASELINE.json
config 0: 10k-doc / 768-d) and answers from the MI355X.

    python scripts/test_rag2.py --query "t120 t77 entity12" --org-id org_1 --top-k 5 --json
"""
import argparse
import asyncio
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def build_client(n_docs: int, dim: int, org_id: str):
    import numpy as np
    import triple_hybrid_rag_amd as T
    from triple_hybrid_rag_amd import synth
    from triple_hybrid_rag_amd.backend import CorpusStore, GpuIndexClient

    v = synth.vocab_size(n_docs)
    doc, term, tf = synth.lexical_rows(0, n_docs, n_docs)
    csr = synth.build_lexical_csr(doc, term, tf, n_docs, v)
    df = csr.df_local.astype(np.float64)
    idf = np.log(1.0 + (n_docs - df + 0.5) / (df + 0.5))
    g = synth.build_graph(n_docs)
    idx = (T.GpuIndex().set_dense(synth.dense_rows(0, n_docs, dim))
           .set_lexical(csr.rowptr, csr.post_doc, csr.post_tf, csr.doclen, idf,
                        csr.sum_dl_local / n_docs)
           .set_graph(g.ent_rowptr, g.ent_col, g.men_rowptr, g.men_chunk, g.men_conf))
    store = CorpusStore.synthetic(n_docs, vocab_size=v, n_entities=synth.n_entities(n_docs))
    return GpuIndexClient(idx, store, org_id=org_id)


def batch_queries(n: int, n_docs: int):
    """Deterministic query texts over the synthetic corpus' vocabulary and entity names: three
    terms and one entity keyword each (the lexical channel sees the terms, the graph channel the
    entity name, the semantic channel a hash embedding of the whole text)."""
    import numpy as np
    from triple_hybrid_rag_amd import synth
    r = np.random.Generator(np.random.PCG64([4321, 9]))
    v, e = synth.vocab_size(n_docs), synth.n_entities(n_docs)
    return [" ".join([f"t{int(t)}" for t in r.integers(0, min(v, 2000), 3)] + [f"entity{int(r.integers(0, e))}"])
            for _ in range(n)]


async def main(argv=None, make_client=None, out=None) -> int:
    """``make_client`` (n_docs, dim, org_id) -> backend: the GPU index by default; the CPU leg of
    BASELINE.json configs[0] (tests/oracle_cli.py) passes the oracle-backed one.  ``out``: a
    list that receives what --json would print (tests)."""
    ap = argparse.ArgumentParser(description="RAG 2.0 Retrieval Test CLI (MI355X index)")
    ap.add_argument("--query", "-q")
    ap.add_argument("--batch", type=int, default=0,
                    help="(not in the reference) run this many synthetic queries -- BASELINE.json "
                         "config 0 is 100 -- against ONE index build; --json then prints a list")
    ap.add_argument("--org-id", "-o", required=True)
    ap.add_argument("--collection", "-c")
    ap.add_argument("--top-k", "-k", type=int, default=5)
    ap.add_argument("--full", "-f", action="store_true")
    ap.add_argument("--graph", "-g", action="store_true")
    ap.add_argument("--verbose", "-v", action="store_true")
    ap.add_argument("--json", action="store_true")
    ap.add_argument("--docs", type=int, default=10_000)
    ap.add_argument("--dim", type=int, default=768)
    args = ap.parse_args(argv)

    from triple_hybrid_rag_amd.config import SETTINGS
    from triple_hybrid_rag_amd.rag2.embedder import HashEmbedder
    from triple_hybrid_rag_amd.rag2.query_planner import QueryPlanner
    from triple_hybrid_rag_amd.rag2.retrieval import RAG2Retriever

    # RRF-scale scores never reach the 0.6 refusal gate without a cross-encoder
    SETTINGS.rag2_safety_threshold = 0.0
    SETTINGS.rag2_denoise_alpha = 0.0
    SETTINGS.rag2_rerank_enabled = False
    SETTINGS.rag2_graph_enabled = args.graph
    retriever = RAG2Retriever(org_id=args.org_id,
                              embedder=HashEmbedder(model_dim=4096, store_dim=args.dim),
                              query_planner=QueryPlanner(graph=args.graph),
                              graph_enabled=args.graph)
    if not args.query and not args.batch:
        ap.error("--query or --batch is required")
    retriever._supabase = (make_client or build_client)(args.docs, args.dim, args.org_id)

    def as_json(result):   # the reference's --json keys (scripts/test_rag2.py:214-235)
        return {"success": result.success, "refused": result.refused,
                "refusal_reason": result.refusal_reason, "max_score": result.max_rerank_score,
                "contexts": [{"child_id": c.child_id, "document_id": c.document_id, "page": c.page,
                              "rrf_score": c.rrf_score, "rerank_score": c.rerank_score,
                              "text": c.text[:500], "section": c.section_heading}
                             for c in result.contexts],
                "timings": result.timings}

    if args.batch:
        rows = []
        for text in batch_queries(args.batch, args.docs):
            r = await retriever.retrieve(query=text, collection=args.collection, top_k=args.top_k)
            rows.append({"query": text, **as_json(r)})
        if out is not None:
            out.extend(rows)
        else:
            print(json.dumps(rows))
        return 0
    result = await retriever.retrieve(query=args.query, collection=args.collection,
                                      top_k=args.top_k)
    if out is not None:
        out.append(as_json(result))
    elif args.json:
        print(json.dumps(as_json(result), indent=2))
    else:
        print(f"success={result.success} refused={result.refused} reason={result.refusal_reason}")
        for stage, seconds in result.timings.items():
            print(f"   {stage}: {seconds * 1e3:.2f}ms")
        for i, c in enumerate(result.contexts[:5], 1):
            print(f"--- Context {i} --- rrf={c.rrf_score:.4f} doc={c.document_id} page={c.page} "
                  f"section={c.section_heading}\n   {c.text if args.full else c.text[:300]}")
    return 0 if result.success and not result.refused else 1


if __name__ == "__main__":
    sys.exit(asyncio.run(main()))
