#!/usr/bin/env python3
"""Per-channel timings on one MI355X (BASELINE.json configs 2-4 at single-GPU size).

Prints one JSON object: for each channel kernel the average launch time (HIP events on the
launch stream), the algorithmic bytes per query (SURVEY.md section 8d formulas evaluated on the
actual queries) and the resulting GB/s; plus queries/s of the hybrid pipelines.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timed(fn, reps=5):
    import torch
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--queries", type=int, default=1024)
    ap.add_argument("--token-docs", type=int, default=100_000)
    ap.add_argument("--stop-frac", type=float, default=0.01,
                    help="terms with df > stop_frac*N are treated as stop words for query sampling")
    args = ap.parse_args()
    import torch
    import triple_hybrid_rag_amd as T
    from triple_hybrid_rag_amd import synth
    n, d, nq = args.docs, args.dim, args.queries
    t0 = time.time()
    v = synth.vocab_size(n)
    doc, term, tf = synth.lexical_rows(0, n, n)
    csr = synth.build_lexical_csr(doc, term, tf, n, v)
    df = csr.df_local.astype(np.float64)
    idf = np.log(1.0 + (n - df + 0.5) / (df + 0.5))
    avgdl = csr.sum_dl_local / n
    dfq = csr.df_local.copy()
    dfq[dfq > args.stop_frac * n] = 0  # stop words are not query terms
    qt = synth.lexical_queries(nq, dfq, 4)
    g = synth.build_graph(n)
    seeds = synth.graph_queries(nq, n, 3)
    x = synth.dense_rows(0, n, d)
    q = synth.dense_queries(nq, d, n)
    gen_s = time.time() - t0
    idx = (T.GpuIndex().set_dense(x)
           .set_lexical(csr.rowptr, csr.post_doc, csr.post_tf, csr.doclen, idf, avgdl)
           .set_graph(g.ent_rowptr, g.ent_col, g.men_rowptr, g.men_chunk, g.men_conf))
    nt = args.token_docs
    rows = torch.nn.functional.normalize(
        torch.randn(nt, 128, 128, device="cuda", dtype=torch.float32), dim=2).to(torch.float16)
    idx.set_tokens(rows)            # fragment-major image (thr_maxsim_pack)
    qtok = torch.nn.functional.normalize(torch.randn(nq, 32, 128, device="cuda"), dim=2).to(torch.float16)
    qd, qtd, sd = torch.from_numpy(q).cuda(), torch.from_numpy(qt).cuda(), torch.from_numpy(seeds).cuda()
    out = {"docs": n, "dim": d, "queries": nq, "input_gen_s": round(gen_s, 1)}

    # BM25
    ms = timed(lambda: idx.bm25_search(qtd, 50))
    post = sum(int(csr.df_local[t]) for row in qt for t in row if t >= 0)
    by = post * 12 + nq * 4 * 16
    out["bm25"] = {"ms": round(ms, 3), "queries_per_s": round(nq / ms * 1e3), "postings_per_query": post / nq,
                   "alg_bytes_per_query": by / nq, "GBps": round(by / ms / 1e6, 2)}
    # graph
    ms = timed(lambda: idx.graph_search(sd, 50, 2))
    out["graph"] = {"ms": round(ms, 3), "queries_per_s": round(nq / ms * 1e3),
                    "alg_bytes_per_query_est": 8800, "GBps": round(8800 * nq / ms / 1e6, 2)}
    # maxsim: 100 candidates per query
    cand = torch.randint(0, nt, (nq, 100), device="cuda", dtype=torch.int32)
    ms = timed(lambda: T._native.maxsim(qtok, idx.tokens, cand, packed=idx.tokens_packed))
    by = nq * 100 * 128 * 128 * 2
    fl = 2.0 * nq * 100 * 32 * 128 * 128
    out["maxsim"] = {"ms": round(ms, 3), "queries_per_s": round(nq / ms * 1e3), "GBps": round(by / ms / 1e6, 1),
                     "frac_hbm_8TBps": round(by / ms / 1e6 / 8000, 4), "TFLOPs": round(fl / ms / 1e9, 2),
                     "token_layout": "fragment-major (thr_maxsim_pack)" if idx.tokens_packed else "row-major"}
    ms_r = timed(lambda: T._native.maxsim(qtok, rows, cand))
    out["maxsim"]["row_major_ms"] = round(ms_r, 3)
    del rows
    # fused pipelines
    for name, kw in (("dense_only", {}), ("dense_bm25", {"query_terms": qtd}),
                     ("triple_hybrid", {"query_terms": qtd, "query_seeds": sd})):
        ms = timed(lambda: idx.retrieve_batch(qd, top_k=10, **kw), reps=3)
        out[name] = {"ms_per_batch": round(ms, 3), "queries_per_s": round(nq / ms * 1e3)}
    ms = timed(lambda: idx.retrieve_batch(qd, qtd, sd, top_k=10, qtok=qtok, rerank_top_k=100), reps=3)
    # rerank candidates are fused ids over the whole corpus; only those inside the token store score
    out["triple_hybrid_rerank"] = {"ms_per_batch": round(ms, 3), "queries_per_s": round(nq / ms * 1e3),
                                   "note": f"token store covers the first {nt} docs"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
