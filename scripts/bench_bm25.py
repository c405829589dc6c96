#!/usr/bin/env python3
"""BM25 kernel alone at the bench shape (1M docs, 2048 queries of 4 terms): the bench's query mix
(stop words excluded) and the df-proportional mix (stop words included), with and without the
WAND-style bounds; results checked equal.  python3 scripts/bench_bm25.py [docs] [queries]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import triple_hybrid_rag_amd as T
    from triple_hybrid_rag_amd import synth
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    nq = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    v = synth.vocab_size(n)
    doc, term, tf = synth.lexical_rows(0, n, n)
    csr = synth.build_lexical_csr(doc, term, tf, n, v)
    df = csr.df_local.astype(np.float64)
    idf = np.log(1.0 + (n - df + 0.5) / (df + 0.5))
    idx = T.GpuIndex()
    idx.n_docs = n
    idx.set_lexical(csr.rowptr, csr.post_doc, csr.post_tf, csr.doclen, idf, csr.sum_dl_local / n)
    dfq = csr.df_local.copy()
    dfq[dfq > 0.01 * n] = 0
    mixes = {"bench_mix_no_stop_words": synth.lexical_queries(nq, dfq, 4),
             "df_proportional_with_stop_words": synth.lexical_queries(min(nq, 256), csr.df_local, 4),
             "df_proportional_full_batch": synth.lexical_queries(nq, csr.df_local, 4)}
    out = {"docs": n}
    from oracle import thr_oracle as O
    for name, qt in mixes.items():
        qd = torch.from_numpy(qt).cuda()

        def timed(fn, reps=5):
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps
        a = idx.bm25_search(qd, 50, prune=False)
        b = idx.bm25_search(qd, 50, prune=True)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
        sub = list(range(0, len(qt), max(1, len(qt) // 12)))[:12]
        Se, Ie = O.bm25_topk(csr.rowptr, csr.post_doc, csr.post_tf, csr.doclen, idf, csr.sum_dl_local / n,
                             qt[sub], n, 50)
        ok = sum(int(np.array_equal(b[1][qi].cpu().numpy()[:len(Ie[j])], Ie[j]) and
                     np.array_equal(b[0][qi].cpu().numpy()[:len(Se[j])], Se[j])) for j, qi in enumerate(sub))
        post = sum(int(csr.df_local[t]) for row in qt for t in row if t >= 0)
        ms_all = timed(lambda: idx.bm25_search(qd, 50, prune=False))
        ms = timed(lambda: idx.bm25_search(qd, 50, prune=True))
        out[name] = {"queries": len(qt), "postings_per_query": round(post / len(qt), 1),
                     "ms_every_posting_scored": round(ms_all, 3), "ms_with_bounds": round(ms, 3),
                     # (bytes of EVERY posting over time: an HBM rate only for the unpruned call)
                     "every_posting_scored_GBps": round((post * 12 + len(qt) * 64) / ms_all / 1e6, 1),
                     "pruned_equivalent_GBps_not_a_traffic_figure": round((post * 12 + len(qt) * 64) / ms / 1e6, 1),
                     "bit_equal_to_oracle": f"{ok}/{len(sub)}"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
