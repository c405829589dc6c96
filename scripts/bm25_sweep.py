#!/usr/bin/env python3
"""BM25 kernel alone, 1M docs x 2048 queries: time against the number of query terms (the posting
count) and k, plus a rare-term mix (lists of <= 200 postings: the fixed cost of a query).
python3 scripts/bm25_sweep.py"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import triple_hybrid_rag_amd as T
    from triple_hybrid_rag_amd import synth
    n, nq = 1_000_000, 2048
    v = synth.vocab_size(n)
    d_, t_, f_ = synth.lexical_rows(0, n, n)
    csr = synth.build_lexical_csr(d_, t_, f_, n, v)
    df = csr.df_local.astype(np.float64)
    idf = np.log(1.0 + (n - df + 0.5) / (df + 0.5))
    idx = T.GpuIndex()
    idx.n_docs = n
    idx.set_lexical(csr.rowptr, csr.post_doc, csr.post_tf, csr.doclen, idf, csr.sum_dl_local / n)
    dfq = csr.df_local.copy()
    dfq[dfq > 0.01 * n] = 0

    def timed(fn, reps=10):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    out = {}
    for nt in (1, 2, 4, 8):
        qt_np = synth.lexical_queries(nq, dfq, nt)
        qt = torch.from_numpy(np.ascontiguousarray(qt_np)).cuda()
        pp = float(np.mean([sum(csr.df_local[t] for t in row if t >= 0) for row in qt_np[:256]]))
        for k in (10, 50):
            out[f"terms{nt}_k{k}"] = {"postings": round(pp, 1),
                                      "ms": round(timed(lambda: idx.bm25_search(qt, k)), 4)}
    dfr = csr.df_local.copy()
    dfr[dfr > 200] = 0
    qt = torch.from_numpy(np.ascontiguousarray(synth.lexical_queries(nq, dfr, 4))).cuda()
    out["rare4_k50"] = {"ms": round(timed(lambda: idx.bm25_search(qt, 50)), 4)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
