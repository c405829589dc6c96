#!/usr/bin/env python3
"""The BM25 kernel alone at the bench shape, for rocprofv3 --pmc passes:
python3 scripts/pmc_bm25.py [terms] [k] [bench|df256|dffull]   (query mix: the bench's, without
stop words; 256 / 2048 queries sampled in proportion to df)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import triple_hybrid_rag_amd as T
    from triple_hybrid_rag_amd import synth
    nt = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 50
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
    mix = sys.argv[3] if len(sys.argv) > 3 else "bench"
    if mix == "df256":
        qt = torch.from_numpy(np.ascontiguousarray(synth.lexical_queries(256, csr.df_local, nt))).cuda()
    elif mix == "dffull":
        qt = torch.from_numpy(np.ascontiguousarray(synth.lexical_queries(nq, csr.df_local, nt))).cuda()
    else:
        qt = torch.from_numpy(np.ascontiguousarray(synth.lexical_queries(nq, dfq, nt))).cuda()
    for _ in range(4):
        idx.bm25_search(qt, k)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
