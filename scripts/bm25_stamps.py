#!/usr/bin/env python3
"""Phase stamps of the BM25 kernel (diagnostic build): one launch per query mix, the library
prints the share of each phase of a pass and a few counters.
    python3 -c "import triple_hybrid_rag_amd as T; print(T._build.build_variant('stamps', ['BM_STAMPS']))"
    THR_LIB_PATH=.../build/libthr_stamps.so python3 scripts/bm25_stamps.py"""
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
    for name, qt in (("bench mix (no stop words)", synth.lexical_queries(nq, dfq, 4)),
                     ("df-proportional, 256 queries", synth.lexical_queries(256, csr.df_local, 4)),
                     ("df-proportional, full batch", synth.lexical_queries(nq, csr.df_local, 4))):
        qd = torch.from_numpy(qt).cuda()
        print(name, file=sys.stderr, flush=True)
        for _ in range(2):
            idx.bm25_search(qd, 50)
            torch.cuda.synchronize()


if __name__ == "__main__":
    main()
