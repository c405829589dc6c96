#!/usr/bin/env python3
"""Diagnostic (BM_STAMPS build + THR_BM25_ITEM_LOG=1): one batch of 2048 survey queries, the
library prints every sweep item's start, duration, passes and survivors to stderr.
    THR_LIB_PATH=.../build/libthr_stamps.so THR_BM25_ITEM_LOG=1 python3 scripts/bm25_items.py 2> items.log"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import triple_hybrid_rag_amd as T
    from triple_hybrid_rag_amd import synth
    n, nq = 1_000_000, int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    doc, term, tf = synth.lexical_rows(0, n, n)
    csr = synth.build_lexical_csr(doc, term, tf, n, synth.vocab_size(n))
    df = csr.df_local.astype(np.float64)
    idf = np.log(1.0 + (n - df + 0.5) / (df + 0.5))
    idx = T.GpuIndex()
    idx.n_docs = n
    idx.set_lexical(csr.rowptr, csr.post_doc, csr.post_tf, csr.doclen, idf, csr.sum_dl_local / n)
    qt = synth.lexical_queries(nq, csr.df_local, 4)
    np.save(os.path.join(ROOT, "gpurun_out", "items_queries.npy"), qt)
    np.save(os.path.join(ROOT, "gpurun_out", "items_df.npy"), csr.df_local[qt])
    qd = torch.from_numpy(qt).cuda()
    for i in range(2):
        print(f"=== call {i}", file=sys.stderr, flush=True)
        idx.bm25_search(qd, 50)
        torch.cuda.synchronize()


if __name__ == "__main__":
    main()
