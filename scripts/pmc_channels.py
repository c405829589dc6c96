#!/usr/bin/env python3
"""One channel kernel alone at the bench shape (1M docs, 2048 queries), for rocprofv3 --pmc passes:
python3 scripts/pmc_channels.py bm25|graph|maxsim"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import triple_hybrid_rag_amd as T
    from triple_hybrid_rag_amd import synth
    which = sys.argv[1]
    n, nq = 1_000_000, 2048
    idx = T.GpuIndex()
    idx.n_docs = n
    if which == "bm25":
        v = synth.vocab_size(n)
        d_, t_, f_ = synth.lexical_rows(0, n, n)
        csr = synth.build_lexical_csr(d_, t_, f_, n, v)
        df = csr.df_local.astype(np.float64)
        idf = np.log(1.0 + (n - df + 0.5) / (df + 0.5))
        idx.set_lexical(csr.rowptr, csr.post_doc, csr.post_tf, csr.doclen, idf, csr.sum_dl_local / n)
        dfq = csr.df_local.copy()
        dfq[dfq > 0.01 * n] = 0
        qt = torch.from_numpy(np.ascontiguousarray(synth.lexical_queries(nq, dfq, 4))).cuda()
        run = lambda: idx.bm25_search(qt, 50)
    elif which == "graph":
        g = synth.build_graph(n)
        idx.set_graph(g.ent_rowptr, g.ent_col, g.men_rowptr, g.men_chunk, g.men_conf)
        seeds = torch.from_numpy(np.ascontiguousarray(synth.graph_queries(nq, n, 3))).cuda()
        run = lambda: idx.graph_search(seeds, 50, 2)
    else:
        import bench
        idx.set_tokens(bench.device_tokens(torch, 0, n))
        gen = torch.Generator(device="cuda")
        gen.manual_seed(4321 + 3)
        qtok = torch.nn.functional.normalize(torch.randn((nq, 32, 128), generator=gen, device="cuda"),
                                             dim=2).to(torch.float16)
        cand = torch.randint(0, n, (nq, 100), generator=gen, device="cuda", dtype=torch.int64)
        run = lambda: idx.maxsim(qtok, cand)
    for _ in range(4):
        run()
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
