#!/usr/bin/env python3
"""Eager step vs the same step replayed as a hipGraph (index.StepGraph), for corpus sizes that
stand for a shard: python3 scripts/graph_probe.py [pipeline dense|dense_bm25|triple] [docs ...]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import triple_hybrid_rag_amd as T
    from triple_hybrid_rag_amd import synth
    from triple_hybrid_rag_amd.index import StepGraph
    cfg = sys.argv[1] if len(sys.argv) > 1 else "dense"
    sizes = [int(a) for a in sys.argv[2:]] or [125_000, 500_000, 1_000_000]
    nq, dim = 2048, 768
    out = {}
    for n in sizes:
        idx = T.GpuIndex().set_dense(synth.dense_rows(0, n, dim))
        qtd = sd = None
        if cfg != "dense":
            doc, term, tf = synth.lexical_rows(0, n, n)
            csr = synth.build_lexical_csr(doc, term, tf, n, synth.vocab_size(n))
            df = csr.df_local.astype(np.float64)
            idx.set_lexical(csr.rowptr, csr.post_doc, csr.post_tf, csr.doclen,
                            np.log(1.0 + (n - df + 0.5) / (df + 0.5)), csr.sum_dl_local / n)
            qtd = torch.from_numpy(synth.lexical_queries(nq, csr.df_local, 4)).cuda()
        if cfg == "triple":
            g = synth.build_graph(n)
            idx.set_graph(g.ent_rowptr, g.ent_col, g.men_rowptr, g.men_chunk, g.men_conf)
            sd = torch.from_numpy(synth.graph_queries(nq, n, 3)).cuda()
        idx.reserve(nq, 100)
        raw = np.zeros((nq, 4096), dtype=np.float32)
        raw[:, :dim] = synth.dense_queries(nq, dim, n) * np.float32(3.7)
        raw_dev = torch.from_numpy(raw).cuda()

        def step():
            return idx.retrieve_batch(T._native.embed_postproc(raw_dev, dim), qtd, sd, top_k=10)

        def wall(fn, steps=20):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                r = fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / steps * 1e3, r
        eager_ms, r0 = wall(step)
        ids0 = r0.ids.clone()
        sg = StepGraph(step)
        graph_ms, r1 = wall(sg.replay)
        same = bool(torch.equal(ids0, r1.ids))
        # host time of enqueueing one eager step (no sync): what the launches cost the CPU
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step()
        host_ms = (time.perf_counter() - t0) * 1e3
        torch.cuda.synchronize()
        out[n] = {"eager_ms": round(eager_ms, 3), "graph_ms": round(graph_ms, 3), "same_ids": same,
                  "host_enqueue_ms_one_eager_step": round(host_ms, 3), "rescued": int(r1.rescued)}
        print(cfg, n, out[n], flush=True)
        del idx, sg
        torch.cuda.empty_cache()
    print(json.dumps({cfg: out}))


if __name__ == "__main__":
    main()
