#!/usr/bin/env python3
"""BASELINE.json configs[0], the CPU leg ("CPU reference retriever via scripts/test_rag2.py,
plumbing, no GPU"): the SAME command line, flags and --json keys as scripts/test_rag2.py, with the
channels answered by the CPU oracle instead of the MI355X index.  Test infrastructure (it imports
``oracle/``): the product package has no CPU path.

    python tests/oracle_cli.py --batch 100 --org-id org_1 --top-k 5 --graph --json

Only the three scorers differ from the GPU run: the host logic above them (RAG2Retriever: plan,
candidate merge, weighted RRF, truncation, parent expansion, safety) is the product's own Python.
"""
import asyncio
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import c_oracle as CO  # noqa: E402
from oracle import thr_oracle as O  # noqa: E402


def _cli():
    spec = importlib.util.spec_from_file_location("thr_test_rag2_cli", os.path.join(ROOT, "scripts", "test_rag2.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def oracle_client(n_docs: int, dim: int, org_id: str):
    """The Supabase-shaped backend of the CLI with every scorer on the CPU oracle."""
    from triple_hybrid_rag_amd import synth
    from triple_hybrid_rag_amd.backend import CorpusStore, GpuIndexClient, tokenize

    v = synth.vocab_size(n_docs)
    doc, term, tf = synth.lexical_rows(0, n_docs, n_docs)
    csr = synth.build_lexical_csr(doc, term, tf, n_docs, v)
    idf = O.bm25_idf(n_docs, csr.df_local)
    avgdl = csr.sum_dl_local / n_docs
    g = synth.build_graph(n_docs)
    x = synth.dense_rows(0, n_docs, dim)
    dn = CO.doc_norms(x)
    store = CorpusStore.synthetic(n_docs, vocab_size=v, n_entities=synth.n_entities(n_docs))

    class OracleIndexClient(GpuIndexClient):
        def __init__(self):   # (no GpuIndex: the parent's set-up is what needs the device)
            self.store, self.org_id, self.lexical_and = store, org_id, False
            self.token_embedder = self.image_index = self.image_rows = None

        def _semantic(self, embedding, limit, collection):
            q = np.asarray([list(map(float, embedding))], dtype=np.float32)
            S, I, cnt = CO.dense_topk_exact(x, q, min(256, limit), dnorm=dn)
            return self._rows(I[0].tolist(), S[0].tolist(), int(cnt[0]), "similarity", limit)

        defers_readback = False

        def _lexical(self, query, limit, collection, defer=False):
            terms = []
            for tok in tokenize(query):
                t = store.vocab.get(tok)
                if t is not None and t not in terms:
                    terms.append(t)
            if not terms:
                return []
            S, I = O.bm25_topk(csr.rowptr, csr.post_doc, csr.post_tf, csr.doclen, idf, avgdl,
                               [terms[:32]], n_docs, min(128, limit))
            return self._rows(list(I[0]), list(S[0]), len(I[0]), "rank", limit)

        def graph_chunks(self, seeds, top_k, hops=2):
            _, I = O.graph_topk(g.ent_rowptr, g.ent_col, g.men_rowptr, g.men_chunk, g.men_conf,
                                [seeds], hops, n_docs, min(128, top_k))
            return [store.child_ids[int(i)] for i in I[0]]

    return OracleIndexClient()


def run(argv):
    """-> the list of result dicts the CLI's --json would print."""
    from triple_hybrid_rag_amd.config import SETTINGS
    saved = dict(SETTINGS.__dict__)   # (the CLI sets the thresholds / switches of its run globally)
    out = []
    try:
        rc = asyncio.run(_cli().main(argv, make_client=oracle_client, out=out))
    finally:
        SETTINGS.__dict__.update(saved)
    return rc, out


if __name__ == "__main__":
    import json
    rc, rows = run(sys.argv[1:])
    if "--json" in sys.argv:
        print(json.dumps(rows if "--batch" in sys.argv else rows[0]))
    sys.exit(rc)
