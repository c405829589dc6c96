#!/usr/bin/env python3
"""Same-box A/B of several builds of libthr_hip.so on the BM25 channel: ONE process, ONE index,
the library handle swapped between the builds (same ABI), three query mixes at the bench shape.

    python3 scripts/ab_bm25.py [docs] [queries] name=path/to/libthr_x.so ...

Every build's pruned result is compared bit for bit with the first build's unpruned one
(every posting scored), and 12 queries per mix with the CPU oracle."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import triple_hybrid_rag_amd as T
    from triple_hybrid_rag_amd import _native as N
    from triple_hybrid_rag_amd import synth
    from oracle import thr_oracle as O
    args = [a for a in sys.argv[1:] if "=" not in a]
    libs = [a.split("=", 1) for a in sys.argv[1:] if "=" in a]
    n = int(args[0]) if len(args) > 0 else 1_000_000
    nq = int(args[1]) if len(args) > 1 else 2048
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
    mixes = {"no_stop_words_2048": synth.lexical_queries(nq, dfq, 4),
             "survey_256": synth.lexical_queries(min(nq, 256), csr.df_local, 4),
             "survey_2048": synth.lexical_queries(nq, csr.df_local, 4),
             "survey_1": synth.lexical_queries(1, csr.df_local, 4)}
    default = N.load()
    handles = {"default": default}
    for name, path in libs:
        lib = C.CDLL(os.path.abspath(path))
        for sym, (res, argt) in N._SIGNATURES.items():
            fn = getattr(lib, sym)
            fn.restype, fn.argtypes = res, argt
        assert lib.thr_abi_version() == N.ABI_VERSION
        handles[name] = lib

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

    out = {"docs": n}
    for mix, qt in mixes.items():
        qd = torch.from_numpy(qt).cuda()
        N._lib = default
        ref = idx.bm25_search(qd, 50, prune=False)
        sub = list(range(0, len(qt), max(1, len(qt) // 12)))[:12]
        Se, Ie = O.bm25_topk(csr.rowptr, csr.post_doc, csr.post_tf, csr.doclen, idf, csr.sum_dl_local / n,
                             qt[sub], n, 50)
        row = {}
        for name, lib in handles.items():
            N._lib = lib
            b = idx.bm25_search(qd, 50)
            torch.cuda.synchronize()
            same = bool(torch.equal(ref[0], b[0]) and torch.equal(ref[1], b[1]))
            ok = sum(int(np.array_equal(b[1][qi].cpu().numpy()[:len(Ie[j])], Ie[j]) and
                         np.array_equal(b[0][qi].cpu().numpy()[:len(Se[j])], Se[j])) for j, qi in enumerate(sub))
            ms = [timed(lambda: idx.bm25_search(qd, 50)) for _ in range(3)]
            row[name] = {"ms": round(min(ms), 4), "ms_runs": [round(m, 4) for m in ms],
                         "equal_to_unpruned": same, "oracle": f"{ok}/{len(sub)}"}
            print(mix, name, row[name], flush=True)
        out[mix] = row
    N._lib = default
    print(json.dumps(out))


if __name__ == "__main__":
    main()
