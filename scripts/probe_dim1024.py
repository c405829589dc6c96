#!/usr/bin/env python3
"""Parity + timing probe of the dim-1024 copy scan: python3 scripts/probe_dim1024.py [docs] [queries]"""
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
    from oracle import c_oracle as CO
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    nq = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    d = 1024
    x = synth.dense_rows(0, n, d)
    q = synth.dense_queries(nq, d, n)
    mask = sys.argv[3] if len(sys.argv) > 3 else ""
    if mask == "lo":      # only dims [0, 512) carry anything
        x[:, 512:] = 0
    elif mask == "hi":
        x[:, :512] = 0
    elif mask.startswith("q"):   # only one quarter
        j = int(mask[1:])
        keep = x[:, 256 * j:256 * (j + 1)].copy()
        x[:] = 0
        x[:, 256 * j:256 * (j + 1)] = keep
    idx = T.GpuIndex().set_dense(x)
    print("shortlist", idx.shortlist, flush=True)
    S, I, cnt, resc = idx.dense_search(torch.from_numpy(q).cuda(), 100)
    torch.cuda.synchronize()
    print("searched, rescued", resc, flush=True)
    sub = list(range(0, nq, max(1, nq // 24)))[:24]
    Se, Ie, _ = CO.dense_topk_exact(x, q[sub], 100)
    ok = sum(int(np.array_equal(I[qi].cpu().numpy(), Ie[j]) and np.array_equal(S[qi].cpu().numpy(), Se[j]))
             for j, qi in enumerate(sub))
    print(f"parity {ok}/{len(sub)}", flush=True)
    qd = torch.from_numpy(q).cuda()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    idx.scan_probe(qd)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(5):
        idx.scan_probe(qd)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"scan {ms:.4f} ms = {2.0 * n * d * nq / ms / 1e9:.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
