import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
import triple_hybrid_rag_amd as T
from triple_hybrid_rag_amd import synth
from oracle import c_oracle as CO
def check(n, d, nq, k, modes=("f16", "f16-inline", "f32"), sub=8):
    x = synth.dense_rows(0, n, d); q = synth.dense_queries(nq, d, n)
    pick = list(range(0, nq, max(1, nq // sub)))[:sub]
    Se, Ie, ce = CO.dense_topk_exact(x, q[pick], k)
    for m in modes:
        if m != "f32" and d not in (512, 768, 1024): continue
        idx = T.GpuIndex().set_dense(x, shortlist=m)
        S, I, cnt, nres = idx.dense_search(torch.from_numpy(q).cuda(), k)
        S, I = S.cpu().numpy(), I.cpu().numpy()
        ok = all(np.array_equal(I[qi][:len(Ie[j])], Ie[j]) and np.array_equal(S[qi][:len(Se[j])], Se[j]) for j, qi in enumerate(pick))
        print(f"n={n} d={d} nq={nq} k={k} {m}: exact={ok} rescued={nres}", flush=True)
        assert ok
check(100_000, 768, 300, 256)
check(100_000, 768, 300, 200)
check(200_000, 768, 8192, 100)
check(50_000, 512, 1000, 100)
check(3_000_000, 1024, 700, 50, sub=4)
check(70_001, 256, 77, 33, modes=("f32",))
print("adhoc ok")
