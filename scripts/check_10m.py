"""Scratch: 10M x 768 on one GPU (BASELINE configs[3] corpus size): runs and spot-checks."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
import triple_hybrid_rag_amd as T
from triple_hybrid_rag_amd import synth
from oracle import c_oracle as CO
n, d, nq = 10_000_000, 768, 1536
t0 = time.time(); x = synth.dense_rows(0, n, d); print("gen", round(time.time() - t0, 1), "s", flush=True)
q = synth.dense_queries(nq, d, n)
idx = T.GpuIndex().set_dense(x); idx.reserve(nq, 100)
print("shortlist", idx.shortlist, "rel_err", idx.doc_rel_err, flush=True)
qd = torch.from_numpy(q).cuda()
for _ in range(2): S, I, cnt, nres = idx.dense_search(qd, 100, sync=False)
torch.cuda.synchronize(); t0 = time.time()
for _ in range(5): S, I, cnt, nres = idx.dense_search(qd, 100, sync=False)
torch.cuda.synchronize(); dt = (time.time() - t0) / 5
print("ms per 1536-query batch", round(dt * 1e3, 2), "qps", round(nq / dt), "rescued", int(nres), flush=True)
sub = [0, 1, 2, 3, 1534, 1535]
Se, Ie, _ = CO.dense_topk_exact(x, q[sub], 100, dnorm=idx.dnorm.cpu().numpy())
S, I = S.cpu().numpy(), I.cpu().numpy()
ok = all(np.array_equal(I[qi], Ie[j]) and np.array_equal(S[qi], Se[j]) for j, qi in enumerate(sub))
print("exact vs C oracle on", len(sub), "queries:", ok)
