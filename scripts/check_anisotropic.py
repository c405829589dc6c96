import os, sys, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
import triple_hybrid_rag_amd as T
from triple_hybrid_rag_amd import synth
from oracle import c_oracle as CO
n, d, nq = 1_000_000, 768, 1536
x = synth.dense_rows(0, n, d)
rng = np.random.default_rng(5)
c = rng.standard_normal(d).astype(np.float32); c /= np.linalg.norm(c)
for mix, name in ((1.0, "cone: cos between rows ~0.5"), (3.0, "narrow cone: ~0.9")):
    y = x + mix * c            # rows share a strong common component (anisotropic embeddings)
    q = synth.dense_queries(nq, d, n) + mix * c
    for mode in ("f16", "f32"):
        idx = T.GpuIndex().set_dense(y, shortlist=mode); idx.reserve(nq, 100)
        qd = torch.from_numpy(q.astype(np.float32)).cuda()
        for _ in range(2): S, I, cnt, nres = idx.dense_search(qd, 100, sync=False)
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(5): S, I, cnt, nres = idx.dense_search(qd, 100, sync=False)
        torch.cuda.synchronize(); ms = (time.time() - t0) / 5 * 1e3
        sub = [0, 1, 2, 777]
        Se, Ie, _ = CO.dense_topk_exact(y, q[sub].astype(np.float32), 100, dnorm=idx.dnorm.cpu().numpy())
        S, I = S.cpu().numpy(), I.cpu().numpy()
        ok = all(np.array_equal(I[qi], Ie[j]) and np.array_equal(S[qi], Se[j]) for j, qi in enumerate(sub))
        print(f"{name} {mode}: {ms:.2f} ms/batch rescued {int(nres)} exact {ok}", flush=True)
        del idx
