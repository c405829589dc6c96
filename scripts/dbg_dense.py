import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import triple_hybrid_rag_amd as T
n, d = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(5)
x = rng.standard_normal((n, d)).astype(np.float32); x /= np.linalg.norm(x, axis=1, keepdims=True)
q = rng.standard_normal((70, d)).astype(np.float32)
idx = T.GpuIndex().set_dense(x)
torch.cuda.synchronize(); print("index ok", flush=True)
S, I, cnt, flg = T._native.dense_topk(idx.docs, idx.dnorm, idx.inv_norm, torch.from_numpy(q).cuda(), 100, 128, 0)
torch.cuda.synchronize(); print("dense ok", flg.cpu().numpy()[:8], cnt.cpu().numpy()[:4], flush=True)
print(S[0,:5], I[0,:5])
