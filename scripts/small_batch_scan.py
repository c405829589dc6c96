import sys, os, json, torch, time
sys.path.insert(0, os.getcwd())
import numpy as np
import triple_hybrid_rag_amd as T
from triple_hybrid_rag_amd import synth
n, d = 1_000_000, 768
x = torch.from_numpy(synth.dense_rows(0, n, d)).cuda()
idx = T.GpuIndex().set_dense(x, shortlist="f16")
out = {}
for nq in (1, 32, 64, 256):
    q = torch.from_numpy(synth.dense_queries(nq, d, n)).cuda()
    idx.dense_search(q, 100)
    def timed(fn, reps=20):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    ts = timed(lambda: idx.scan_probe(q)); tt = timed(lambda: idx.dense_search(q, 100, rescue=False))
    out[nq] = {"scan_ms": round(ts, 4), "hbm_tbps": round(n * d * 2 / ts / 1e9, 3), "search_ms": round(tt, 4)}
print(json.dumps(out))
