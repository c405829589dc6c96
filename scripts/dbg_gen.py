import os, time, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), flush=True)
try:
    print(open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e:
    print("no cgroup info", e)
r = np.random.Generator(np.random.PCG64([1234, 0]))
t = time.time(); x = r.standard_normal((65536, 768), dtype=np.float32); print("gen1 block", time.time() - t, flush=True)
t = time.time(); n = np.sqrt(np.einsum("ij,ij->i", x, x)); x /= n[:, None]; print("norm", time.time() - t, flush=True)
from concurrent.futures import ThreadPoolExecutor
def f(b):
    r = np.random.Generator(np.random.PCG64([1234, b])); return r.standard_normal((65536, 768), dtype=np.float32).sum()
for w in (1, 4, 16):
    t = time.time()
    with ThreadPoolExecutor(w) as ex: list(ex.map(f, range(16)))
    print("workers", w, time.time() - t, flush=True)
from triple_hybrid_rag_amd import synth
t = time.time(); synth.dense_rows(0, 1_000_000, 768); print("dense_rows 1M", time.time() - t)
t = time.time(); synth.dense_queries(1024, 768, 1_000_000); print("queries", time.time() - t)
