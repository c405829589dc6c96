"""Scratch: time select_rescore / kth_select variants under rocprofv3 (not part of the product)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import triple_hybrid_rag_amd as T
from triple_hybrid_rag_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 125000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 100
x = torch.from_numpy(synth.dense_rows(0, n, 768)).cuda()
q = torch.from_numpy(synth.dense_queries(1024, 768, n)).cuda()
idx = T.GpuIndex().set_dense(x, shortlist="f16-inline")
for _ in range(5):
    idx.dense_search(q, k, rescue=False)
torch.cuda.synchronize()
