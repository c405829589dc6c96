import os, sys, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
import triple_hybrid_rag_amd as T
from triple_hybrid_rag_amd import synth
from oracle import thr_oracle as O
n, nq = 4_000_000, 256
t0 = time.time()
v = synth.vocab_size(n)
doc, term, tf = synth.lexical_rows(0, n, n)
csr = synth.build_lexical_csr(doc, term, tf, n, v)
df = csr.df_local.astype(np.float64)
idf = np.log(1.0 + (n - df + 0.5) / (df + 0.5))
avgdl = csr.sum_dl_local / n
qt = synth.lexical_queries(nq, csr.df_local, 4)      # stop words included: the longest lists
g = synth.build_graph(n)
seeds = synth.graph_queries(nq, n, 3)
print("gen", round(time.time() - t0, 1), "s; postings", len(csr.post_doc), flush=True)
idx = T.GpuIndex()
idx.n_docs = n
idx.set_lexical(csr.rowptr, csr.post_doc, csr.post_tf, csr.doclen, idf, avgdl)
idx.set_graph(g.ent_rowptr, g.ent_col, g.men_rowptr, g.men_chunk, g.men_conf)
qtd = torch.from_numpy(qt).cuda()
def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ms_all = timed(lambda: idx.bm25_search(qtd, 50, prune=False))
ms_wand = timed(lambda: idx.bm25_search(qtd, 50, prune=True))
print(f"bm25 {nq} stop-word queries, 4M docs: every posting scored {ms_all:.2f} ms, with WAND-style bounds "
      f"{ms_wand:.2f} ms ({ms_all / ms_wand:.2f}x)", flush=True)
S0, I0, c0 = idx.bm25_search(qtd, 50, prune=False)
S, I, cnt = idx.bm25_search(qtd, 50)
assert torch.equal(I0, I) and torch.equal(S0, S)
sub = list(range(0, nq, 32))
Se, Ie = O.bm25_topk(csr.rowptr, csr.post_doc, csr.post_tf, csr.doclen, idf, avgdl, qt[sub], n, 50)
S, I = S.cpu().numpy(), I.cpu().numpy()
ok = all(np.array_equal(I[qi][:len(Ie[j])], Ie[j]) and np.array_equal(S[qi][:len(Se[j])], Se[j]) for j, qi in enumerate(sub))
print("bm25 exact:", ok, "max postings/query", max(sum(int(csr.df_local[t]) for t in row if t >= 0) for row in qt), flush=True)
Sg, Ig, cg = idx.graph_search(torch.from_numpy(seeds).cuda(), 50, 2)
Se, Ie = O.graph_topk(g.ent_rowptr, g.ent_col, g.men_rowptr, g.men_chunk, g.men_conf, seeds[sub], 2, n, 50)
Sg, Ig = Sg.cpu().numpy(), Ig.cpu().numpy()
ok2 = all(np.array_equal(Ig[qi][:len(Ie[j])], Ie[j]) and np.array_equal(Sg[qi][:len(Se[j])], Se[j]) for j, qi in enumerate(sub))
print("graph exact:", ok2)
assert ok and ok2
