"""GPU parity at the BASELINE.json sizes: configs[2] (1M docs, dense + BM25 + RRF(0.8/0.7)) and
ONE rank's shard of configs[3] / configs[4] (10M docs over 8 GPUs: 1.25M rows per GPU, global
idf / avgdl, replicated entity graph, token store generated on the device) -- every channel's
per-shard top-k, the fusion and the shard's MaxSim scores against the CPU oracle run on the same
slice.  ids / order / cosine / BM25 / graph / RRF scores bit-identical; MaxSim within 1e-4.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from oracle import c_oracle as CO  # noqa: E402
from oracle import thr_oracle as O  # noqa: E402

D = 768


@pytest.fixture(scope="module")
def T():
    import triple_hybrid_rag_amd as T
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    T._native.load()
    return T


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def assert_lists_equal(S, I, cnt, Se, Ie, what):
    S, I, cnt = S.cpu().numpy(), I.cpu().numpy(), cnt.cpu().numpy()
    for q in range(len(Ie)):
        n = len(Ie[q])
        assert int(cnt[q]) == n, f"{what} q{q}: count {cnt[q]} != {n}"
        assert np.array_equal(I[q, :n], Ie[q]), f"{what} q{q}: ids differ"
        assert np.array_equal(S[q, :n], Se[q]), f"{what} q{q}: scores differ (bits)"


def mixed_lexical_queries(synth, nq, df, n_docs):
    """Half SURVEY 8(d)'s mix (4 terms sampled in proportion to df: stop words, lists of up to
    ~n_docs postings, the sliced path of thr_bm25_topk), half the same draw with the terms held
    by more than 1 % of the docs removed (short lists: one work item per query)."""
    a = synth.lexical_queries(nq, df, 4)
    dfq = df.copy()
    dfq[dfq > 0.01 * n_docs] = 0
    b = synth.lexical_queries(nq, dfq, 4)
    a[1::2] = b[1::2]
    return a


def test_config2_1m_dense_bm25_rrf(T):
    """BASELINE configs[2]: 1M x 768 dense + BM25 (avg 64 postings/term) + RRF(0.8/0.7), 64
    queries, both channels and the fused top-10 against the oracle; plus thr_doc_norms at 1M rows."""
    from triple_hybrid_rag_amd import synth
    n, nq = 1_000_000, 64
    x = synth.dense_rows(0, n, D)
    q = synth.dense_queries(nq, D, n)
    d_, t_, f_ = synth.lexical_rows(0, n, n)
    v = synth.vocab_size(n)
    csr = synth.build_lexical_csr(d_, t_, f_, n, v)
    idf = O.bm25_idf(n, csr.df_local)
    avgdl = csr.sum_dl_local / n
    qt = mixed_lexical_queries(synth, nq, csr.df_local, n)
    # SURVEY 8f.1 at full size: the index is BUILT ON THE DEVICE from the 32M (doc, term, tf) rows
    # (thr_lexical_build) -- rows to a searchable lexical index, bounds / impacts / dense-term rows
    # included, in well under a minute, its arrays synth.build_lexical_csr's
    import time
    rng = np.random.default_rng(1)
    perm = rng.permutation(len(d_))
    rows = [dev(a[perm].astype(np.int32)) for a in (d_, t_, f_)]
    del d_, t_, f_, perm
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    idx = T.GpuIndex().set_dense(x).set_lexical_rows(*rows, v, n_docs=n)
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t0
    assert build_s < 60.0, build_s
    del rows
    L = idx.lex
    for got, want in ((L["rowptr"], csr.rowptr), (L["post_doc"], csr.post_doc), (L["post_tf"], csr.post_tf),
                      (L["doclen"], csr.doclen), (L["idf"], idf)):
        assert np.array_equal(got.cpu().numpy(), want)
    assert L["avgdl"] == avgdl
    # the index's norms are the oracle's, bit for bit, on all 1M rows (the oracle below gets its own)
    dn = CO.doc_norms(x)
    assert np.array_equal(idx.dnorm.cpu().numpy(), dn)
    w = {"lexical": 0.7, "semantic": 0.8}
    res = idx.retrieve_batch(dev(q), dev(qt), None, top_k=10, weights=w)
    assert int(res.rescued) == 0
    Sd, Id = O.dense_topk_fast(x, q, 100, dnorm=dn)
    Sl, Il = O.bm25_topk(csr.rowptr, csr.post_doc, csr.post_tf, csr.doclen, idf, avgdl, qt, n, 50)
    assert_lists_equal(*res.channels["semantic"], Sd, Id, "dense")
    assert_lists_equal(*res.channels["lexical"], Sl, Il, "bm25")
    sub = [0, 1, 30, 63]   # the exhaustive float64 scan of the C oracle on a few of them
    Se, Ie, _ = CO.dense_topk_exact(x, q[sub], 100, dnorm=dn)
    for j, qi in enumerate(sub):
        assert np.array_equal(Ie[j], Id[qi]) and np.array_equal(Se[j], Sd[qi])
    ids, sc, cnt = res.ids.cpu().numpy(), res.scores.cpu().numpy(), res.counts.cpu().numpy()
    for i in range(nq):
        ei, es = O.fused_topk_ids(list(Il[i]), list(Id[i]), None, 10, w)
        assert list(ids[i, :cnt[i]]) == ei and list(sc[i, :cnt[i]]) == es, f"fused differs for query {i}"
    # the stop-word half really took the sliced path: lists of >= 24576 postings
    long_q = [i for i in range(nq) if sum(int(csr.df_local[t]) for t in qt[i] if t >= 0) >= 24576]
    assert len(long_q) >= nq // 4


_GLOBAL = {}


def global_lexical_stats(synth, n_global):
    if n_global not in _GLOBAL:
        _GLOBAL[n_global] = synth.lexical_global_stats(n_global)
    return _GLOBAL[n_global]


@pytest.mark.parametrize("rank", [0, 7])
def test_config3_config4_one_shard_of_the_10m_corpus(T, rank):
    """BASELINE configs[3] / configs[4]: rank ``rank`` of 8 of the 10M-doc corpus -- 1.25M rows,
    doc_base = lo, idf / avgdl of the WHOLE corpus, the replicated entity graph with this shard's
    mentions, late-interaction tokens generated on the device.  Per-shard top-k of every channel,
    the fusion of those lists and the shard's MaxSim scores against the oracle on the same slice."""
    from triple_hybrid_rag_amd import synth
    from triple_hybrid_rag_amd.distributed import shard_range
    n_global, world, nq = 10_000_000, 8, 48
    lo, hi = shard_range(n_global, rank, world)
    n = hi - lo
    assert n == 1_250_000
    x = synth.dense_rows(lo, n, D)
    q = synth.dense_queries(nq, D, n_global)
    df, sum_dl = global_lexical_stats(synth, n_global)
    v = synth.vocab_size(n_global)
    idf = O.bm25_idf(n_global, df)
    avgdl = sum_dl / n_global
    d_, t_, f_ = synth.lexical_rows(lo, n, n_global)
    csr = synth.build_lexical_csr(d_, t_, f_, n, v)
    del d_, t_, f_
    qt = mixed_lexical_queries(synth, nq, df, n_global)
    g = synth.build_graph(n_global, lo, hi)
    seeds = synth.graph_queries(nq, n_global, 3)
    qtok = synth.query_tokens(nq)
    idx = (T.GpuIndex(doc_base=lo).set_dense(x)
           .set_lexical(csr.rowptr, csr.post_doc, csr.post_tf, csr.doclen, idf, avgdl)
           .set_graph(g.ent_rowptr, g.ent_col, g.men_rowptr, g.men_chunk, g.men_conf)
           .set_tokens(synth.device_tokens(lo, n)))
    res = idx.retrieve_batch(dev(q), dev(qt), dev(seeds), top_k=10)
    assert int(res.rescued) == 0
    dn = CO.doc_norms(x)
    Sd, Id = O.dense_topk_fast(x, q, 100, doc_id_base=lo, dnorm=dn)
    Sl, Il = O.bm25_topk(csr.rowptr, csr.post_doc, csr.post_tf, csr.doclen, idf, avgdl, qt, n, 50,
                         doc_id_base=lo)
    Sg, Ig = O.graph_topk(g.ent_rowptr, g.ent_col, g.men_rowptr, g.men_chunk, g.men_conf, seeds, 2, n,
                          50, chunk_base=lo)
    assert_lists_equal(*res.channels["semantic"], Sd, Id, "dense")
    assert_lists_equal(*res.channels["lexical"], Sl, Il, "bm25")
    assert_lists_equal(*res.channels["graph"], Sg, Ig, "graph")
    assert all(lo <= i < hi for row in Id for i in row)
    Se, Ie, _ = CO.dense_topk_exact(x, q[:2], 100, doc_id_base=lo, dnorm=dn)
    for j in range(2):
        assert np.array_equal(Ie[j], Id[j]) and np.array_equal(Se[j], Sd[j])
    ids, sc, cnt = res.ids.cpu().numpy(), res.scores.cpu().numpy(), res.counts.cpu().numpy()
    for i in range(nq):
        ei, es = O.fused_topk_ids(list(Il[i]), list(Id[i]), list(Ig[i]), 10)
        assert list(ids[i, :cnt[i]]) == ei and list(sc[i, :cnt[i]]) == es, f"fused differs for query {i}"
    # configs[4]: the shard's MaxSim scores of the fused top-100 and the reranked order
    f100 = [O.fused_topk_ids(list(Il[i]), list(Id[i]), list(Ig[i]), 100)[0] for i in range(nq)]
    width = max(len(f) for f in f100)
    cand = np.full((nq, width), -1, dtype=np.int64)
    for i, f in enumerate(f100):
        cand[i, :len(f)] = f
    got = idx.maxsim(dev(qtok), dev(cand)).cpu().numpy().astype(np.float64)
    res4 = idx.retrieve_batch(dev(q), dev(qt), dev(seeds), top_k=10, qtok=dev(qtok), rerank_top_k=100)
    ids4 = res4.ids.cpu().numpy()
    block = 8192
    for i in range(0, nq, 6):   # the oracle's MaxSim on 8 queries: their candidates' token rows are
        rows = {}               # regenerated block by block on the device and copied back to the host
        for b in sorted({d // block for d in f100[i]}):   # (blocks are seeded by the GLOBAL block index)
            blk = synth.device_tokens(b * block, block)
            for d in f100[i]:
                if d // block == b:
                    rows[d] = blk[d - b * block].cpu().numpy()
            del blk
        dt = np.stack([rows[d] for d in f100[i]])
        exp = CO.maxsim(qtok[i:i + 1], dt, np.arange(len(f100[i]), dtype=np.int64)[None])[0]
        assert np.max(np.abs(got[i, :len(f100[i])] - exp)) < 1e-4
        assert np.all(got[i, len(f100[i]):] == -np.inf)
        order = O.rerank_order([float(np.float32(s)) for s in exp])[:10]
        es = np.array([exp[p] for p in order])
        for a in range(len(order)):   # identical wherever the oracle's scores are > 2e-4 apart
            if (a == 0 or es[a - 1] - es[a] > 2e-4) and (a == len(order) - 1 or es[a] - es[a + 1] > 2e-4):
                assert int(ids4[i][a]) == f100[i][order[a]]


def test_config3_dense_channel_eight_shards_with_the_shard_floor(T):
    """BASELINE configs[3]'s dense channel at its full size: 10M x 768 as the 8 document shards of
    1.25M rows, all resident on this GPU.  The shard-floor path (thr_dense_shortlist_f16 -> the
    gathered bounds -> thr_dense_finish_f16 -> thr_merge_topk) against every shard ranking a
    top-100 of its own: the same bits for every query; against the oracle (per-shard fast path,
    merged on the host under (score desc, id asc)) for the first 16; a shard rescores ~k / 8 rows."""
    from triple_hybrid_rag_amd import synth
    from triple_hybrid_rag_amd.distributed import shard_range
    n_global, world, nq, k, n_or = 10_000_000, 8, 256, 100, 16
    q = synth.dense_queries(nq, D, n_global)
    shards, oracle = [], []
    for r in range(world):
        lo, hi = shard_range(n_global, r, world)
        x = synth.dense_rows(lo, hi - lo, D)
        shards.append(T.GpuIndex(doc_base=lo).set_dense(x))
        assert shards[-1].shortlist == "f16"
        oracle.append(O.dense_topk_fast(x, q[:n_or], k, doc_id_base=lo))
        del x
    qd = dev(q)
    classic = [ix.dense_search(qd, k, sync=False) for ix in shards]
    S0, I0, _ = T._native.merge_topk(torch.stack([c[0] for c in classic]), torch.stack([c[1] for c in classic]), k)
    lbs = torch.stack([ix.dense_shortlist(qd, k, world) for ix in shards])
    outs = []
    for ix in shards:
        ix.dense_shortlist(qd, k, world)      # (a shard's finish reads the lists ITS shortlist call left)
        outs.append(ix.dense_finish(qd, k, lb_all=lbs))
    S1, I1, c1 = T._native.merge_topk(torch.stack([o[0] for o in outs]), torch.stack([o[1] for o in outs]), k)
    assert torch.equal(S0, S1) and torch.equal(I0, I1)
    assert sum(int(o[4]) for o in outs) == 0 and sum(int(c[3]) for c in classic) == 0
    cnts = torch.stack([o[2] for o in outs]).cpu().numpy()
    assert np.all(cnts.sum(0) >= k) and cnts.mean() < 2.0 * k / world, cnts.mean()
    assert all(bool(torch.all(o[3] & 1)) for o in outs), "every shard's list certified by the shortlist path"
    S1, I1 = S1.cpu().numpy(), I1.cpu().numpy()
    for i in range(n_or):
        pairs = sorted(((-float(s), int(d)) for Sd, Id in oracle for s, d in zip(Sd[i], Id[i])))[:k]
        assert [d for _, d in pairs] == list(I1[i]), f"query {i}: ids differ from the oracle's merged top-{k}"
        assert [-s for s, _ in pairs] == list(S1[i]), f"query {i}: scores differ (bits)"
