"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle.

Bars (BASELINE.json north_star): ids / rank order bit-exact; cosine and BM25
scores within 1e-5 (the float64 rescoring in fact reproduces the oracle's bits,
which is what is asserted); MaxSim within 1e-4 absolute (fp32 MFMA accumulate).
"""
import os
import numpy as np
import torch
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from oracle import c_oracle as CO  # noqa: E402
from oracle import thr_oracle as O  # noqa: E402


@pytest.fixture(scope="module")
def T():
    import triple_hybrid_rag_amd as T
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    T._native.load()
    return T


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return t if dtype is None else t.to(dtype)


def rand_docs(n, d, seed):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n, d)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return x, rng


def assert_topk_equal(S, I, cnt, Se, Ie, cnte, what=""):
    S, I, cnt = S.cpu().numpy(), I.cpu().numpy(), cnt.cpu().numpy()
    for q in range(len(cnte)):
        n = int(cnte[q])
        assert int(cnt[q]) == n, f"{what} q{q}: count {cnt[q]} != {n}"
        assert np.array_equal(I[q, :n], Ie[q][:n]), f"{what} q{q}: ids differ"
        assert np.array_equal(S[q, :n], Se[q][:n]), f"{what} q{q}: scores differ (bits)"
        assert np.all(I[q, n:] == -1)


def test_library_and_device(T):
    cus, hbm, arch = T._native.device_info()
    assert arch.startswith("gfx950"), arch
    assert cus >= 64 and hbm > 64 * 2 ** 30
    import os
    assert os.path.exists(T._native.lib_path())


def test_doc_norms_and_embed_postproc(T):
    x, rng = rand_docs(3000, 768, 1)
    x[7] = 0
    x[8] *= 3.5
    dn, inv = T._native.doc_norms(dev(x))
    assert np.array_equal(dn.cpu().numpy(), O.doc_norms_f64(x))
    assert inv[7].item() == 0.0
    full = rng.standard_normal((257, 1024)).astype(np.float32)
    full[3] = 0
    for store in (1024, 768, 100, 2048):
        got = T._native.embed_postproc(dev(full), store).cpu().numpy()
        exp = O.embed_postproc_batch(full, store)
        assert got.shape == exp.shape
        assert np.allclose(got, exp, rtol=3e-7, atol=1e-9)
    # the reference's own known answer: normalize_l2([3,4])
    got = T._native.embed_postproc(dev(np.array([[3.0, 4.0]], dtype=np.float32)), 1024).cpu().numpy()
    assert got.tolist() == [[0.6000000238418579, 0.800000011920929]]


@pytest.mark.parametrize("n,d,nq,k", [(5000, 768, 37, 100), (3000, 256, 5, 10), (2500, 1024, 64, 100),
                                      (777, 512, 33, 50), (60, 768, 3, 100)])
def test_dense_small_exhaustive_tau(T, n, d, nq, k):
    """n <= 8192: no sampling pass, every row is a candidate (tau = -inf)."""
    x, rng = rand_docs(n, d, n + d)
    q = rng.standard_normal((nq, d)).astype(np.float32)
    q[0] = x[11] + 0.1 * q[0]
    idx = T.GpuIndex().set_dense(x)
    S, I, cnt, nres = idx.dense_search(dev(q), k)
    Se, Ie, cnte = CO.dense_topk_exact(x, q, k)
    assert_topk_equal(S, I, cnt, Se, Ie, cnte, "dense-small")


@pytest.mark.parametrize("n,d", [(50000, 768), (131072 + 5, 768), (40001, 1024)])
def test_dense_sampled_path(T, n, d):
    x, rng = rand_docs(n, d, 5)
    q = rng.standard_normal((70, d)).astype(np.float32)
    q[::2] = x[rng.integers(0, n, 35)] + 0.5 * q[::2]
    idx = T.GpuIndex(doc_base=1_000_000).set_dense(x)
    qd = dev(q)
    S, I, cnt, flg = T._native.dense_topk(idx.docs, idx.dnorm, idx.inv_norm, qd, 100, 128, 1_000_000)
    flags = flg.cpu().numpy()
    assert np.all(flags & 1), "random data must certify without the exhaustive path"
    Se, Ie, cnte = CO.dense_topk_exact(x, q, 100, doc_id_base=1_000_000)
    assert_topk_equal(S, I, cnt, Se, Ie, cnte, "dense-sampled")
    # cosine within 1e-5 of an independent float64 BLAS evaluation (north-star tolerance)
    ref = (x[Ie[3] - 1_000_000].astype(np.float64) @ q[3].astype(np.float64)) / (
        np.linalg.norm(q[3].astype(np.float64)) * np.linalg.norm(x[Ie[3] - 1_000_000].astype(np.float64), axis=1))
    assert np.max(np.abs(ref - S[3].cpu().numpy())) < 1e-5


def test_dense_ties_duplicates_null_rows_need_rescue(T):
    x, rng = rand_docs(30000, 768, 9)
    x[100:1700] = x[99]         # 1601 exact duplicates: top-100 is one giant tie
    x[5000] = 0                 # NULL embedding
    x[6000:6004] = 0
    x[7000:7300] = x[6999]      # 301 duplicates: too many for the first 256-row band, but the
    q = rng.standard_normal((6, 768)).astype(np.float32)   # 1024-row second chance holds them all
    q[0] = x[99] * 2.0
    q[1] = x[99] + 0.01 * q[1]
    q[2] = 0                    # zero query: every similarity is 0 -> first ids win
    q[3] = x[6999] * 0.5
    idx = T.GpuIndex().set_dense(x)
    S, I, cnt, flg = T._native.dense_topk(idx.docs, idx.dnorm, idx.inv_norm, dev(q), 100, 128, 0)
    flg = flg.cpu().numpy()
    assert (flg[0] & 1) == 0, "a 1601-way tie cannot be certified from a 1024-row band"
    assert (flg[3] & 1) == 1, "a 301-way tie is settled exactly once every tied row is rescored"
    assert list(I[3, :100].cpu().numpy()) == list(range(6999, 7099))
    S, I, cnt, nres = idx.dense_search(dev(q), 100)
    assert nres >= 1
    Se, Ie, cnte = CO.dense_topk_exact(x, q, 100)
    assert_topk_equal(S, I, cnt, Se, Ie, cnte, "dense-ties")
    assert list(I[0, :100].cpu().numpy()) == list(range(99, 199))
    assert 5000 not in set(I.cpu().numpy().ravel())


@pytest.mark.parametrize("shortlist", ["f32", "f16", "f16-inline"])
def test_dense_candidate_list_overflow_goes_to_rescue(T, shortlist):
    """Forces THR_FLAG_OVERFLOW: 60 queries sit on a 20 000-row duplicate cluster, so each of them
    has more rows >= tau than CAND_CAP (16 384) and every query tile's shared list (qtile * 8192
    entries: 32 / 64 / 96 queries per tile) overflows while the ordinary queries of the same tile
    have short lists.  The workspace is prefilled with 0xFF: round 1's select_rescore read all
    CAND_CAP slots of such an ordinary query and used the stale words as row indices (the abort of
    gpurun_out/t1.log); it must read only the slots that were written.  After thr_dense_rescue the
    batch equals the oracle."""
    n, d, nq, ndup = 60000, 768, 96, 20000
    x, rng = rand_docs(n, d, 17)
    x[30000:30000 + ndup] = x[29999]
    q = rng.standard_normal((nq, d)).astype(np.float32)
    q[:60] = x[29999] + 1e-3 * q[:60]
    q[60:80] = x[rng.integers(0, 29999, 20)] + 0.5 * q[60:80]
    idx = T.GpuIndex().set_dense(x, shortlist=shortlist)
    idx.reserve(nq, 100)
    idx._ws.fill_(0xFF)
    qd = dev(q)
    if shortlist == "f32":
        S, I, cnt, flg = T._native.dense_topk(idx.docs, idx.dnorm, idx.inv_norm, qd, 100, 128, 0, idx._ws)
    else:
        S, I, cnt, flg = T._native.dense_topk_f16(idx.docs, idx.docs16, idx.doc_rel_err, idx.dnorm,
                                                  idx.inv_norm, qd, 100, 192, 0, idx._ws)
    flags = flg.cpu().numpy()
    assert np.all(flags[:60] & T._native.THR_FLAG_OVERFLOW), "cluster queries overflow CAND_CAP"
    assert np.all((flags[:60] & 1) == 0)
    if shortlist != "f16":   # (the default scan keeps one list per query: nothing shared to overflow)
        assert np.any(flags[60:] & T._native.THR_FLAG_OVERFLOW), "their tile's list overflowed too"
    assert np.all((flags[60:][(flags[60:] & T._native.THR_FLAG_OVERFLOW) != 0] & 1) == 0)
    idx._ws.fill_(0xFF)
    S, I, cnt, nres = idx.dense_search(qd, 100)
    assert nres >= 60
    Se, Ie, cnte = CO.dense_topk_exact(x, q, 100)
    assert_topk_equal(S, I, cnt, Se, Ie, cnte, "dense-overflow")
    assert list(I[0, :100].cpu().numpy()) == list(range(29999, 30099))


@pytest.mark.parametrize("shortlist", ["f32", "f16", "f16-inline"])
def test_dense_matches_reference_python_cosine(T, golden, shortlist):
    """a2 pinned by the reference's own Python (tests/golden/dense_cosine.json, generated from
    MultimodalEmbedder.cosine_similarity, core/embedder.py:316-331, and
    HybridSearcher._vector_search_fallback, hybrid_search.py:260-320): the same rows and queries
    through thr_dense_topk / thr_dense_topk_f16.  Scores within 1e-5 (north star); identical
    order wherever the reference's own scores are more than 1e-5 apart."""
    from conftest import decode_array
    g = golden("dense_cosine.json")
    rows, q = decode_array(g["rows"]), decode_array(g["queries"])
    ref = np.array(g["cosine_similarity"], dtype=np.float64)
    null = set(g["null_rows"])
    n = rows.shape[0]
    idx = T.GpuIndex().set_dense(rows, shortlist=shortlist)
    S, I, cnt, _ = idx.dense_search(dev(q), 100)
    S, I, cnt = S.cpu().numpy(), I.cpu().numpy(), cnt.cpu().numpy()
    for i in range(q.shape[0]):
        c = int(cnt[i])
        assert c == 100 and not (set(I[i, :c].tolist()) & null)
        assert np.max(np.abs(S[i, :c] - ref[i][I[i, :c]])) < 1e-5
        if not q[i].any():
            continue
        live = np.array([j for j in range(n) if j not in null])
        order = live[np.lexsort((live, -ref[i][live]))][:100]    # the reference's scores, our tie rule
        sep = np.abs(np.diff(ref[i][order])) > 1e-5
        for a in range(100):
            if (a == 0 or sep[a - 1]) and (a == 99 or sep[a]):
                assert I[i, a] == order[a]
    # the fallback scorer's table: unit rows, NULL embeddings, optional category filter
    unit, qunit = decode_array(g["unit_rows"]), decode_array(g["unit_queries"])
    unit[sorted(null)] = 0                      # NULL embedding = zero row = excluded
    idx = T.GpuIndex().set_dense(unit, shortlist=shortlist)
    S, I, cnt, _ = idx.dense_search(dev(qunit), 60)
    S, I = S.cpu().numpy(), I.cpu().numpy()
    for c in g["fallback"]:
        if c["category"] is not None or not qunit[c["query"]].any():
            continue
        ref_ids = [int(o["chunk_id"][1:]) for o in c["out"]]
        ref_s = np.array([o["similarity_score"] for o in c["out"]])
        got = I[c["query"], :60].tolist()
        assert np.max(np.abs(S[c["query"], :60] - ref_s)) < 1e-5
        gaps = np.abs(np.diff(ref_s))
        for a in range(60):
            if (a == 0 or gaps[a - 1] > 1e-5) and (a == 59 or gaps[a] > 1e-5):
                assert got[a] == ref_ids[a]
        assert sorted(got) == sorted(ref_ids) or np.min(gaps[58:]) <= 1e-5


def test_dense_f16_batch_limit_and_chunking(T, monkeypatch):
    """The copy scan addresses candidate segments with 32-bit byte offsets: one call takes at
    most thr_dense_f16_max_queries queries (a larger one is refused, not wrapped around), and
    GpuIndex.dense_search splits larger batches -- same bits as the unsplit call."""
    import ctypes as C
    x, rng = rand_docs(30000, 768, 77)
    q = rng.standard_normal((700, 768)).astype(np.float32)
    idx = T.GpuIndex().set_dense(x, shortlist="f16")
    lim = T._native.dense_f16_max_queries(768, True)
    assert lim == 32512
    lib = T._native.load()
    qd = dev(q)
    ws = torch.empty(1024, dtype=torch.uint8, device="cuda")
    o = torch.empty(16, dtype=torch.int64, device="cuda")
    rc = lib.thr_dense_topk_f16(idx.docs.data_ptr(), idx.docs16.data_ptr(), float(idx.doc_rel_err),
                                idx.dnorm.data_ptr(), idx.inv_norm.data_ptr(), 30000, 768, 0, qd.data_ptr(),
                                lim + 1, 10, 100, None, None, o.data_ptr(), o.data_ptr(), o.data_ptr(),
                                o.data_ptr(), ws.data_ptr(), 1024, None)
    assert rc == -2, rc   # THR_ERR_UNSUPPORTED, before anything is launched
    S, I, cnt, nres = idx.dense_search(qd, 100)
    monkeypatch.setattr(idx, "max_batch", lambda: 256)
    S2, I2, cnt2, nres2 = idx.dense_search(qd, 100)
    assert torch.equal(S, S2) and torch.equal(I, I2) and torch.equal(cnt, cnt2) and nres == nres2 == 0
    S3, I3, cnt3, nres3 = idx.dense_search(qd, 100, sync=False)
    assert torch.equal(I, I3) and int(nres3) == 0


def test_dense_exact_path_alone(T):
    x, rng = rand_docs(20000, 256, 21)
    q = rng.standard_normal((9, 256)).astype(np.float32)
    idx = T.GpuIndex(doc_base=7).set_dense(x)
    S, I, cnt, flg = T._native.dense_topk_exact(idx.docs, idx.dnorm, dev(q), 50, 7)
    Se, Ie, cnte = CO.dense_topk_exact(x, q, 50, doc_id_base=7)
    assert_topk_equal(S, I, cnt, Se, Ie, cnte, "dense-exact")


def lexical_fixture(T, n):
    from triple_hybrid_rag_amd import synth
    v = synth.vocab_size(n)
    doc, term, tf = synth.lexical_rows(0, n, n)
    csr = synth.build_lexical_csr(doc, term, tf, n, v)
    idf = O.bm25_idf(n, csr.df_local)
    avgdl = csr.sum_dl_local / n
    return csr, idf, avgdl, v


def test_bm25_matches_oracle(T):
    n = 30000
    csr, idf, avgdl, v = lexical_fixture(T, n)
    rng = np.random.default_rng(4)
    qt = rng.integers(0, v, size=(48, 6)).astype(np.int32)
    qt[0] = [0, 1, 2, 3, 4, 5]            # the hottest terms: lists far beyond the LDS stage
    qt[1] = [v - 1, v - 2, -1, -1, -1, -1]  # rare terms, padding
    qt[2] = [7, 7, 9, 7, -1, 9]           # repeated ids count once per occurrence
    qt[3] = -1                             # empty query
    qt[4] = [0, -1, -1, -1, -1, -1]
    empty = np.nonzero(csr.df_local == 0)[0]
    if len(empty):
        qt[5] = [empty[0], -1, -1, -1, -1, -1]
    idx = T.GpuIndex(doc_base=500).set_lexical(csr.rowptr, csr.post_doc, csr.post_tf, csr.doclen,
                                               idf, avgdl)
    S, I, cnt = idx.bm25_search(dev(qt), 50)
    Se, Ie = O.bm25_topk(csr.rowptr, csr.post_doc, csr.post_tf, csr.doclen, idf, avgdl, qt, n, 50,
                         doc_id_base=500)
    assert_topk_equal(S, I, cnt, Se, Ie, [len(s) for s in Se], "bm25")
    assert int(cnt[3]) == 0


@pytest.mark.parametrize("knob", ["THR_DENSE_MFMA=32", "THR_DENSE_F16=q", "THR_DENSE_QW=32"])
def test_dense_alternate_scan_builds_in_a_subprocess(knob):
    """The documented A/B knobs of the default scan are read once per process:
    THR_DENSE_MFMA=32 (the 32x32x16 build of the staggered kernel, with its own copy / query
    image layouts), THR_DENSE_F16=q (the two-blocks-per-CU kernel at dim <= 768) and
    THR_DENSE_QW=32 (dim 1024 with 32 queries per wave and the emit threaded through the next
    tile's MFMAs, instead of 48 per wave).  The f16 parity tests run again under each."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    name, value = knob.split("=")
    if os.environ.get(name) == value:
        pytest.skip("already inside that run")
    out = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", os.path.abspath(__file__), "-k",
                          "test_dense_f16_shortlist_is_still_exact or test_dense_collection_filter_before_topk "
                          "or test_dense_candidate_list_overflow_goes_to_rescue"],
                         env=dict(os.environ, **{name: value}), cwd=root, capture_output=True, text=True,
                         timeout=1200)
    assert out.returncode == 0 and " passed" in out.stdout and "failed" not in out.stdout, \
        out.stdout[-2000:] + out.stderr[-1000:]


def test_bm25_small_block_shape_in_a_subprocess():
    """THR_BM25_SHAPE=small (256 threads / 4096 staged ids, four queries per CU) is read once per
    process: the two BM25 tests of this file run again under it."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if os.environ.get("THR_BM25_SHAPE") == "small":
        pytest.skip("already inside the small-shape run")
    out = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", os.path.abspath(__file__), "-k",
                          "test_bm25_matches_oracle or test_bm25_pruning_and_filters_stay_exact"],
                         env=dict(os.environ, THR_BM25_SHAPE="small"), cwd=root, capture_output=True,
                         text=True, timeout=900)
    assert out.returncode == 0 and "2 passed" in out.stdout, out.stdout[-2000:] + out.stderr[-1000:]


@pytest.mark.parametrize("knobs", ["THR_BM25_DENSE=0", "THR_BM25_WALK_DIV=8 THR_BM25_ITEMS=1024 THR_BM25_FUSE_DIV=0",
                                   "THR_BM25_SHAPE=small THR_BM25_DENSE_SHARE=0.05 THR_BM25_FUSE_DIV=1000000",
                                   # the workgroup walk for every query (round 3's path; the default since
                                   # round 4 gives OR queries of <= 8 terms to bm25_walk_wave_kernel), its
                                   # stage A on a launch of its own / fused with the ordinary items
                                   "THR_BM25_WALK=block THR_BM25_FUSE_DIV=0", "THR_BM25_WALK=block"])
def test_bm25_dense_row_knobs_in_a_subprocess(knobs):
    """The A/B knobs around the dense-term rows change the cost, never the result: without the rows
    in the kernels (THR_BM25_DENSE=0), with nearly every row term walked and the fewest work items
    (coarse slices, the sweep filter deciding on other thresholds) and stage A always a launch of
    its own, with the small block shape, another share and stage A always fused with the ordinary
    items' launch -- the BM25 parity tests run again under each (read once per process)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    if env.get("THR_BM25_KNOB_RUN"):
        pytest.skip("already inside a knob run")
    env["THR_BM25_KNOB_RUN"] = "1"
    for kv in knobs.split():
        name, value = kv.split("=")
        env[name] = value
    out = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", os.path.abspath(__file__), "-k",
                          "test_bm25_matches_oracle or test_bm25_pruning_and_filters_stay_exact or "
                          "test_bm25_dense_term_rows_equal_the_posting_walk"],
                         env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "3 passed" in out.stdout, out.stdout[-2000:] + out.stderr[-1000:]


def test_bm25_pruning_and_filters_stay_exact(T):
    """The WAND-style pruning (term / block score bounds), the conjunctive mode, the collection
    filter and an out-of-vocabulary term id: each equals the oracle bit for bit -- also on
    queries made of the longest lists (stop words), where almost every doc is dropped on its
    bound."""
    from triple_hybrid_rag_amd import synth
    n = 60000
    csr, idf, avgdl, v = lexical_fixture(T, n)
    idx = T.GpuIndex().set_lexical(csr.rowptr, csr.post_doc, csr.post_tf, csr.doclen, idf, avgdl)
    coll = (np.arange(n) * 7919 % 50).astype(np.int32)          # 50 collections of 2 % each
    coll[n // 2:] = np.where(np.arange(n - n // 2) % 2 == 0, 60, coll[n // 2:])   # + one fat one (25 %)
    idx.set_collections(coll)
    qt = synth.lexical_queries(48, csr.df_local, 4)               # sampled by df: stop words included
    top = np.argsort(-csr.df_local)[:6].astype(np.int32)
    qt[0] = top[:4]                                               # the four longest lists
    qt[1] = [top[0], top[1], top[0], -1]                          # a repeated term, padding
    qt[2] = [top[2], v + 5, 2 ** 30, -1]                          # ids outside the vocabulary: ignored
    qt[3] = [-1, -1, -1, -1]                                      # no term at all
    qd = dev(qt)
    Se, Ie = O.bm25_topk(csr.rowptr, csr.post_doc, csr.post_tf, csr.doclen, idf, avgdl, qt, n, 50)
    for prune in (True, False):
        S, I, cnt = idx.bm25_search(qd, 50, prune=prune)
        assert_topk_equal(S, I, cnt, Se, Ie, [len(s) for s in Se], f"bm25 prune={prune}")
    # the per-term / per-block bounds alone (no per-posting impacts): the mask path with a threshold
    L = idx.lex
    S, I, cnt = T._native.bm25_topk(L["rowptr"], L["post_doc"], L["post_tf"], L["doclen"], L["idf"], L["avgdl"],
                                    qd, 50, bounds=L["bounds"][:2])
    assert_topk_equal(S, I, cnt, Se, Ie, [len(s) for s in Se], "bm25 term/block bounds only")
    # every quantised impact bounds its posting's contribution / idf from above, within 2 steps
    imp = L["bounds"][2].cpu().numpy().astype(np.float64)[:len(csr.post_tf)]   # (4 bytes of padding behind)
    tf_ = csr.post_tf.astype(np.float64)
    nrm = 1.2 * ((1.0 - 0.75) + 0.75 * (csr.doclen[csr.post_doc].astype(np.float64) / avgdl))
    true_imp = tf_ * 2.2 / (tf_ + nrm)
    assert np.all(imp * (2.2 / 255.0) >= true_imp) and np.all(imp * (2.2 / 255.0) <= true_imp + 2.01 * 2.2 / 255.0)
    Sa, Ia = O.bm25_topk(csr.rowptr, csr.post_doc, csr.post_tf, csr.doclen, idf, avgdl, qt, n, 50,
                         conjunctive=True)
    S, I, cnt = idx.bm25_search(qd, 50, conjunctive=True)
    assert_topk_equal(S, I, cnt, Sa, Ia, [len(s) for s in Sa], "bm25 AND")
    assert len(Ia[0]) > 0 and len(Ia[0]) <= len(Ie[0])
    # an id outside the vocabulary: ignored by the OR form, unsatisfiable in the AND form
    assert len(Ie[2]) == 50 and len(Ia[2]) == 0 and int(cnt[2]) == 0
    qc = np.full(48, -1, dtype=np.int32)
    qc[0::3], qc[1::3] = 7, 60                                    # thin, fat, unfiltered
    qc[5] = 12345                                                 # a collection no doc is in
    Sf, If = O.bm25_topk(csr.rowptr, csr.post_doc, csr.post_tf, csr.doclen, idf, avgdl, qt, n, 50,
                         doc_coll=coll, query_coll=qc)
    S, I, cnt = idx.bm25_search(qd, 50, collections=dev(qc))
    assert_topk_equal(S, I, cnt, Sf, If, [len(s) for s in Sf], "bm25 collection")
    assert len(If[5]) == 0 and all(coll[i] == 7 for i in If[0])


def test_bm25_dense_term_rows_equal_the_posting_walk(T):
    """Terms held by a large share of the docs get per-doc rows of impacts / term frequencies
    (thr_bm25_dense_rows) and their queries take the doc-window kernel: same bits as the posting
    walk and as the oracle -- at the default share, at a low one (most query terms dense), on an
    odd-sized shard with a doc base, with the collection filter, for k = 1 / 50 / 128, and with a
    list that is sparse overall but packed into one stretch of doc ids."""
    from triple_hybrid_rag_amd import synth
    n = 70001
    csr, idf, avgdl, v = lexical_fixture(T, n)
    # a list of 6000 postings packed into docs [30000, 36000): sparse by its share (8.6 %), locally dense
    rp, pd, pt = csr.rowptr.copy(), csr.post_doc.copy(), csr.post_tf.copy()
    packed = int(np.argsort(-csr.df_local)[60])
    lo, hi = int(rp[packed]), int(rp[packed + 1])
    docs = np.arange(30000, 36000, dtype=np.int32)
    pd = np.concatenate([pd[:lo], docs, pd[hi:]])
    pt = np.concatenate([pt[:lo], 1 + (docs % 3), pt[hi:]]).astype(np.int32)
    rp[packed + 1:] += len(docs) - (hi - lo)
    df = (rp[1:] - rp[:-1]).astype(np.int64)
    top = np.argsort(-df)[:12].astype(np.int32)
    qt = synth.lexical_queries(64, df, 4)
    qt[0] = top[:4]
    qt[1] = [top[0], packed, top[5], -1]
    qt[2] = [packed, -1, -1, -1]
    qt[3] = [top[1], top[1], v + 3, -1]          # repeated dense term, an id outside the vocabulary
    qt[4] = [int(np.argsort(-df)[5000]), top[2], -1, top[3]]
    qt[5] = -1
    coll = (np.arange(n) % 9).astype(np.int32)
    qc = np.full(64, -1, dtype=np.int32)
    qc[::4] = 3
    for share in (0.125, 0.01):
        idx = T.GpuIndex(doc_base=1000).set_lexical(rp, pd, pt, csr.doclen, idf, avgdl, dense_share=share)
        idx.set_collections(coll)
        dense = idx.lex["dense"]
        assert dense is not None and dense[1].shape[0] == min(512, int((df >= share * n).sum()))
        assert (int(dense[0][packed]) >= 0) == (share < 0.08)
        # the rows are the postings, scattered: impact and tf of a few docs of the longest list
        t0 = int(top[0])
        row = int(dense[0][t0])
        a, b = int(rp[t0]), int(rp[t0 + 1])
        got_tf = dense[2][row].cpu().numpy().astype(np.int64)[:n]
        exp_tf = np.zeros(n, dtype=np.int64)
        exp_tf[pd[a:b]] = pt[a:b]
        assert np.array_equal(got_tf, exp_tf)
        imp = idx.lex["bounds"][2].cpu().numpy()
        exp_imp = np.zeros(n, dtype=np.uint8)
        exp_imp[pd[a:b]] = imp[a:b]
        assert np.array_equal(dense[1][row].cpu().numpy()[:n], exp_imp)
        assert not dense[1][row][n:].any()          # zero padding behind the shard's docs
        for k in (1, 50, 128):
            Se, Ie = O.bm25_topk(rp, pd, pt, csr.doclen, idf, avgdl, qt, n, k, doc_id_base=1000)
            S, I, cnt = idx.bm25_search(dev(qt), k)
            assert_topk_equal(S, I, cnt, Se, Ie, [len(s) for s in Se], f"bm25 dense rows share={share} k={k}")
            S2, I2, cnt2 = idx.bm25_search(dev(qt), k, dense_rows=False)
            assert torch.equal(S, S2) and torch.equal(I, I2) and torch.equal(cnt, cnt2)
        Sf, If = O.bm25_topk(rp, pd, pt, csr.doclen, idf, avgdl, qt, n, 50, doc_id_base=1000,
                             doc_coll=coll, query_coll=qc)
        S, I, cnt = idx.bm25_search(dev(qt), 50, collections=dev(qc))
        assert_topk_equal(S, I, cnt, Sf, If, [len(s) for s in Sf], f"bm25 dense rows + collection share={share}")
        # one query alone (unsliced below 24576 work units, sliced above), and a 1-doc-wide last slice
        S1, I1, cnt1 = idx.bm25_search(dev(qt[:1]), 50)
        Se, Ie = O.bm25_topk(rp, pd, pt, csr.doclen, idf, avgdl, qt[:1], n, 50, doc_id_base=1000)
        assert_topk_equal(S1, I1, cnt1, Se, Ie, [len(s) for s in Se], "bm25 dense rows, one query")


def test_bm25_term_counts_k_and_tiny_corpora(T):
    """Every phase-1 form of the BM25 pass against the oracle: queries of 1 / 2 / 5 / 8 terms
    (accumulated impact bounds), 9 / 12 / 32 terms (32-bit doc masks), k = 1 / 128, stop words in
    every position, a 300-doc corpus (one workgroup, one pass) and a sliced one; with and without
    the bounds, OR and AND."""
    from triple_hybrid_rag_amd import synth
    for n in (300, 90000):
        csr, idf, avgdl, v = lexical_fixture(T, n)
        idx = T.GpuIndex().set_lexical(csr.rowptr, csr.post_doc, csr.post_tf, csr.doclen, idf, avgdl)
        top = np.argsort(-csr.df_local)[:40].astype(np.int32)     # the longest lists
        rng = np.random.default_rng(n)
        for nt in (1, 2, 5, 8, 9, 12, 32):
            qt = np.full((24, nt), -1, dtype=np.int32)
            for i in range(24):
                m = 1 + i % nt                                     # 1 .. nt terms, the rest padding
                pick = np.concatenate([top[rng.permutation(min(len(top), 8 + nt))[:(m + 1) // 2]],
                                       rng.integers(0, v, size=m)])[:m]
                qt[i, rng.permutation(nt)[:m]] = pick              # padding anywhere in the row
            for k in (1, 50, 128):
                Se, Ie = O.bm25_topk(csr.rowptr, csr.post_doc, csr.post_tf, csr.doclen, idf, avgdl, qt, n, k)
                for prune in (True, False):
                    S, I, cnt = idx.bm25_search(dev(qt), k, prune=prune)
                    assert_topk_equal(S, I, cnt, Se, Ie, [len(s) for s in Se], f"bm25 n={n} nt={nt} k={k} prune={prune}")
            Sa, Ia = O.bm25_topk(csr.rowptr, csr.post_doc, csr.post_tf, csr.doclen, idf, avgdl, qt, n, 20,
                                 conjunctive=True)
            S, I, cnt = idx.bm25_search(dev(qt), 20, conjunctive=True)
            assert_topk_equal(S, I, cnt, Sa, Ia, [len(s) for s in Sa], f"bm25 AND n={n} nt={nt}")


@pytest.mark.parametrize("shortlist", ["f32", "f16", "f16-inline"])
def test_dense_collection_filter_before_topk(T, shortlist):
    """p_collection is a WHERE clause (rag2_schema.sql:404-408): the limit best rows OF THE
    COLLECTION, however deep they sit in the unfiltered ranking.  A 2 % collection has ~0.4
    members among the unfiltered top-20 and ~1.6 among the top-80 round 1 over-fetched."""
    n, d, k = 60000, 768, 20
    x, rng = rand_docs(n, d, 23)
    x[77] = 0
    coll = (np.arange(n) * 7919 % 50).astype(np.int32)
    coll[n // 2:] = np.where(np.arange(n - n // 2) % 2 == 0, 60, coll[n // 2:])
    q = rng.standard_normal((9, d)).astype(np.float32)
    q[:4] = x[[5, 6000, 31000, 59999]] + 0.5 * q[:4]
    qc = np.array([7, 60, -1, 7, 60, 12345, -1, 3, 60], dtype=np.int32)
    idx = T.GpuIndex(doc_base=500).set_dense(x, shortlist=shortlist).set_collections(coll)
    S, I, cnt, nres = idx.dense_search(dev(q), k, collections=dev(qc))
    S, I, cnt = S.cpu().numpy(), I.cpu().numpy(), cnt.cpu().numpy()
    dn = O.doc_norms_f64(x)
    for i in range(9):
        s = O.cosine_scores_f64(x, q[i], dn)
        if qc[i] != -1:
            s[coll != qc[i]] = -np.inf
        ts, ti = O.topk_desc(s, k)
        assert cnt[i] == len(ti) and np.array_equal(I[i, :len(ti)], ti + 500), (i, qc[i])
        assert np.array_equal(S[i, :len(ti)], ts)
    assert cnt[5] == 0
    # every scan applies the filter as it emits (round 4: the f32 and f16-inline scans too -- until
    # then they filtered in K4 only, let c times the aimed candidates through for a collection of
    # 1/c of the corpus and answered from the exhaustive path): even the 2 % collections stay on
    # the shortlist path, nothing is rescued
    assert nres == 0
    unf = O.topk_desc(O.cosine_scores_f64(x, q[0], dn), 4 * k)[1]
    assert np.sum(coll[unf] == 7) < k            # post-filtering an over-fetched list falls short
    # ... and every query is CERTIFIED by the shortlist path itself (flags straight from the scan
    # entry points, before any rescue), the fat collection (25 % of the rows) included
    fl = "f32" != shortlist
    kp = 192 if fl else 128
    args = (idx.docs, idx.dnorm, idx.inv_norm, dev(q), k, kp, 500)
    S2, I2, c2, flg = (T._native.dense_topk_f16(idx.docs, idx.docs16, idx.doc_rel_err, *args[1:],
                                                doc_coll=idx.doc_coll, query_coll=dev(qc)) if fl else
                       T._native.dense_topk(*args, doc_coll=idx.doc_coll, query_coll=dev(qc)))
    flg = flg.cpu().numpy()
    assert all(flg[i] & 1 for i in range(9) if i != 5), flg


def test_backend_rpcs_filter_by_collection_on_device(T):
    """The two RPCs with p_collection through the Supabase-shaped client."""
    from triple_hybrid_rag_amd import synth
    from triple_hybrid_rag_amd.backend import CorpusStore, GpuIndexClient
    n, d = 30000, 768
    x = synth.dense_rows(0, n, d)
    csr, idf, avgdl, v = lexical_fixture(T, n)
    store = CorpusStore.synthetic(n, vocab_size=v)
    store.collections = [f"col{i * 7919 % 40}" if i % 11 else None for i in range(n)]
    idx = T.GpuIndex().set_dense(x).set_lexical(csr.rowptr, csr.post_doc, csr.post_tf, csr.doclen, idf, avgdl)
    client = GpuIndexClient(idx, store, org_id="org")
    q = synth.dense_queries(2, d, n)
    rows = client.rpc("rag2_semantic_search", {"p_org_id": "org", "p_embedding": q[0].tolist(),
                                               "p_limit": 25, "p_collection": "col3"}).execute().data
    assert len(rows) == 25 and all(store.collections[int(r["child_id"][1:])] == "col3" for r in rows)
    mask = np.array([c == "col3" for c in store.collections])
    s = O.cosine_scores_f64(x, q[0])
    s[~mask] = -np.inf
    ts, ti = O.topk_desc(s, 25)
    assert [int(r["child_id"][1:]) for r in rows] == list(ti) and [r["similarity"] for r in rows] == list(ts)
    top = np.argsort(-csr.df_local)[:3]
    text = " ".join(f"t{t}" for t in top)
    rows = client.rpc("rag2_lexical_search", {"p_org_id": "org", "p_query": text, "p_limit": 30,
                                              "p_collection": "col3"}).execute().data
    coll_id = np.array([1 if c == "col3" else 0 for c in store.collections], dtype=np.int32)
    _, Il = O.bm25_topk(csr.rowptr, csr.post_doc, csr.post_tf, csr.doclen, idf, avgdl, [list(top)], n, 30,
                        doc_coll=coll_id, query_coll=[1])
    assert [int(r["child_id"][1:]) for r in rows] == list(Il[0]) and len(rows) == 30
    assert client.rpc("rag2_semantic_search", {"p_org_id": "org", "p_embedding": q[0].tolist(),
                                               "p_limit": 5, "p_collection": "nope"}).execute().data == []
    # the reference's lexical RPC is an AND over the query terms (plainto_tsquery): optional here
    client_and = GpuIndexClient(idx, store, org_id="org", lexical_and=True)
    rows_and = client_and.rpc("rag2_lexical_search", {"p_org_id": "org", "p_query": text, "p_limit": 30,
                                                      "p_collection": None}).execute().data
    _, Ia = O.bm25_topk(csr.rowptr, csr.post_doc, csr.post_tf, csr.doclen, idf, avgdl, [list(top)], n, 30,
                        conjunctive=True)
    assert [int(r["child_id"][1:]) for r in rows_and] == list(Ia[0])
    # a token no chunk holds: the SQL AND returns no rows; the OR form ranks on the known terms
    q_oov = {"p_org_id": "org", "p_query": text + " palavradesconhecida", "p_limit": 30, "p_collection": None}
    assert client_and.rpc("rag2_lexical_search", q_oov).execute().data == []
    rows_or = client.rpc("rag2_lexical_search", q_oov).execute().data
    _, Io = O.bm25_topk(csr.rowptr, csr.post_doc, csr.post_tf, csr.doclen, idf, avgdl, [list(top)], n, 30)
    assert [int(r["child_id"][1:]) for r in rows_or] == list(Io[0])


def test_graph_matches_oracle(T):
    from triple_hybrid_rag_amd import synth
    n = 40000
    g = synth.build_graph(n)
    seeds = synth.graph_queries(40, n, 3)
    seeds[0] = [5, 5, -1]
    seeds[1] = -1
    e = synth.n_entities(n)
    for hops, lo, hi in ((2, 0, n), (1, 0, n), (0, 0, n), (2, 10000, 25000)):
        gs = synth.build_graph(n, lo, hi)
        idx = T.GpuIndex(doc_base=lo)
        idx.n_docs = hi - lo
        idx.set_graph(gs.ent_rowptr, gs.ent_col, gs.men_rowptr, gs.men_chunk, gs.men_conf)
        S, I, cnt = idx.graph_search(dev(seeds), 50, hops)
        Se, Ie = O.graph_topk(gs.ent_rowptr, gs.ent_col, gs.men_rowptr, gs.men_chunk, gs.men_conf,
                              seeds, hops, hi - lo, 50, chunk_base=lo)
        assert_topk_equal(S, I, cnt, Se, Ie, [len(s) for s in Se], f"graph hops={hops} [{lo},{hi})")
    assert e == len(g.ent_rowptr) - 1


def test_graph_capacity_tiers(T):
    """Queries beyond the first launch's small on-chip capacities (1024 entities / 2048
    contributions) are redone with the full ones; beyond those (> 4096 entities) the walk runs
    in global memory from the transposed mention CSR -- every tier gives the oracle's bits, and
    no overflow flag comes back.  Without the transposed CSR the overflow is reported."""
    rng = np.random.default_rng(9)
    n_ent, n_chunks = 20000, 50000
    deg = np.full(n_ent, 2, dtype=np.int64)
    deg[0], deg[1], deg[2] = 1200, 40, 6000           # hubs: medium, small-but-many-mentions, huge
    ent_rowptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    ent_col = rng.integers(3, n_ent, ent_rowptr[-1]).astype(np.int32)
    ent_col[ent_rowptr[0]:ent_rowptr[1]] = rng.choice(np.arange(3, n_ent), 1200, replace=False)
    ent_col[ent_rowptr[2]:ent_rowptr[3]] = rng.choice(np.arange(3, n_ent), 6000, replace=False)
    men = np.full(n_ent, 1, dtype=np.int64)
    men[1] = 3000                                      # > 2048 contributions from one entity
    men[3] = 9000                                      # > 8192 contributions: beyond the full capacities
    men_rowptr = np.concatenate([[0], np.cumsum(men)]).astype(np.int64)
    men_chunk = rng.integers(0, n_chunks, men_rowptr[-1]).astype(np.int32)
    men_conf = rng.uniform(0.5, 1.0, men_rowptr[-1]).astype(np.float32)
    seeds = np.array([[0, -1, -1], [1, -1, -1], [2, -1, -1], [7, 8, 9], [3, 2, 5]], dtype=np.int32)
    for lo, hi in ((0, n_chunks), (20000, 45000)):     # whole corpus / a document shard
        idx = T.GpuIndex(doc_base=lo)
        idx.n_docs = hi - lo
        idx.set_graph(ent_rowptr, ent_col, men_rowptr, men_chunk, men_conf)
        G = idx.graph
        for hops in (0, 1, 2):
            Se, Ie = O.graph_topk(ent_rowptr, ent_col, men_rowptr, men_chunk, men_conf, seeds, hops,
                                  hi - lo, 50, chunk_base=lo)
            S, I, cnt, flg = T._native.graph_topk(G["ent_rowptr"], G["ent_col"], G["men_rowptr"],
                                                  G["men_chunk"], G["men_conf"], dev(seeds), hops, 50,
                                                  lo, hi - lo)
            flg = flg.cpu().numpy()
            big = [4] + ([2] if hops >= 1 else [])     # beyond the full on-chip capacities
            ok = [q for q in range(5) if q not in big]
            assert all(flg[q] & 2 for q in big) and all(flg[q] & 1 for q in ok)
            assert_topk_equal(S[ok], I[ok], cnt[ok], [Se[q] for q in ok], [Ie[q] for q in ok],
                              [len(Se[q]) for q in ok], f"graph tiers hops={hops}")
            S, I, cnt = idx.graph_search(dev(seeds), 50, hops)      # with the third tier
            assert_topk_equal(S, I, cnt, Se, Ie, [len(s) for s in Se], f"graph fallback hops={hops}")


def test_rerank_order_matches_reference_sort(T, golden):
    """thr_rerank_order = the reference's stable descending sort on ``rerank_score or 0``
    (retrieval.py:449-455; ordering pinned by tests/golden/legacy_rerank.json), with the per-shard
    score lists of a document-sharded index merged by max."""
    rng = np.random.default_rng(3)
    nq, n = 37, 100
    ids = rng.permutation(10 ** 6)[: nq * n].reshape(nq, n).astype(np.int64)
    sc = rng.standard_normal((nq, n)).astype(np.float32).round(1)      # many ties
    cnt = rng.integers(0, n + 1, nq).astype(np.int32)
    cnt[0], cnt[1] = n, 0
    none = rng.random((nq, n)) < 0.1                                    # nobody scored: -> 0.0
    owner = rng.integers(0, 3, (nq, n))
    lists = np.full((3, nq, n), -np.inf, dtype=np.float32)
    for l in range(3):
        lists[l][(owner == l) & ~none] = sc[(owner == l) & ~none]
    for top_k in (10, 100):
        oi, os_, oc = T._native.rerank_order(dev(lists), dev(ids), dev(cnt), top_k)
        oi, os_, oc = oi.cpu().numpy(), os_.cpu().numpy(), oc.cpu().numpy()
        for q in range(nq):
            c = int(cnt[q])
            vals = [None if none[q, p] else float(sc[q, p]) for p in range(c)]
            order = O.rerank_order(vals)[:top_k]
            assert oc[q] == len(order)
            assert list(oi[q, :len(order)]) == [int(ids[q, p]) for p in order]
            assert list(os_[q, :len(order)]) == [vals[p] or 0.0 for p in order]
            assert np.all(oi[q, len(order):] == -1)
    # the reference's own ordering cases (Qwen3VLReranker.rerank with injected scores)
    for c in golden("legacy_rerank.json")["qwen_rerank"]:
        n_sent = len(c["documents_sent"] or [])
        if not c["enabled"] or n_sent == 0:
            continue
        k = min(c["top_k"] or c["default_top_k"], n_sent)
        s1 = np.array([c["scores"][:n_sent]], dtype=np.float32)
        i1 = np.arange(n_sent, dtype=np.int64)[None]
        oi, _, _ = T._native.rerank_order(dev(s1), dev(i1), None, k)
        # float32 holds the fixture's 3-decimal scores distinctly enough: same ties, same order
        exp = [int(o["chunk_id"][1:]) for o in c["out"]][:k]
        assert list(oi[0].cpu().numpy()) == exp


def test_rrf_fuse_matches_reference_python(T, golden):
    rng = np.random.default_rng(11)
    nq = 64
    def lists(width, pool, p_dup):
        out = np.full((nq, width), -1, dtype=np.int64)
        for q in range(nq):
            n = rng.integers(0, width + 1)
            ids = rng.integers(0, pool, n) if rng.random() < p_dup else rng.choice(pool, n, replace=False)
            out[q, :n] = ids
        return out
    lex, sem, gra = lists(50, 300, 0.3), lists(100, 300, 0.3), lists(50, 300, 0.3)
    for w in ({"lexical": 0.7, "semantic": 0.8, "graph": 1.0}, {"lexical": 2.0, "semantic": 0.5, "graph": 0.0}):
        for use in ((1, 1, 1), (1, 1, 0), (0, 1, 0), (0, 1, 1)):
            a = dev(lex) if use[0] else None
            c = dev(gra) if use[2] else None
            ids, sc, rk, cnt = T._native.rrf_fuse(a, dev(sem), c, 10, w["lexical"], w["semantic"],
                                                  w["graph"], 60, want_ranks=True)
            ids, sc, cnt = ids.cpu().numpy(), sc.cpu().numpy(), cnt.cpu().numpy()
            for q in range(nq):
                cut = lambda l: [int(x) for x in l[q][: list(l[q] < 0).index(True) if (l[q] < 0).any() else None]]
                ei, es = O.fused_topk_ids(cut(lex) if use[0] else None, cut(sem),
                                          cut(gra) if use[2] else None, 10, w)
                assert list(ids[q, :cnt[q]]) == ei and list(sc[q, :cnt[q]]) == es
    # the reference's own _fuse_rrf outputs (tests/golden/rrf_fuse.json): single-candidate lists
    for c in golden("rrf_fuse.json"):
        if len(c["ids"]) != 1 or c["ranks"][0] == [None, None, None]:
            continue
        l, s, g = c["ranks"][0]
        w = {"lexical": 0.7, "semantic": 0.8, "graph": 1.0, **c["weights"]}
        def one(rank):
            a = np.full((1, 100), -1, dtype=np.int64)
            if rank:
                a[0, :rank] = np.arange(1000, 1000 + rank)
                a[0, rank - 1] = 42
            return dev(a)
        ids, sc, _, cnt = T._native.rrf_fuse(one(l), one(s), one(g), 256, w["lexical"], w["semantic"], w["graph"])
        ids, sc = ids.cpu().numpy()[0], sc.cpu().numpy()[0]
        assert sc[list(ids).index(42)] == c["scores"][0]


def test_maxsim_matches_oracle(T):
    from triple_hybrid_rag_amd import synth
    dt = synth.doc_tokens(0, 300, 128, 128)
    qt = synth.query_tokens(6, 32, 128)
    rng = np.random.default_rng(3)
    cand = rng.integers(0, 300, size=(6, 100)).astype(np.int32)
    cand[0, 5] = -1
    got = T._native.maxsim(dev(qt), dev(dt), dev(cand)).cpu().numpy().astype(np.float64)
    packed = T._native.maxsim_pack(dev(dt))
    got_p = T._native.maxsim(dev(qt), packed, dev(cand), packed=True).cpu().numpy().astype(np.float64)
    assert np.array_equal(got, got_p)        # same products, same accumulation order
    # thr_maxsim_ids: the same candidates as global ids of a shard that starts at id 7000; ids of
    # other shards (below the base, past the end) and negative ids score -inf
    gids = cand.astype(np.int64) + 7000
    gids[0, 5] = -1
    gids[1, 0], gids[1, 1] = 6999, 7300
    got_g = T._native.maxsim_ids(dev(qt), packed, dev(gids), 7000, packed=True).cpu().numpy().astype(np.float64)
    assert got_g[1, 0] == -np.inf and got_g[1, 1] == -np.inf and got_g[0, 5] == -np.inf
    got_g[1, :2] = got[1, :2]
    assert np.array_equal(got_g, got)
    exp = CO.maxsim(qt, dt, cand)
    assert got[0, 5] == -np.inf and exp[0, 5] == -np.inf
    ok = np.isfinite(exp)
    assert np.max(np.abs(got[ok] - exp[ok])) < 1e-4
    # other shapes: 64 query tokens, 64 doc tokens, dim 64
    rng = np.random.default_rng(8)
    qt2 = rng.standard_normal((3, 64, 64)).astype(np.float16)
    dt2 = rng.standard_normal((40, 64, 64)).astype(np.float16)
    c2 = rng.integers(0, 40, size=(3, 7)).astype(np.int32)
    c2[1, 3] = 40                             # out of range scores -inf, like a negative index
    got = T._native.maxsim(dev(qt2), dev(dt2), dev(c2)).cpu().numpy().astype(np.float64)
    got_p = T._native.maxsim(dev(qt2), T._native.maxsim_pack(dev(dt2)), dev(c2), packed=True)
    assert np.array_equal(got, got_p.cpu().numpy().astype(np.float64))
    assert got[1, 3] == -np.inf
    c2[1, 3] = -1
    exp = O.maxsim_scores(qt2, dt2, c2)
    got[1, 3] = exp[1, 3] = 0.0
    assert np.max(np.abs(got - exp) / np.maximum(1.0, np.abs(exp))) < 2e-4


def test_merge_topk(T):
    import torch
    rng = np.random.default_rng(2)
    G, nq, k = 8, 33, 100
    S = np.sort(rng.standard_normal((G, nq, k)), axis=2)[:, :, ::-1].copy()
    I = rng.permutation(G * nq * k).reshape(G, nq, k).astype(np.int64)
    S[3, :, 60:] = -np.inf
    I[3, :, 60:] = -1
    S[1, 0, :5] = S[0, 0, :5]  # cross-shard ties -> id asc
    def check(S, I, s, i, k_out=100):
        s, i = s.cpu().numpy(), i.cpu().numpy()
        for q in range(S.shape[1]):
            es, ei = O.topk_desc(S[:, q, :].ravel(), k_out, I[:, q, :].ravel())
            m = len(es)
            assert np.array_equal(i[q][:m], ei) and np.array_equal(s[q][:m], es)
            assert np.all(i[q][m:] == -1) and np.all(np.isneginf(s[q][m:]))

    s, i, cnt = T._native.merge_topk(dev(S), dev(I), 100)
    check(S, I, s, i)
    # read in place from one gathered [G, 2, nq, k] tile (what distributed.gather_topk returns)
    tile = torch.stack([dev(S).view(torch.int64), dev(I)], dim=1).contiguous()
    s, i, cnt = T._native.merge_topk(tile[:, 0].view(torch.float64), tile[:, 1], 10)
    check(S, I, s, i, 10)
    # a list that is not ranked: the block falls back to sorting everything
    S2 = S.copy()
    S2[5, :, :] = S2[5, :, ::-1]
    s, i, cnt = T._native.merge_topk(dev(S2), dev(I), 100)
    check(S2, I, s, i)
    # fewer valid entries than k_out; and more than 1024 candidates per query (general kernel)
    s, i, cnt = T._native.merge_topk(dev(S[3:4, :, :]), dev(I[3:4, :, :]), 100)
    check(S[3:4], I[3:4], s, i)
    assert np.all(cnt.cpu().numpy() == 60)
    S3 = np.concatenate([S, S + 0.25], axis=0)
    I3 = np.concatenate([I, I + G * nq * k], axis=0)
    s, i, cnt = T._native.merge_topk(dev(S3), dev(I3), 100)
    check(S3, I3, s, i)


def test_triple_hybrid_pipeline_fused_top10(T):
    """Config-4 shape at small N: dense + BM25 + graph -> RRF -> top-10, then MaxSim rerank."""
    from triple_hybrid_rag_amd import synth
    n, d, nq = 30000, 768, 40
    x = synth.dense_rows(0, n, d)
    q = synth.dense_queries(nq, d, n)
    csr, idf, avgdl, v = lexical_fixture(T, n)
    qt = synth.lexical_queries(nq, csr.df_local, 4)
    g = synth.build_graph(n)
    seeds = synth.graph_queries(nq, n, 3)
    dtok = synth.doc_tokens(0, n, 32, 64)
    qtok = synth.query_tokens(nq, 32, 64)
    idx = (T.GpuIndex().set_dense(x)
           .set_lexical(csr.rowptr, csr.post_doc, csr.post_tf, csr.doclen, idf, avgdl)
           .set_graph(g.ent_rowptr, g.ent_col, g.men_rowptr, g.men_chunk, g.men_conf)
           .set_tokens(dtok))
    res = idx.retrieve_batch(dev(q), dev(qt), dev(seeds), top_k=10)
    Sd, Id, _ = CO.dense_topk_exact(x, q, 100)
    Sl, Il = O.bm25_topk(csr.rowptr, csr.post_doc, csr.post_tf, csr.doclen, idf, avgdl, qt, n, 50)
    Sg, Ig = O.graph_topk(g.ent_rowptr, g.ent_col, g.men_rowptr, g.men_chunk, g.men_conf, seeds, 2, n, 50)
    ids, sc, cnt = res.ids.cpu().numpy(), res.scores.cpu().numpy(), res.counts.cpu().numpy()
    for i in range(nq):
        ei, es = O.fused_topk_ids(list(Il[i]), list(Id[i]), list(Ig[i]), 10)
        assert list(ids[i, :cnt[i]]) == ei, f"fused order differs for query {i}"
        assert list(sc[i, :cnt[i]]) == es
    # dense + BM25 only with weights 0.8 / 0.7 (config 3)
    res2 = idx.retrieve_batch(dev(q), dev(qt), None, top_k=10)
    ids2 = res2.ids.cpu().numpy()
    for i in range(nq):
        ei, _ = O.fused_topk_ids(list(Il[i]), list(Id[i]), None, 10)
        assert list(ids2[i]) == ei
    # rerank: fused top-100 -> MaxSim -> stable sort -> top-10
    res3 = idx.retrieve_batch(dev(q), dev(qt), dev(seeds), top_k=10, qtok=dev(qtok), rerank_top_k=100)
    ids3 = res3.ids.cpu().numpy()
    for i in range(nq):
        f100, _ = O.fused_topk_ids(list(Il[i]), list(Id[i]), list(Ig[i]), 100)
        ms = O.maxsim_scores(qtok[i:i + 1], dtok, np.array([f100], dtype=np.int64))[0]
        order = O.rerank_order(list(ms))[:10]
        gaps = np.abs(np.diff(np.sort(ms)[::-1][:11]))
        if gaps.min() > 1e-3:  # ordering is pinned wherever oracle gaps exceed the tolerance
            assert [f100[j] for j in order] == list(ids3[i])


def test_lexical_index_built_on_the_device(T, tmp_path):
    """SURVEY 8f.1 at scale: thr_lexical_build turns tokenised rows into the CSR on the GPU --
    arrays equal to the host builders' (synth.build_lexical_csr for (doc, term, tf) rows,
    index_build.build_lexical for texts), whatever the order of the rows, with repeated pairs
    adding up and out-of-range entries dropped; ``save(gpu_index=...)`` carries what set-up computed
    on the device and ``load().to_gpu()`` searches with it, not recomputing any of it."""
    import time
    from triple_hybrid_rag_amd import index_build as IB
    from triple_hybrid_rag_amd import synth
    n = 200_000
    v = synth.vocab_size(n)
    doc, term, tf = synth.lexical_rows(0, n, n)
    csr = synth.build_lexical_csr(doc, term, tf, n, v)
    rng = np.random.default_rng(3)
    perm = rng.permutation(len(doc))
    # shuffled rows + junk the build must drop (doc / term out of range, tf <= 0) + a pair split in two
    d2 = np.concatenate([doc[perm], [n, -1, 5, 7, 7]]).astype(np.int32)
    t2 = np.concatenate([term[perm], [3, 3, v, 11, 11]]).astype(np.int32)
    f2 = np.concatenate([tf[perm], [1, 1, 1, 2, 3]]).astype(np.int32)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rowptr, pd_, ptf, dl, df = T._native.lexical_build(dev(d2), dev(t2), dev(f2), n, v)
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t0
    # (doc 7 may already hold term 11: the reference CSR adds the extra 5 the same way)
    e_doc, e_term, e_tf = np.concatenate([doc, [7]]), np.concatenate([term, [11]]), np.concatenate([tf, [5]])
    key = e_term.astype(np.int64) << 32 | e_doc
    uk, inv = np.unique(key, return_inverse=True)
    e_sum = np.bincount(inv, weights=e_tf).astype(np.int32)
    assert np.array_equal(pd_.cpu().numpy(), (uk & 0xFFFFFFFF).astype(np.int32)) and np.array_equal(ptf.cpu().numpy(), e_sum)
    assert np.array_equal(df.cpu().numpy(), np.bincount((uk >> 32).astype(np.int64), minlength=v))
    assert np.array_equal(rowptr.cpu().numpy(), np.concatenate([[0], np.cumsum(df.cpu().numpy())]))
    dl_exp = np.bincount(e_doc, weights=e_tf, minlength=n).astype(np.float32)
    dl_exp[5] += 1                                          # (term out of range: still a token of chunk 5)
    assert np.array_equal(dl.cpu().numpy(), dl_exp) and build_s < 5.0
    # the clean rows: exactly synth.build_lexical_csr's arrays, and a searchable index
    idx = T.GpuIndex()
    idx.set_lexical_rows(dev(doc[perm].astype(np.int32)), dev(term[perm].astype(np.int32)), dev(tf[perm].astype(np.int32)), v, n_docs=n)
    L = idx.lex
    for got, want in ((L["rowptr"], csr.rowptr), (L["post_doc"], csr.post_doc), (L["post_tf"], csr.post_tf),
                      (L["doclen"], csr.doclen), (idx.df_local, csr.df_local)):
        assert np.array_equal(got.cpu().numpy(), want)
    idf = O.bm25_idf(n, csr.df_local)
    assert np.array_equal(L["idf"].cpu().numpy(), idf) and L["avgdl"] == csr.sum_dl_local / n
    qt = synth.lexical_queries(16, csr.df_local, 4)
    S, I, cnt = idx.bm25_search(dev(qt), 50)
    Se, Ie = O.bm25_topk(csr.rowptr, csr.post_doc, csr.post_tf, csr.doclen, idf, csr.sum_dl_local / n, qt, n, 50)
    assert_topk_equal(S, I, cnt, Se, Ie, [len(s_) for s_ in Se], "device-built index")
    # texts: one entry per token occurrence (tf = None), against the host builder
    words = [f"w{i}" for i in range(300)]
    texts = [" ".join(words[int(t_)] for t_ in rng.integers(0, 300, int(rng.integers(0, 30)))) for _ in range(5000)]
    vocab_h, rp_h, pd_h, tf_h, dl_h, idf_h, avg_h = IB.build_lexical(texts)
    vocab_d, d_tok, t_tok = IB.lexical_rows(texts)
    assert vocab_d == vocab_h
    rp_d, pd_d, tf_d, dl_d, df_d = T._native.lexical_build(dev(d_tok), dev(t_tok), None, len(texts), len(vocab_d))
    for got, want in ((rp_d, rp_h), (pd_d, pd_h), (tf_d, tf_h), (dl_d, dl_h)):
        assert np.array_equal(got.cpu().numpy(), want)
    # a saved index carries the device-side set-up; a loaded one searches without recomputing it
    x, _ = rand_docs(5000, 768, 8)
    children = [{"id": f"c{i}", "parent_id": f"p{i // 4}", "document_id": "d", "text": texts[i], "page": 1,
                 "modality": "text", "embedding_1024": x[i].tolist()} for i in range(5000)]
    hi = IB.from_rows(children, device=True)
    hi_host = IB.from_rows(children)
    for name in ("rowptr", "post_doc", "post_tf", "doclen", "idf"):
        assert np.array_equal(getattr(hi, name), getattr(hi_host, name)), name
    assert hi.avgdl == hi_host.avgdl and hi.store.vocab == hi_host.store.vocab
    g = hi.to_gpu()
    IB.save(hi, str(tmp_path / "idx"), gpu_index=g)
    back = IB.load(str(tmp_path / "idx"))
    assert back.derived["f16_layout"] == T._native.dense_f16_layout(768) and back.store.texts[17] == texts[17]
    assert back.store.child_ids == [f"c{i}" for i in range(5000)] and back.store.row_index("c4999") == 4999
    orig = (T._native.bm25_bounds, T._native.bm25_dense_terms, T._native.dense_quantize_f16)

    def boom(*a, **k):
        raise AssertionError("a loaded index recomputed what was saved with it")
    T._native.bm25_bounds = T._native.bm25_dense_terms = T._native.dense_quantize_f16 = boom
    try:
        g2 = back.to_gpu()
    finally:
        T._native.bm25_bounds, T._native.bm25_dense_terms, T._native.dense_quantize_f16 = orig
    q = x[:8] + 0.3 * rng.standard_normal((8, 768)).astype(np.float32)
    qt = np.array([[vocab_h[words[int(t_)]] for t_ in rng.integers(0, 300, 3)] for _ in range(8)], dtype=np.int32)
    r1, r2 = g.retrieve_batch(dev(q), dev(qt), top_k=10), g2.retrieve_batch(dev(q), dev(qt), top_k=10)
    assert torch.equal(r1.ids, r2.ids) and torch.equal(r1.scores, r2.scores)


def test_index_build_rows_through_the_kernels(T, tmp_path):
    """reference-shaped table rows (rag_child_chunks / rag_parent_chunks / rag_entities /
    rag_relations / rag_entity_mentions) -> index_build.from_rows -> save / load -> to_gpu(), and
    every channel of the resulting GpuIndex equals the oracle run on the HostIndex arrays; the
    drop-in RPC client answers from the same index with the store's ids and texts."""
    from triple_hybrid_rag_amd import index_build as IB
    from triple_hybrid_rag_amd.backend import GpuIndexClient
    rng = np.random.default_rng(41)
    n, d, n_ent = 4000, 768, 300
    words = [f"w{i}" for i in range(400)]
    zipf = 1.0 / np.arange(1, 401)
    zipf /= zipf.sum()
    emb = rng.standard_normal((n, d)).astype(np.float32)
    children = []
    for i in range(n):
        toks = rng.choice(400, size=int(rng.integers(3, 40)), p=zipf)
        children.append({"id": f"c{i}", "parent_id": f"p{i // 4}", "document_id": f"d{i // 50}",
                         "text": " ".join(words[t] for t in toks), "page": 1 + i % 7,
                         "modality": "table" if i % 13 == 0 else "text",
                         "collection": f"col{i % 5}",
                         "embedding_1024": None if i % 97 == 0 else emb[i].tolist()})
    parents = [{"id": f"p{j}", "text": f"parent text {j}", "section_heading": f"S{j % 9}"}
               for j in range(n // 4)]
    ents = [{"id": f"e{j}", "name": f"entity{j}"} for j in range(n_ent)]
    rels = [{"subject_entity_id": f"e{int(a)}", "object_entity_id": f"e{int(b)}"}
            for a, b in rng.integers(0, n_ent, size=(900, 2))]
    mens = [{"entity_id": f"e{int(e)}", "child_chunk_id": f"c{int(c)}", "confidence": float(cf)}
            for e, c, cf in zip(rng.integers(0, n_ent, 2500), rng.integers(0, n, 2500),
                                rng.uniform(0.2, 1.0, 2500).astype(np.float32))]
    hi = IB.from_rows(children, parents, ents, rels, mens)
    IB.save(hi, str(tmp_path / "idx"))
    hi = IB.load(str(tmp_path / "idx"))
    idx = hi.to_gpu()
    x = np.asarray(hi.docs)
    assert idx.shortlist == "f16" and not x[97].any()
    nq = 24
    q = (x[rng.integers(1, n, nq)] + 0.7 * rng.standard_normal((nq, d))).astype(np.float32)
    vocab = hi.store.vocab
    qt = np.array([[vocab[words[int(t)]] for t in rng.choice(400, size=3, p=zipf)] for _ in range(nq)],
                  dtype=np.int32)
    seeds = rng.integers(0, n_ent, size=(nq, 2)).astype(np.int32)
    res = idx.retrieve_batch(dev(q), dev(qt), dev(seeds), top_k=10)
    _, Id, _ = CO.dense_topk_exact(x, q, 100)
    _, Il = O.bm25_topk(hi.rowptr, hi.post_doc, hi.post_tf, hi.doclen, hi.idf, hi.avgdl, qt, n, 50)
    _, Ig = O.graph_topk(hi.ent_rowptr, hi.ent_col, hi.men_rowptr, hi.men_chunk, hi.men_conf, seeds, 2, n, 50)
    ids, sc, cnt = res.ids.cpu().numpy(), res.scores.cpu().numpy(), res.counts.cpu().numpy()
    for i in range(nq):
        assert 97 not in Id[i]                      # the row without an embedding never ranks
        ei, es = O.fused_topk_ids(list(Il[i]), list(Id[i]), list(Ig[i]), 10)
        assert list(ids[i, :cnt[i]]) == ei and list(sc[i, :cnt[i]]) == es, i
    # the Supabase-shaped client over the same index: ids, texts and the collection filter
    client = GpuIndexClient(idx, hi.store, org_id="org", lexical_and=True)
    rows = client.rpc("rag2_semantic_search", {"p_org_id": "org", "p_embedding": q[0].tolist(),
                                               "p_limit": 20, "p_collection": "col2"}).execute().data
    s = O.cosine_scores_f64(x, q[0])
    s[np.array([c["collection"] != "col2" for c in children])] = -np.inf
    _, ti = O.topk_desc(s, 20)
    assert [r["child_id"] for r in rows] == [f"c{int(j)}" for j in ti]
    assert all(r["text"] == children[int(r["child_id"][1:])]["text"] for r in rows)
    rows = client.rpc("rag2_lexical_search", {"p_org_id": "org", "p_query": f"{words[3]} {words[11]}",
                                              "p_limit": 15}).execute().data
    _, Il2 = O.bm25_topk(hi.rowptr, hi.post_doc, hi.post_tf, hi.doclen, hi.idf, hi.avgdl,
                         np.array([[vocab[words[3]], vocab[words[11]]]], dtype=np.int32), n, 15,
                         conjunctive=True)
    assert [r["child_id"] for r in rows] == [f"c{int(j)}" for j in Il2[0]]


def test_side_stream_channels_equal_the_single_stream_pipeline(T, monkeypatch):
    """retrieve_batch runs the lexical and graph kernels on a second (high-priority) HIP stream.
    Twelve batches whose inputs are produced on the main stream right before each call -- by
    asynchronous host-to-device copies and device ops -- must give what the one-stream pipeline
    (THR_SIDE_STREAM=0) gives, bit for bit, with the rerank leg on."""
    from triple_hybrid_rag_amd import synth
    n, d, nq = 20000, 768, 96
    x = synth.dense_rows(0, n, d)
    csr, idf, avgdl, v = lexical_fixture(T, n)
    g = synth.build_graph(n)
    dtok = synth.doc_tokens(0, n, 32, 64)
    idx = (T.GpuIndex().set_dense(x)
           .set_lexical(csr.rowptr, csr.post_doc, csr.post_tf, csr.doclen, idf, avgdl)
           .set_graph(g.ent_rowptr, g.ent_col, g.men_rowptr, g.men_chunk, g.men_conf)
           .set_tokens(dtok))
    rng = np.random.default_rng(5)
    for it in range(12):
        q = synth.dense_queries(nq, d, n)[rng.permutation(nq)]
        qt = synth.lexical_queries(nq, csr.df_local, 4)[rng.permutation(nq)]
        seeds = synth.graph_queries(nq, n, 3)[rng.permutation(nq)]
        qtok = synth.query_tokens(nq, 32, 64)[rng.permutation(nq)]
        out = {}
        for mode in ("0", "hi"):
            monkeypatch.setenv("THR_SIDE_STREAM", mode)
            # fresh device tensors every time, some through device-side ops on the main stream
            qd = torch.from_numpy(q).pin_memory().cuda(non_blocking=True) * 1.0
            qtd = torch.from_numpy(np.ascontiguousarray(qt)).pin_memory().cuda(non_blocking=True) + 0
            sd = torch.from_numpy(np.ascontiguousarray(seeds)).pin_memory().cuda(non_blocking=True) + 0
            qk = torch.from_numpy(qtok).pin_memory().cuda(non_blocking=True)
            r = idx.retrieve_batch(qd, qtd, sd, top_k=10, qtok=qk, rerank_top_k=50)
            out[mode] = (r.ids.clone(), r.scores.clone(), r.counts.clone(),
                         r.channels["lexical"][1].clone(), r.channels["graph"][1].clone())
            del qd, qtd, sd, qk, r
        torch.cuda.synchronize()
        for a, b in zip(out["0"], out["hi"]):
            assert torch.equal(a, b), it


def test_full_size_1m_dense(T):
    """BASELINE config 1 (1M x 768, top-10 of top-100): exact check of 8 queries against the
    C oracle + size-independent properties on the whole batch."""
    from triple_hybrid_rag_amd import synth
    n, d, nq = 1_000_000, 768, 128
    x = synth.dense_rows(0, n, d)
    q = synth.dense_queries(nq, d, n)
    idx = T.GpuIndex().set_dense(x)
    S, I, cnt, nres = idx.dense_search(dev(q), 100)
    S, I = S.cpu().numpy(), I.cpu().numpy()
    assert nres == 0
    assert np.all(np.diff(S, axis=1) <= 0) and np.all(cnt.cpu().numpy() == 100)
    for row in I:
        assert len(set(row)) == 100 and row.min() >= 0 and row.max() < n
    # every returned score is the oracle's score of that row
    for qi in (0, 1, 77):
        s = O.cosine_scores_f64(x[I[qi]], q[qi])
        assert np.array_equal(s, S[qi])
    # planted queries find their planted row first (cos ~ 0.894 vs ~0.19 for noise)
    assert np.all(S[::2, 0] > 0.8) and np.all(S[1::2, 0] < 0.4)
    sub = [0, 1, 2, 3, 64, 65, 126, 127]
    dn = CO.doc_norms(x)   # the oracle's own norms (and thr_doc_norms equals them on all 1M rows)
    assert np.array_equal(idx.dnorm.cpu().numpy(), dn)
    Se, Ie, _ = CO.dense_topk_exact(x, q[sub], 100, dnorm=dn)
    for j, qi in enumerate(sub):
        assert np.array_equal(I[qi], Ie[j]) and np.array_equal(S[qi], Se[j])
    # a 1536-query batch (6 workgroup tiles of 256 queries on the f16 copy scan; the bench's is
    # 2048): every shortlist flavour returns the same bits, and they are the oracle's fast path's
    qb = synth.dense_queries(1536, d, n)
    Sf, If = O.dense_topk_fast(x, qb, 100, dnorm=dn)
    Sf, If = np.stack(Sf), np.stack(If)
    for mode in ("f16", "f16-inline", "f32"):
        idx = T.GpuIndex().set_dense(x, shortlist=mode)
        S, I, cnt, nres = idx.dense_search(dev(qb), 100)
        assert nres == 0
        assert np.array_equal(I.cpu().numpy(), If) and np.array_equal(S.cpu().numpy(), Sf), mode
        del idx


def test_dropin_retrieve_end_to_end(T):
    """The reference-shaped surface (RAG2Retriever.retrieve over GpuIndexClient) on a synthetic
    corpus: contexts, ranks and RRF scores equal the CPU oracle pipeline."""
    import asyncio
    from triple_hybrid_rag_amd import synth
    from triple_hybrid_rag_amd.backend import CorpusStore, GpuIndexClient
    from triple_hybrid_rag_amd.config import SETTINGS
    from triple_hybrid_rag_amd.rag2.embedder import PrecomputedEmbedder
    from triple_hybrid_rag_amd.rag2.query_planner import QueryPlanner
    from triple_hybrid_rag_amd.rag2.retrieval import RAG2Retriever

    n, d = 20000, 768
    x = synth.dense_rows(0, n, d)
    csr, idf, avgdl, v = lexical_fixture(T, n)
    g = synth.build_graph(n)
    dtok = synth.doc_tokens(0, n, 32, 64)
    idx = (T.GpuIndex().set_dense(x)
           .set_lexical(csr.rowptr, csr.post_doc, csr.post_tf, csr.doclen, idf, avgdl)
           .set_graph(g.ent_rowptr, g.ent_col, g.men_rowptr, g.men_chunk, g.men_conf)
           .set_tokens(dtok))
    store = CorpusStore.synthetic(n, vocab_size=v, n_entities=synth.n_entities(n))

    class TokEmb:
        def embed_query_tokens(self, query):
            return synth.query_tokens(1, 32, 64)[0]

    client = GpuIndexClient(idx, store, org_id="org", token_embedder=TokEmb())
    saved = dict(SETTINGS.__dict__)
    SETTINGS.rag2_graph_enabled = True
    SETTINGS.rag2_safety_threshold = 0.0
    SETTINGS.rag2_denoise_alpha = 0.0
    try:
        rng = np.random.default_rng(12)
        emb = PrecomputedEmbedder(store_dim=d)
        qraw = synth.dense_queries(6, d, n)
        for qi in range(6):
            terms = [int(t) for t in rng.integers(50, v, 3)]
            ents = [int(e) for e in rng.integers(0, synth.n_entities(n), 2)]
            text = " ".join([f"t{t}" for t in terms] + [f"entity{e}" for e in ents])
            raw = np.concatenate([qraw[qi] * 7.0, rng.standard_normal(300).astype(np.float32)])
            emb.register(text, raw.tolist())
            r = RAG2Retriever(org_id="org", embedder=emb, query_planner=QueryPlanner(graph=True),
                              graph_enabled=True)
            r._supabase = client
            res = asyncio.run(r.retrieve(text, top_k=10, skip_rerank=True))
            assert res.success and not res.refused and len(res.contexts) == 10
            assert sorted(res.timings) == ["expansion", "fusion", "planning", "retrieval", "safety"]
            # oracle pipeline
            qv = np.asarray(emb.embed_query(text), dtype=np.float32)
            _, Id, _ = CO.dense_topk_exact(x, qv[None], 100)
            kw = text.split()
            tids = [store.vocab[w] for w in kw if w in store.vocab]
            _, Il = O.bm25_topk(csr.rowptr, csr.post_doc, csr.post_tf, csr.doclen, idf, avgdl, [tids], n, 50)
            seeds = client.find_entities(kw, 50)
            _, Ig = O.graph_topk(g.ent_rowptr, g.ent_col, g.men_rowptr, g.men_chunk, g.men_conf,
                                 [seeds], 2, n, 50)
            ei, es = O.fused_topk_ids(list(Il[0]), list(Id[0]), list(Ig[0]), 10)
            assert [c.child_id for c in res.contexts] == [f"c{i}" for i in ei]
            assert [c.rrf_score for c in res.contexts] == es
            assert all(c.parent_text == f"parent text {int(c.child_id[1:]) // 4}" for c in res.contexts)
            # with the late-interaction reranker in the loop
            res2 = asyncio.run(r.retrieve(text, top_k=5))
            assert "rerank" in res2.timings and len(res2.contexts) <= 5
            f20, _ = O.fused_topk_ids(list(Il[0]), list(Id[0]), list(Ig[0]), 20)
            ms = CO.maxsim(synth.query_tokens(1, 32, 64), dtok, np.array([f20], dtype=np.int32))[0] / 32.0
            got = {c.child_id: c.rerank_score for c in res2.contexts}
            for cid, sc in got.items():
                assert abs(sc - ms[f20.index(int(cid[1:]))]) < 1e-5
            assert [c.rerank_score for c in res2.contexts] == sorted(got.values(), reverse=True)
        # another tenant sees nothing
        r = RAG2Retriever(org_id="other", embedder=emb, query_planner=QueryPlanner())
        r._supabase = client
        res = asyncio.run(r.retrieve(text, skip_planning=True, skip_rerank=True))
        assert res.refused and res.refusal_reason == "No candidates found"
    finally:
        SETTINGS.__dict__.update(saved)


@pytest.mark.parametrize("mode", ["f16", "f16-inline"])
@pytest.mark.parametrize("n,d", [(50000, 768), (20000, 1024), (4000, 512)])
def test_dense_f16_shortlist_is_still_exact(T, n, d, mode):
    """Shortlist on the f16 matrix cores (from a float16 copy, or from float32 rows rounded in
    registers): results must be the SAME bits as the float32 oracle (scores come from float64
    rescoring of float32 rows; the certificate covers quantisation)."""
    x, rng = rand_docs(n, d, 77)
    x[13] = 0
    x[200:230] = x[199]                      # 31-way tie inside the top-100 of query 0
    q = rng.standard_normal((70, d)).astype(np.float32)
    q[0] = x[199] + 0.05 * q[0]
    q[::2] = x[rng.integers(0, n, 35)] + 0.5 * q[::2]
    q[5] = 0
    idx = T.GpuIndex(doc_base=123).set_dense(x, shortlist=mode)
    assert 0 < idx.doc_rel_err < 6e-4        # ~2^-12/sqrt(3) typical, 2^-11 worst case
    nz = x.any(axis=1)
    if mode == "f16":
        # the copy is fragment-major: [tile of 32 rows][k-step of 16 dims][h][r][8 halves]; it
        # holds the NORMALISED rows, NaN for rows without an embedding and for the tile padding
        img = idx.docs16.cpu().numpy()
        tiles = img.shape[0] // 32
        if os.environ.get("THR_DENSE_MFMA") != "32":
            # [tile][k32][row half ra][dim group g][r16][8]: the 16x16x32 MFMA's A fragments
            rows = (img.reshape(tiles, d // 32, 2, 4, 16, 8).transpose(0, 2, 4, 1, 3, 5)
                    .reshape(tiles * 32, d))
        else:
            rows = img.reshape(tiles, d // 16, 2, 32, 8).transpose(0, 3, 1, 2, 4).reshape(tiles * 32, d)
        unit = x.astype(np.float64) / np.maximum(np.linalg.norm(x.astype(np.float64), axis=1), 1e-300)[:, None]
        exp16 = unit.astype(np.float32).astype(np.float16)
        assert np.isnan(rows[n:]).all() and np.isnan(rows[:n][~nz]).all()
        diff = np.abs(rows[:n][nz].astype(np.float32) - exp16[nz].astype(np.float32))
        assert np.count_nonzero(diff) <= 1e-5 * diff.size and diff.max() <= 2.0 ** -11
        rel = np.linalg.norm(rows[:n][nz].astype(np.float64) - unit[nz], axis=1)
    else:
        assert idx.docs16 is None            # no second copy of the corpus
        x16 = x.astype(np.float16)
        rel = (np.linalg.norm(x16.astype(np.float64) - x, axis=1)[nz]
               / np.linalg.norm(x.astype(np.float64), axis=1)[nz])
    assert rel.max() <= idx.doc_rel_err
    S, I, cnt, flg = T._native.dense_topk_f16(idx.docs, idx.docs16, idx.doc_rel_err, idx.dnorm,
                                              idx.inv_norm, dev(q), 100, 256, 123)
    flags = flg.cpu().numpy()
    assert flags[5] & 1 == 0                 # zero query is never certified
    assert np.mean(flags & 1) > 0.9          # random data certifies through the f16 bound
    S, I, cnt, nres = idx.dense_search(dev(q), 100)
    Se, Ie, cnte = CO.dense_topk_exact(x, q, 100, doc_id_base=123)
    assert_topk_equal(S, I, cnt, Se, Ie, cnte, "dense-f16")


def _floor_search(T, x, q, k, n_shards, shortlist="f16", coll=None, qc=None, use_floor=True):
    """The dense channel of a document-sharded index with every shard in THIS process: shortlist on
    every shard, the lower bounds stacked (what the all-gather delivers), thr_dense_floor, finish on
    every shard, thr_merge_topk.  -> merged (S, I, cnt), per-shard counts [G, nq], the flags the
    finish wrote (before any rescue) [G, nq], the rescued queries per shard, the floor [nq]."""
    n = x.shape[0]
    shards = []
    for s in range(n_shards):
        lo, hi = s * n // n_shards, (s + 1) * n // n_shards
        idx = T.GpuIndex(doc_base=lo).set_dense(x[lo:hi], shortlist=shortlist)
        if coll is not None:
            idx.set_collections(coll[lo:hi])
        shards.append(idx)
    qd = dev(q)
    qcd = None if qc is None else dev(qc)
    gfloor = None
    if use_floor:
        lbs = torch.stack([ix.dense_shortlist(qd, k, n_shards, collections=qcd) for ix in shards])
        assert lbs.shape == (n_shards, q.shape[0], T.index.floor_width(k, n_shards))
        gfloor = T._native.dense_floor(lbs, k)
    outs = []
    for si, ix in enumerate(shards):
        if not use_floor:
            ix.dense_shortlist(qd, k, n_shards, collections=qcd)
        # the floor as thr_dense_floor's output on the even shards, as the gathered bounds
        # themselves (the k-th largest found inside the band kernel) on the odd ones: the same thing
        if use_floor and si % 2:
            outs.append(ix.dense_finish(qd, k, lb_all=lbs, collections=qcd))
        else:
            outs.append(ix.dense_finish(qd, k, gfloor, collections=qcd))
    S = torch.stack([o[0] for o in outs])
    I = torch.stack([o[1] for o in outs])
    Sm, Im, cm = T._native.merge_topk(S, I, k)
    cnts = torch.stack([o[2] for o in outs]).cpu().numpy()
    flags = torch.stack([o[3] for o in outs]).cpu().numpy()
    nres = [int(o[4]) for o in outs]
    return (Sm, Im, cm), cnts, flags, nres, (None if gfloor is None else gfloor.cpu().numpy())


@pytest.mark.parametrize("n,d,nq,k,g,shortlist", [(160000, 768, 48, 100, 8, "f16"), (32000, 768, 20, 100, 8, "f16"),
                                                  (90000, 512, 33, 10, 4, "f16"), (60000, 1024, 17, 100, 2, "f16"),
                                                  (120000, 768, 24, 100, 8, "f16-inline")])
def test_dense_shard_floor_rescoring_only_what_can_reach_the_global_topk(T, n, d, nq, k, g, shortlist):
    """thr_dense_shortlist_f16 / thr_dense_floor / thr_dense_finish_f16 (thr_hip.h e1): every shard
    sends its top-m scan scores as lower bounds, the k-th largest of all of them is a floor under
    the GLOBAL k-th score, and a shard rescores only rows that can clear it -- about k / G instead
    of k.  The merged lists are the oracle's top-k, bit for bit, with nothing rescued."""
    x, rng = rand_docs(n, d, 41)
    x[n // 3] = 0
    q = rng.standard_normal((nq, d)).astype(np.float32)
    q[:3] = x[[11, n // 2, n - 5]] + 0.3 * q[:3]
    (S, I, cnt), cnts, flags, nres, gfloor = _floor_search(T, x, q, k, g, shortlist)
    Se, Ie, cnte = CO.dense_topk_exact(x, q, k)
    assert_topk_equal(S, I, cnt, Se, Ie, cnte, "shard-floor")
    assert sum(nres) == 0 and np.all(flags & 1), "every shard's list is certified by the shortlist path"
    # the floor is a LOWER bound of ||q|| x the global k-th cosine
    qn = np.linalg.norm(q.astype(np.float64), axis=1)
    assert np.all(gfloor <= qn * Se[:, k - 1])
    assert np.all(gfloor > qn * Se[:, k - 1] - 0.02 * qn), "and a close one"
    # what it buys: the shards together rescored little more than k rows per query, not G * k
    assert np.all(cnts.sum(0) >= k)
    assert cnts.sum(0).mean() < 1.35 * k + 4 * g, cnts.sum(0).mean()
    # same bits without the floor (every shard rescoring a top-k of its own)
    (S0, I0, c0), cnts0, _, _, _ = _floor_search(T, x, q, k, g, shortlist, use_floor=False)
    assert torch.equal(S0, S) and torch.equal(I0, I)
    assert cnts0.sum(0).mean() >= min(g * k, 0.9 * n / g * g if n < g * k else g * k) * 0.99


def test_dense_shard_floor_ties_nulls_skew_and_collections(T):
    """The cases a floor could get wrong: the k best all on ONE shard (the floor from top-m values
    is loose there, never wrong), a giant tie across shards, a shard of NULL rows, a collection
    filter thinner than k -- each merged result equal to the oracle's."""
    n, d, k, g = 64000, 768, 100, 8
    x, rng = rand_docs(n, d, 43)
    per = n // g
    x[3 * per + 100:3 * per + 400] = x[3 * per + 99]          # 301-way tie inside shard 3
    x[per - 50:per + 50] = x[5 * per + 7]                      # 101-way tie over shards 0, 1 (and 5)
    x[6 * per:7 * per] = 0                                     # shard 6: no embeddings at all
    q = rng.standard_normal((8, d)).astype(np.float32)
    q[0] = x[3 * per + 99] * 1.5
    q[1] = x[5 * per + 7] + 0.001 * q[1]
    q[2] = 0
    # skew: query 3's best 150 rows all on shard 2
    x[2 * per + 1000:2 * per + 1150] = q[3] / np.linalg.norm(q[3]) + 0.05 * x[2 * per + 1000:2 * per + 1150]
    (S, I, cnt), cnts, flags, nres, gfloor = _floor_search(T, x, q, k, g)
    Se, Ie, cnte = CO.dense_topk_exact(x, q, k)
    assert_topk_equal(S, I, cnt, Se, Ie, cnte, "shard-floor-ties")
    assert cnts[6].sum() == 0
    # the skewed query: shard 2 holds the whole top-k but can only send its top m = 26, so the
    # floor is the 74th best of the OTHER shards' rows -- loose (they rescore ~80 rows for nothing)
    # and still far from G * k
    assert cnts[2, 3] == k and k < cnts[:, 3].sum() < 2 * k
    # collections: 2 % ones (thinner than k on a shard: its list is everything it has of them)
    coll = (np.arange(n) * 7919 % 50).astype(np.int32)
    qc = np.array([7, -1, 7, 3, 12345, 11, -1, 49], dtype=np.int32)
    (S, I, cnt), cnts, flags, nres, _ = _floor_search(T, x, q, k, g, coll=coll, qc=qc)
    S, I, cnt = S.cpu().numpy(), I.cpu().numpy(), cnt.cpu().numpy()
    dn = O.doc_norms_f64(x)
    for i in range(8):
        sc = O.cosine_scores_f64(x, q[i], dn)
        if qc[i] != -1:
            sc[coll != qc[i]] = -np.inf
        ts, ti = O.topk_desc(sc, k)
        assert cnt[i] == len(ti) and np.array_equal(I[i, :len(ti)], ti), (i, qc[i])
        assert np.array_equal(S[i, :len(ti)], ts)
    assert cnt[4] == 0


@pytest.mark.parametrize("seed", range(8))
def test_dense_shard_floor_random_shapes(T, seed):
    """Random corpus sizes (uneven shards, some below the sampling threshold), row lengths, shard
    counts, k from 1 to 200, clustered rows, NULL rows and a collection filter on half the draws:
    the merged shard-floor result is the oracle's top-k."""
    rng = np.random.default_rng(1000 + seed)
    d = int(rng.choice([512, 768, 1024]))
    g = int(rng.choice([2, 3, 5, 8]))
    k = int(rng.choice([1, 10, 50, 100, 128, 200]))
    n = int(rng.integers(3000, 70000))
    nq = int(rng.integers(1, 40))
    x, _ = rand_docs(n, d, 2000 + seed)
    c0 = int(rng.integers(0, n - 600))
    x[c0:c0 + 500] = x[c0] + 0.02 * x[c0:c0 + 500]          # a cluster of near-duplicates on one shard
    x[rng.integers(0, n, 20)] = 0
    q = rng.standard_normal((nq, d)).astype(np.float32)
    q[0] = x[c0] * 0.7
    coll = qc = None
    if seed % 2:
        coll = rng.integers(0, 6, n).astype(np.int32)
        qc = rng.integers(-1, 7, nq).astype(np.int32)          # (6: a collection nobody has)
    (S, I, cnt), cnts, flags, nres, gfloor = _floor_search(T, x, q, k, g, "f16", coll, qc)
    S, I, cnt = S.cpu().numpy(), I.cpu().numpy(), cnt.cpu().numpy()
    dn = O.doc_norms_f64(x)
    for i in range(nq):
        sc = O.cosine_scores_f64(x, q[i], dn)
        if qc is not None and qc[i] != -1:
            sc[coll != qc[i]] = -np.inf
        ts, ti = O.topk_desc(sc, k)
        assert cnt[i] == len(ti) and np.array_equal(I[i, :len(ti)], ti), (seed, i, d, g, k, n)
        assert np.array_equal(S[i, :len(ti)], ts), (seed, i)


def test_dense_floor_kernel_known_answers(T):
    """thr_dense_floor alone: k-th largest of the shards' values, duplicates counted, -inf when
    fewer than k are finite."""
    rng = np.random.default_rng(5)
    for g, m, nq, k in ((8, 16, 33, 100), (2, 100, 7, 100), (3, 16, 5, 50), (1, 16, 4, 10), (64, 128, 3, 128)):
        v = rng.standard_normal((g, nq, m)).astype(np.float32)
        v[:, 0, :] = 0.25                       # all equal
        if nq > 2:
            v[:, 2, m // 2:] = -np.inf          # half the slots empty
        got = T._native.dense_floor(dev(v), k).cpu().numpy()
        for i in range(nq):
            flat = np.sort(v[:, i, :].ravel())[::-1]
            exp = flat[k - 1] if flat.size >= k and np.isfinite(flat[k - 1]) else -np.inf
            assert got[i] == exp, (g, m, k, i)
    with pytest.raises(T._native.NativeError):
        T._native.dense_floor(dev(np.zeros((65, 2, 128), dtype=np.float32)), 10)


def test_dense_finish_needs_its_shortlist(T):
    """dense_finish works on the candidate lists a dense_shortlist call left in the index's
    workspace: anything in between, or another batch shape, is refused instead of ranking stale lists."""
    x, rng = rand_docs(20000, 768, 77)
    q = rng.standard_normal((5, 768)).astype(np.float32)
    idx = T.GpuIndex().set_dense(x)
    with pytest.raises(T._native.NativeError):
        idx.dense_finish(dev(q), 10)
    idx.dense_shortlist(dev(q), 10, 4)
    S, I, cnt, _, _ = idx.dense_finish(dev(q), 10)              # no floor: thr_dense_topk_f16's result
    Se, Ie, cnte = CO.dense_topk_exact(x, q, 10)
    assert_topk_equal(S, I, cnt, Se, Ie, cnte, "finish-without-floor")
    with pytest.raises(T._native.NativeError):
        idx.dense_finish(dev(q[:3]), 10)                          # another batch
    idx.dense_search(dev(q), 10)
    with pytest.raises(T._native.NativeError):
        idx.dense_finish(dev(q), 10)                              # the lists were overwritten


def _sharded_worker(rank, world, port, n, d, out_dir):
    import os
    import sys
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import triple_hybrid_rag_amd as T
    from triple_hybrid_rag_amd import synth
    from triple_hybrid_rag_amd.distributed import ShardedIndex, shard_range
    torch.cuda.set_device(0)
    lo, hi = shard_range(n, rank, world)
    v = synth.vocab_size(n)
    doc, term, tf = synth.lexical_rows(lo, hi - lo, n)
    csr = synth.build_lexical_csr(doc, term, tf, hi - lo, v)
    # global statistics: every rank needs the df / doc-length sums of the whole corpus
    df = torch.from_numpy(csr.df_local.copy())
    sdl = torch.tensor([csr.sum_dl_local], dtype=torch.float64)
    dist.all_reduce(df)
    dist.all_reduce(sdl)
    idf = O.bm25_idf(n, df.numpy())
    g = synth.build_graph(n, lo, hi)
    idx = (T.GpuIndex(doc_base=lo).set_dense(synth.dense_rows(lo, hi - lo, d))
           .set_lexical(csr.rowptr, csr.post_doc, csr.post_tf, csr.doclen, idf, float(sdl.item()) / n)
           .set_graph(g.ent_rowptr, g.ent_col, g.men_rowptr, g.men_chunk, g.men_conf))
    idx.set_tokens(synth.doc_tokens(lo, hi - lo, 32, 64))
    q = synth.dense_queries(24, d, n)
    dfq = df.numpy().copy()
    qt = synth.lexical_queries(24, dfq, 4)
    seeds = synth.graph_queries(24, n, 3)
    sh = ShardedIndex(idx)
    assert sh._floor_exchange() is not None     # the dense channel is split around the floor exchange
    res = sh.retrieve_batch(dev(q), dev(qt), dev(seeds), top_k=10)
    # ... which changes what a shard rescored, not a bit of any result
    res0 = ShardedIndex(idx, floor=False).retrieve_batch(dev(q), dev(qt), dev(seeds), top_k=10)
    assert torch.equal(res0.ids, res.ids) and torch.equal(res0.scores, res.scores)
    for name in ("semantic", "lexical", "graph"):
        assert torch.equal(res0.channels[name][0], res.channels[name][0]), name
        assert torch.equal(res0.channels[name][1], res.channels[name][1]), name
    _, _, c_floor, _ = idx.dense_search(dev(q), 100, sync=False, floor_exchange=sh._floor_exchange())
    np.save(os.path.join(out_dir, f"floor_cnt_{rank}.npy"), c_floor.cpu().numpy())
    rr = sh.retrieve_batch(dev(q), dev(qt), dev(seeds), top_k=10, qtok=dev(synth.query_tokens(24, 32, 64)),
                           rerank_top_k=100)
    if rank == 0:
        np.save(os.path.join(out_dir, "ids.npy"), res.ids.cpu().numpy())
        np.save(os.path.join(out_dir, "sc.npy"), res.scores.cpu().numpy())
        np.save(os.path.join(out_dir, "df.npy"), df.numpy())
        np.save(os.path.join(out_dir, "rr_ids.npy"), rr.ids.cpu().numpy())
        np.save(os.path.join(out_dir, "rr_sc.npy"), rr.scores.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_doc_sharded_pipeline_on_one_gpu(T, tmp_path):
    """Config-4 shape, 2 document shards (two processes sharing this GPU, gloo rendezvous standing
    in for RCCL): per-shard kernels + all-gather + thr_merge_topk + RRF == unsharded oracle."""
    import os
    import torch.multiprocessing as mp
    from triple_hybrid_rag_amd import synth
    n, d, world = 30000, 768, 2
    port = 29700 + os.getpid() % 1500
    mp.get_context("spawn")
    mp.spawn(_sharded_worker, args=(world, port, n, d, str(tmp_path)), nprocs=world, join=True)
    ids, sc = np.load(tmp_path / "ids.npy"), np.load(tmp_path / "sc.npy")
    df = np.load(tmp_path / "df.npy")
    x = synth.dense_rows(0, n, d)
    q = synth.dense_queries(24, d, n)
    csr, idf, avgdl, v = lexical_fixture(T, n)
    assert np.array_equal(df, csr.df_local)
    qt = synth.lexical_queries(24, csr.df_local, 4)
    g = synth.build_graph(n)
    seeds = synth.graph_queries(24, n, 3)
    _, Id, _ = CO.dense_topk_exact(x, q, 100)
    _, Il = O.bm25_topk(csr.rowptr, csr.post_doc, csr.post_tf, csr.doclen, idf, avgdl, qt, n, 50)
    _, Ig = O.graph_topk(g.ent_rowptr, g.ent_col, g.men_rowptr, g.men_chunk, g.men_conf, seeds, 2, n, 50)
    # the floor path: the two shards together rescored little more than the 100 rows of the answer
    fc = np.load(tmp_path / "floor_cnt_0.npy").astype(int) + np.load(tmp_path / "floor_cnt_1.npy")
    assert np.all(fc >= 100) and fc.mean() < 140, fc
    rr_ids, rr_sc = np.load(tmp_path / "rr_ids.npy"), np.load(tmp_path / "rr_sc.npy")
    dtok, qtok = synth.doc_tokens(0, n, 32, 64), synth.query_tokens(24, 32, 64)
    for i in range(24):
        ei, es = O.fused_topk_ids(list(Il[i]), list(Id[i]), list(Ig[i]), 10)
        assert list(ids[i]) == ei and list(sc[i]) == es
        # configs[4]: fused top-100 -> MaxSim on the owning shard -> second all-gather -> stable sort
        fi, _ = O.fused_topk_ids(list(Il[i]), list(Id[i]), list(Ig[i]), 100)
        ms = CO.maxsim(qtok[i:i + 1], dtok, np.array([fi], dtype=np.int64))[0]
        order = O.rerank_order([float(np.float32(v)) for v in ms])[:10]
        got = list(rr_ids[i])
        assert np.max(np.abs(rr_sc[i] - ms[[fi.index(g) for g in got]])) < 1e-4
        # identical order wherever the oracle's MaxSim scores are more than 2e-4 apart
        exp = [fi[p] for p in order]
        es_ = np.array([ms[p] for p in order])
        for a in range(10):
            if (a == 0 or es_[a - 1] - es_[a] > 2e-4) and (a == 9 or es_[a] - es_[a + 1] > 2e-4):
                assert got[a] == exp[a]


def _dropin_calls(client, n, d):
    """What a user of the reference does with its retriever, over ``client``: retrieve() with a
    graph-asking plan and a collection filter, the agent tool with nothing injected (tenant
    discovery, default embedder / planner, MaxSim rerank), a refusal -- as JSON-able data."""
    import asyncio
    from triple_hybrid_rag_amd import backend, synth
    from triple_hybrid_rag_amd.config import SETTINGS
    from triple_hybrid_rag_amd.rag2.query_planner import QueryPlan
    from triple_hybrid_rag_amd.rag2.retrieval import RAG2Retriever
    from triple_hybrid_rag_amd.tools import crm_knowledge as crm
    saved = dict(SETTINGS.__dict__)
    q = synth.dense_queries(6, d, n)

    class Emb:
        def embed_query(self, text):
            return q[sum(map(ord, text)) % 6].tolist()

    class Planner:
        async def plan_async(self, query, collection=None):
            return QueryPlan(original_query=query, keywords=query.split(), semantic_query_text=query,
                             requires_graph=True, cypher_query="MATCH (e) RETURN e")

    out = {"retrieve": [], "tool": []}
    try:
        backend.set_default_client(client)
        SETTINGS.rag2_enabled = SETTINGS.rag2_rerank_enabled = SETTINGS.rag2_graph_enabled = True
        SETTINGS.rag2_embed_dim_store = d
        SETTINGS.rag2_safety_threshold = SETTINGS.rag2_denoise_alpha = 0.0
        loop = asyncio.new_event_loop()
        for text, coll, rerank in (("t100 t2000 t77", None, False), ("t5 t9 entity12", "faq", False),
                                   ("t3 t4000 t17 t2", None, True), ("entity3 t11", "pricing", True)):
            r = RAG2Retriever(org_id="org-7", embedder=Emb(), query_planner=Planner(), graph_enabled=True)
            r._supabase = client
            res = loop.run_until_complete(r.retrieve(text, collection=coll, top_k=8, skip_rerank=not rerank))
            out["retrieve"].append([[c.child_id, c.lexical_rank, c.semantic_rank, c.graph_rank, c.rrf_score,
                                     c.rerank_score, c.parent_text] for c in res.contexts])
        loop.close()
        for text, cat in (("t100 t2000 t77", None), ("t100 t2000 t77", "faq")):
            t = crm.search_knowledge_base(text, cat, 5)
            t.pop("timings_ms", None)
            out["tool"].append(t)
        SETTINGS.rag2_safety_threshold = 10.0
        out["tool"].append(crm.search_knowledge_base("t100 t2000 t77", None, 5))
    finally:
        SETTINGS.__dict__.update(saved)
        backend.set_default_client(None)
    return out


class _TokEmb:
    def embed_query_tokens(self, query):
        from triple_hybrid_rag_amd import synth
        return synth.query_tokens(4, 32, 64)[sum(map(ord, query)) % 4]


def _dropin_store(n):
    from triple_hybrid_rag_amd import synth
    from triple_hybrid_rag_amd.backend import CorpusStore
    store = CorpusStore.synthetic(n, vocab_size=synth.vocab_size(n),
                                  n_entities=synth.build_graph(n).ent_rowptr.shape[0] - 1)
    store.collections = ["faq" if i % 5 == 0 else "pricing" for i in range(n)]
    return store


def _sharded_client_worker(rank, world, port, n, d, out_dir):
    import json
    import os
    import sys
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import triple_hybrid_rag_amd as T
    from triple_hybrid_rag_amd import synth
    from triple_hybrid_rag_amd.distributed import shard_range
    from triple_hybrid_rag_amd.sharded_client import ShardedIndexClient, ShardWorker
    torch.cuda.set_device(0)
    lo, hi = shard_range(n, rank, world)
    doc, term, tf = synth.lexical_rows(lo, hi - lo, n)
    csr = synth.build_lexical_csr(doc, term, tf, hi - lo, synth.vocab_size(n))
    df, sum_dl = synth.lexical_global_stats(n)   # the whole corpus' statistics on every shard
    g = synth.build_graph(n, lo, hi)
    idx = (T.GpuIndex(doc_base=lo).set_dense(synth.dense_rows(lo, hi - lo, d))
           .set_lexical(csr.rowptr, csr.post_doc, csr.post_tf, csr.doclen, O.bm25_idf(n, df), sum_dl / n)
           .set_graph(g.ent_rowptr, g.ent_col, g.men_rowptr, g.men_chunk, g.men_conf)
           .set_tokens(synth.doc_tokens(lo, hi - lo, 32, 64))
           .set_collections(np.array([0 if i % 5 == 0 else 1 for i in range(lo, hi)], dtype=np.int32)))
    names = ["faq", "pricing"]
    if rank != 0:
        served = ShardWorker(idx, collection_names=names).serve()
        with open(os.path.join(out_dir, f"served{rank}.json"), "w") as f:
            json.dump(served, f)
    else:
        with ShardedIndexClient(idx, _dropin_store(n), collection_names=names, org_id="org-7",
                                token_embedder=_TokEmb()) as client:
            got = _dropin_calls(client, n, d)
        with open(os.path.join(out_dir, "got.json"), "w") as f:
            json.dump(got, f)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_dropin_client_on_one_gpu(T, tmp_path):
    """Row e2: ``RAG2Retriever.retrieve()`` and ``search_knowledge_base()`` over a DOCUMENT-SHARDED
    index (two processes sharing this GPU, rank 0 the ShardedIndexClient, rank 1 in
    ShardWorker.serve(); gloo standing in for RCCL) return the contexts -- ids, per-channel ranks,
    float64 RRF scores, MaxSim rerank scores, parent texts, tool dicts, refusals -- of the
    one-shard GpuIndexClient over the whole corpus."""
    import json
    import os
    import torch.multiprocessing as mp
    from triple_hybrid_rag_amd import synth
    from triple_hybrid_rag_amd.backend import GpuIndexClient
    n, d, world = 30000, 768, 2
    port = 30900 + os.getpid() % 1500
    mp.spawn(_sharded_client_worker, args=(world, port, n, d, str(tmp_path)), nprocs=world, join=True)
    got = json.load(open(tmp_path / "got.json"))
    assert json.load(open(tmp_path / "served1.json")) > 20      # every RPC of every call reached the worker
    csr, idf, avgdl, v = lexical_fixture(T, n)
    g = synth.build_graph(n)
    idx = (T.GpuIndex().set_dense(synth.dense_rows(0, n, d))
           .set_lexical(csr.rowptr, csr.post_doc, csr.post_tf, csr.doclen, idf, avgdl)
           .set_graph(g.ent_rowptr, g.ent_col, g.men_rowptr, g.men_chunk, g.men_conf)
           .set_tokens(synth.doc_tokens(0, n, 32, 64)))
    one = GpuIndexClient(idx, _dropin_store(n), org_id="org-7", token_embedder=_TokEmb())
    exp = json.loads(json.dumps(_dropin_calls(one, n, d)))
    assert len(got["retrieve"]) == len(exp["retrieve"]) == 4
    for gq, eq in zip(got["retrieve"], exp["retrieve"]):
        assert len(gq) == len(eq) > 0
        for gc, ec in zip(gq, eq):
            assert gc[:5] == ec[:5] and gc[6] == ec[6]          # id, three ranks, float64 RRF score, parent text
            assert (gc[5] is None) == (ec[5] is None) and (gc[5] is None or abs(gc[5] - ec[5]) < 1e-6)
    assert any(c[3] is not None for ctx in got["retrieve"] for c in ctx)          # the graph channel answered
    assert any(c[5] is not None for ctx in got["retrieve"] for c in ctx)          # the rerank leg ran
    # 'faq' = every fifth chunk (the graph channel takes no collection filter, reference retrieval.py:316-356)
    assert all(int(c[0][1:]) % 5 == 0 for c in got["retrieve"][1] if c[1] is not None or c[2] is not None)
    assert got["tool"] == exp["tool"] and got["tool"][0]["result_count"] == 5 and got["tool"][2]["refused"]


def test_dense_rows_of_a_length_no_scan_is_built_for(T, caplog):
    """The reference's legacy RAG 1.0 store keeps 4000-d rows (src/voice_agent/config.py:216,
    database/migrations/20260113_halfvec_4000.sql:70-105): ``set_dense`` routes such a row length
    to the exhaustive float64 path -- with a warning, not THR_ERR_UNSUPPORTED -- and the legacy
    ``kb_chunks_vector_search`` RPC answers with the oracle's ids and float64 cosines."""
    import logging
    from triple_hybrid_rag_amd.backend import CorpusStore, GpuIndexClient
    n, d = 3000, 4000
    rng = np.random.default_rng(11)
    x = rng.standard_normal((n, d)).astype(np.float32)
    x[7] = 0.0                                   # a chunk without an embedding
    q = rng.standard_normal((6, d)).astype(np.float32)
    with caplog.at_level(logging.WARNING, logger="triple_hybrid_rag_amd.index"):
        idx = T.GpuIndex().set_dense(x)
    assert idx.shortlist == "exact" and any("4000" in r.getMessage() for r in caplog.records)
    S, I, cnt, resc = idx.dense_search(dev(q), 50)
    Se, Ie, _ = CO.dense_topk_exact(x, q, 50)
    assert resc == 0 and np.array_equal(I.cpu().numpy(), np.stack(Ie)) and np.array_equal(S.cpu().numpy(), np.stack(Se))
    res = idx.retrieve_batch(dev(q), top_k=10)   # the batch pipeline over it: dense -> RRF
    for i in range(6):
        assert list(res.ids[i].cpu().numpy()) == O.fused_topk_ids(None, list(Ie[i]), None, 10)[0]
    with pytest.raises(T._native.NativeError, match="row length"):
        T.GpuIndex().set_dense(x, shortlist="f16")
    client = GpuIndexClient(idx, CorpusStore.synthetic(n), org_id="org")
    rows = client.rpc("kb_chunks_vector_search", {"p_org_id": "org", "p_embedding": q[2].tolist(), "p_limit": 5}).execute().data
    assert [r["id"] for r in rows] == [f"c{i}" for i in Ie[2][:5]]
    assert [r["similarity"] for r in rows] == [float(v) for v in Se[2][:5]]


def test_gather_topk_over_rccl_single_rank(T):
    """The ``nccl`` (= RCCL) branch of the exchange -- all_gather_into_tensor on device tensors --
    run with a 1-rank process group on this GPU, merge included."""
    import os
    import torch.distributed as dist
    from triple_hybrid_rag_amd.distributed import ShardedIndex, gather_rows, gather_topk
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29950 + os.getpid() % 40))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        assert dist.get_backend() == "nccl"
        x, rng = rand_docs(20000, 768, 3)
        q = rng.standard_normal((40, 768)).astype(np.float32)
        idx = T.GpuIndex(doc_base=1000).set_dense(x)
        S, I, cnt, _ = idx.dense_search(dev(q), 100)
        Sg, Ig = gather_topk(S, I)
        torch.cuda.synchronize()
        assert Sg.shape == (1, 40, 100) and torch.equal(Sg[0], S) and torch.equal(Ig[0], I)
        Sm, Im, cm = T._native.merge_topk(Sg, Ig, 100)
        assert torch.equal(Im, I) and torch.equal(Sm, S)
        r = gather_rows(S.to(torch.float32))
        assert r.shape == (1, 40, 100) and torch.equal(r[0], S.to(torch.float32))
        # world == 1 through ShardedIndex's own code path as well
        sh = ShardedIndex(idx)
        sh.world = 2                      # force the exchange although the group has one rank
        Sx, Ix = sh._merge(S, I, 100)
        assert torch.equal(Ix, I)
        # several channels in one collective (what ShardedIndex.retrieve_batch sends), merged in
        # place from the gathered flat buffer
        from triple_hybrid_rag_amd.distributed import gather_topk_many
        S2, I2, _, _ = idx.dense_search(dev(q), 50)
        many = gather_topk_many([(S, I), (S2, I2)])
        torch.cuda.synchronize()
        for (Sg2, Ig2), (s_, i_), k_ in zip(many, [(S, I), (S2, I2)], (100, 50)):
            assert Sg2.shape == (1, 40, k_) and torch.equal(Sg2[0], s_) and torch.equal(Ig2[0], i_)
            Sm2, Im2, _ = T._native.merge_topk(Sg2, Ig2, k_)
            assert torch.equal(Im2, i_) and torch.equal(Sm2, s_)
        # and the whole sharded pipeline over the 1-rank RCCL group
        csr, idf, avgdl, v = lexical_fixture(T, 20000)
        idx.set_lexical(csr.rowptr, csr.post_doc, csr.post_tf, csr.doclen, idf, avgdl)
        from triple_hybrid_rag_amd import synth
        qt = synth.lexical_queries(40, csr.df_local, 4)
        ref = idx.retrieve_batch(dev(q), dev(qt), top_k=10)
        got = sh.retrieve_batch(dev(q), dev(qt), top_k=10)
        assert torch.equal(ref.ids, got.ids) and torch.equal(ref.scores, got.scores)
    finally:
        dist.destroy_process_group()


def test_dense_f16_rows_outside_half_range(T):
    """A value that rounds to +inf in float16: the in-flight-rounding scan must refuse the rows;
    the float16 COPY holds normalised rows, so it takes them (and stays exact)."""
    x, _ = rand_docs(3000, 768, 5)
    x[17, 3] = 1e6
    with pytest.raises(T._native.NativeError, match="float16"):
        T.GpuIndex().set_dense(x, shortlist="f16-inline")
    q = x[:4].copy()
    Se, Ie, cnte = CO.dense_topk_exact(x, q, 10)
    for mode in ("f32", "f16"):
        idx = T.GpuIndex().set_dense(x, shortlist=mode)
        S, I, cnt, _ = idx.dense_search(dev(q), 10)
        assert_topk_equal(S, I, cnt, Se, Ie, cnte, "wide-range rows " + mode)


@pytest.mark.parametrize("shortlist", ["f32", "f16", "f16-inline"])
def test_dense_edge_shapes_and_near_ties(T, shortlist):
    rng = np.random.default_rng(31)
    # tiny corpora, k > n, single query, k = 1
    # ... and more query tiles than CUs per XCD (2100 queries = 33 tiles of 64 / 66 of 32)
    for n, d, nq, k in ((1, 768, 1, 5), (33, 768, 2, 100), (300, 512, 1, 1), (9000, 768, 3, 7),
                        (20001, 768, 2100, 10)):
        x = rng.standard_normal((n, d)).astype(np.float32)
        q = rng.standard_normal((nq, d)).astype(np.float32)
        idx = T.GpuIndex(doc_base=5).set_dense(x, shortlist=shortlist)
        S, I, cnt, _ = idx.dense_search(dev(q), k)
        Se, Ie, cnte = CO.dense_topk_exact(x, q, k, doc_id_base=5)
        assert_topk_equal(S, I, cnt, Se, Ie, cnte, f"edge n={n}")
    # a cloud of near-duplicates around the query: hundreds of rows inside the error band
    n, d = 40000, 768
    x, _ = rand_docs(n, d, 41)
    base = x[7].copy()
    x[1000:2500] = base + 1e-6 * rng.standard_normal((1500, d)).astype(np.float32)
    x[3000:3050] = base + 1e-3 * rng.standard_normal((50, d)).astype(np.float32)
    q = np.stack([base, base + 0.01 * x[8], x[9]]).astype(np.float32)
    idx = T.GpuIndex().set_dense(x, shortlist=shortlist)
    S, I, cnt, nres = idx.dense_search(dev(q), 100)
    assert nres >= 1                      # 1500 rows in the band: more than the 1024-row second chance holds
    Se, Ie, cnte = CO.dense_topk_exact(x, q, 100)
    assert_topk_equal(S, I, cnt, Se, Ie, cnte, "near-ties")
    # unnormalised rows and queries (cosine must divide by both norms)
    x2 = (x[:20000] * rng.uniform(0.1, 30.0, (20000, 1))).astype(np.float32)
    q2 = (q * 13.0).astype(np.float32)
    idx = T.GpuIndex().set_dense(x2, shortlist=shortlist)
    S, I, cnt, _ = idx.dense_search(dev(q2), 50)
    Se, Ie, cnte = CO.dense_topk_exact(x2, q2, 50)
    assert_topk_equal(S, I, cnt, Se, Ie, cnte, "unnormalised")


def test_hybrid_rrf_rpc_and_legacy_searcher_on_gpu(T):
    """8f.3 / 8f.4: the single-call hybrid RPC and the RAG 1.0 searcher run on the same kernels."""
    import asyncio
    from triple_hybrid_rag_amd import synth
    from triple_hybrid_rag_amd.backend import CorpusStore, GpuIndexClient
    from triple_hybrid_rag_amd.retrieval.hybrid_search import HybridSearcher, SearchConfig
    n, d = 20000, 768
    x = synth.dense_rows(0, n, d)
    csr, idf, avgdl, v = lexical_fixture(T, n)
    idx = T.GpuIndex().set_dense(x).set_lexical(csr.rowptr, csr.post_doc, csr.post_tf, csr.doclen, idf, avgdl)
    client = GpuIndexClient(idx, CorpusStore.synthetic(n, vocab_size=v), org_id="org")
    q = synth.dense_queries(3, d, n)
    for qi in range(3):
        terms = [100 + qi, 2000 + qi, 77]
        text = " ".join(f"t{t}" for t in terms)
        rows = client.rpc("rag2_hybrid_rrf_search", {"p_org_id": "org", "p_embedding": q[qi].tolist(),
                                                     "p_query": text, "p_limit": 20}).execute().data
        _, Id, _ = CO.dense_topk_exact(x, q[qi:qi + 1], 40)
        _, Il = O.bm25_topk(csr.rowptr, csr.post_doc, csr.post_tf, csr.doclen, idf, avgdl, [terms], n, 40)
        ei, es = O.fused_topk_ids(list(Il[0]), list(Id[0]), None, 20)
        assert [r["child_id"] for r in rows] == [f"c{i}" for i in ei]
        assert [r["rrf_score"] for r in rows] == [float(np.float32(s)) for s in es]
        assert all((r["lexical_rank"] or r["semantic_rank"]) for r in rows)

        class Emb:
            def embed_query(self, text, _v=q[qi].tolist()):
                return _v, None

        hs = HybridSearcher("org", embedder=Emb(), config=SearchConfig(top_k_retrieve=50))._with(client)
        out = asyncio.run(hs.search(text, top_k=10))
        _, Id50, _ = CO.dense_topk_exact(x, q[qi:qi + 1], 50)
        _, Il50 = O.bm25_topk(csr.rowptr, csr.post_doc, csr.post_tf, csr.doclen, idf, avgdl, [terms], n, 50)
        exp = O.legacy_rrf_fusion([[{"chunk_id": f"c{i}"} for i in Id50[0]],
                                   [{"chunk_id": f"c{i}"} for i in Il50[0]]], 60)[:10]
        assert [r.chunk_id for r in out] == [e["chunk_id"] for e in exp]
        assert [r.rrf_score for r in out] == [e["rrf_score"] for e in exp]

    # the image channel (hybrid_search.py:423-458, kb_chunks_image_search): every 7th chunk has an
    # image vector; three lists fuse (vector, bm25, image with top_k_image = 3)
    rng = np.random.default_rng(77)
    img_rows = np.arange(0, n, 7)
    xi = rng.standard_normal((len(img_rows), 768)).astype(np.float32)
    img_idx = T.GpuIndex().set_dense(xi)
    client_i = GpuIndexClient(idx, client.store, org_id="org", image_index=img_idx, image_rows=img_rows)
    qimg = (xi[123] + 0.4 * rng.standard_normal(768)).astype(np.float32)
    rows = client_i.rpc("kb_chunks_image_search", {"p_org_id": "org", "p_image_embedding": qimg.tolist(),
                                                   "p_limit": 5}).execute().data
    si = O.cosine_scores_f64(xi, qimg)
    ts, ti = O.topk_desc(si, 5)
    assert [r["id"] for r in rows] == [f"c{int(img_rows[j])}" for j in ti] and rows[0]["id"] == f"c{int(img_rows[123])}"
    assert [r["similarity"] for r in rows] == [float(np.float32(v)) for v in ts]
    terms = [100, 2000, 77]
    text = " ".join(f"t{t}" for t in terms)

    class EmbI:
        def embed_query(self, text, _v=q[0].tolist(), _i=qimg.tolist()):
            return _v, _i

    hs = HybridSearcher("org", embedder=EmbI(),
                        config=SearchConfig(top_k_retrieve=50, use_image_search=True))._with(client_i)
    out = asyncio.run(hs.search(text, top_k=10))
    _, Id50, _ = CO.dense_topk_exact(x, q[0:1], 50)
    _, Il50 = O.bm25_topk(csr.rowptr, csr.post_doc, csr.post_tf, csr.doclen, idf, avgdl, [terms], n, 50)
    _, ti3 = O.topk_desc(si, 3)
    exp = O.legacy_rrf_fusion([[{"chunk_id": f"c{i}"} for i in Id50[0]],
                               [{"chunk_id": f"c{i}"} for i in Il50[0]],
                               [{"chunk_id": f"c{int(img_rows[j])}"} for j in ti3]], 60)[:10]
    assert [r.chunk_id for r in out] == [e["chunk_id"] for e in exp]
    assert [r.rrf_score for r in out] == [e["rrf_score"] for e in exp]
    assert f"c{int(img_rows[123])}" in [r.chunk_id for r in out]       # 1/61 ranks it with the best


def test_dense_rows_below_half_precision_range(T):
    """Rows scaled into float16's subnormal range: the in-flight-rounding scan refuses them, the
    normalised float16 copy does not care about the scale; results stay exact."""
    x, rng = rand_docs(20000, 768, 12)
    x *= 1e-6
    q = (x[:5] + 1e-7 * rng.standard_normal((5, 768))).astype(np.float32)
    idx = T.GpuIndex().set_dense(x)
    assert idx.shortlist == "f16" and idx.doc_rel_err < 6e-4
    S, I, cnt, nres = idx.dense_search(dev(q), 20)
    Se, Ie, cnte = CO.dense_topk_exact(x, q, 20)
    assert_topk_equal(S, I, cnt, Se, Ie, cnte, "tiny rows")
    with pytest.raises(T._native.NativeError, match="float16"):
        T.GpuIndex().set_dense(x, shortlist="f16-inline")
    T.GpuIndex.AUTO_COPY_FRACTION, keep = 0.0, T.GpuIndex.AUTO_COPY_FRACTION
    try:
        idx = T.GpuIndex().set_dense(x)      # no room for a copy and the rows do not fit float16
        assert idx.shortlist == "f32"
        S, I, cnt, nres = idx.dense_search(dev(q), 20)
        assert_topk_equal(S, I, cnt, Se, Ie, cnte, "tiny rows f32")
    finally:
        T.GpuIndex.AUTO_COPY_FRACTION = keep


def test_dense_auto_shortlist_choice(T):
    x, _ = rand_docs(3000, 768, 6)
    idx = T.GpuIndex().set_dense(x)
    assert idx.shortlist == "f16" and idx.docs16 is not None
    T.GpuIndex.AUTO_COPY_FRACTION, keep = 0.0, T.GpuIndex.AUTO_COPY_FRACTION
    try:
        idx = T.GpuIndex().set_dense(x)
        assert idx.shortlist == "f16-inline" and idx.docs16 is None
        x[5, 7] = -2e5                       # does not fit float16 as it is
        idx = T.GpuIndex().set_dense(x)
        assert idx.shortlist == "f32" and idx.docs16 is None and idx.doc_rel_err == 0.0
    finally:
        T.GpuIndex.AUTO_COPY_FRACTION = keep
    assert T.GpuIndex().set_dense(x).shortlist == "f16"                    # the copy is normalised
    assert T.GpuIndex().set_dense(x[:, :256].copy()).shortlist == "f32"   # no f16 kernel at dim 256


def test_cli_config0_100_queries_match_the_oracle_pipeline(T):
    """BASELINE.json configs[0] / SURVEY 8a11: the scripts/test_rag2.py counterpart (the
    reference's flags and --json keys) run as a process over the 10k-doc / 768-d synthetic corpus
    for 100 queries, every channel on; its contexts must be what the CPU oracle's channels +
    merge + RRF + truncation give for the same texts."""
    import json
    import os
    import subprocess
    import sys
    from triple_hybrid_rag_amd import synth
    from triple_hybrid_rag_amd.rag2.embedder import HashEmbedder
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    n, d, top_k = 10000, 768, 5
    out = subprocess.run([sys.executable, os.path.join(root, "scripts", "test_rag2.py"), "--batch", "100",
                          "--org-id", "org_1", "--top-k", str(top_k), "--graph", "--json", "--docs", str(n)],
                         capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("[")][-1])
    assert len(res) == 100
    assert set(res[0]) == {"query", "success", "refused", "refusal_reason", "max_score", "contexts", "timings"}
    assert set(res[0]["contexts"][0]) == {"child_id", "document_id", "page", "rrf_score", "rerank_score",
                                          "text", "section"}
    # the CPU leg of configs[0]: the same command line over the oracle-backed client -- same
    # contexts, same float64 RRF scores, query by query (recall@k = 1, order identical)
    from tests import oracle_cli
    rc, cpu = oracle_cli.run(["--batch", "100", "--org-id", "org_1", "--top-k", str(top_k), "--graph", "--json",
                              "--docs", str(n)])
    assert rc == 0 and [r["query"] for r in cpu] == [r["query"] for r in res]
    for a, b in zip(res, cpu):
        assert a["contexts"] == b["contexts"] and (a["success"], a["refused"]) == (b["success"], b["refused"])
    # the oracle pipeline
    x = synth.dense_rows(0, n, d)
    csr, idf, avgdl, v = lexical_fixture(T, n)
    g = synth.build_graph(n)
    names = [f"entity{e}" for e in range(synth.n_entities(n))]
    emb = HashEmbedder(model_dim=4096, store_dim=d)
    for r in res:
        text = r["query"]
        kws = text.split()
        terms = []
        for kw in kws:                              # tokens of the vocabulary "t<id>", first sighting
            if kw.startswith("t") and kw[1:].isdigit() and int(kw[1:]) < v and int(kw[1:]) not in terms:
                terms.append(int(kw[1:]))
        _, Il = O.bm25_topk(csr.rowptr, csr.post_doc, csr.post_tf, csr.doclen, idf, avgdl, [terms], n, 50)
        qv = np.array([emb.embed_query(text)], dtype=np.float32)
        _, Id, _ = CO.dense_topk_exact(x, qv, 100)
        per, seeds = max(1, 50 // len(kws)), []     # graph_search.py:161-170: name ILIKE %kw%, 5 keywords
        for kw in kws[:5]:
            hits = 0
            for e, name in enumerate(names):
                if kw.lower() in name:
                    if e not in seeds:
                        seeds.append(e)
                    hits += 1
                    if hits == per:
                        break
        seeds = seeds[:16]
        Ig = [[]]
        if seeds:
            _, Ig = O.graph_topk(g.ent_rowptr, g.ent_col, g.men_rowptr, g.men_chunk, g.men_conf,
                                 np.array([seeds + [-1] * (16 - len(seeds))], dtype=np.int32), 2, n, 50)
        ei, es = O.fused_topk_ids(list(Il[0]), list(Id[0]), list(Ig[0]) if len(Ig[0]) else None, top_k)
        assert r["success"] and not r["refused"]
        assert [c["child_id"] for c in r["contexts"]] == [f"c{i}" for i in ei], text
        assert [c["rrf_score"] for c in r["contexts"]] == es
        assert {"planning", "retrieval", "fusion", "expansion", "safety"} <= set(r["timings"])


def test_bench_multi_rank_control_flow():
    """bench.py --gpus N end to end on one GPU (THR_BENCH_REHEARSAL: all ranks on cuda:0, gloo in
    place of RCCL): document sharding, replicas, exchange, merge and the single JSON line of rank 0,
    which at N > 1 always carries the north star's layout (``config.strong_doc_sharded``: the corpus
    cut into N document shards, one batch served by all N GPUs) next to whatever layout ran."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, THR_BENCH_REHEARSAL="1")
    port = 29800 + os.getpid() % 1000
    for gpus, extra, label, pipeline in (
            (2, [], "doc-shard x2", "dense"),
            (2, ["--doc-shards", "1"], "doc-shard x1 x 2 replicas", "dense"),
            (2, ["--config", "triple_rerank", "--token-docs", "20000"], "doc-shard x2", "triple_rerank"),
            (4, ["--doc-shards", "2", "--config", "dense_bm25"], "doc-shard x2 x 2 replicas", "dense_bm25"),
            (2, ["--preset", "configs3"], "doc-shard x2", "triple")):     # configs[3]'s layout, one flag away
        out = subprocess.run(
            [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
             "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"),
             "--gpus", str(gpus), "--steps", "2", "--warmup", "1", "--docs", "40000", "--queries", "192",
             "--no-extras", "--no-cpu-baseline"] + extra,
            env=env, cwd=root, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1
        rec = json.loads(lines[0])
        assert rec["n_gpus"] == gpus and rec["steps"] == 2 and rec["value"] > 0
        assert rec["config"]["parallelism"].startswith(label) and rec["config"]["rescued_queries"] == 0
        replicas = "replicas" in label
        assert rec["scaling"] == ("weak" if replicas else "strong")
        # where a rank's share of the step goes (scan / exchange / the rest)
        assert set(rec["config"]["per_rank_ms"]) >= {"step_ms", "scan_ms", "exchange_ms", "fixed_ms"}
        assert rec["config"]["collective_backend"] == "gloo" and rec["config"]["world_size_seen"] == gpus
        assert rec["config"]["pipeline"] == pipeline
        strong = rec["config"]["strong_doc_sharded"]
        assert strong["layout"] == f"doc-shard x{gpus}" and strong["scaling"] == "strong"
        assert strong["ms_per_step"] > 0 and strong["value"] > 0 and strong["same_run_as_the_headline"] == (not replicas)
        assert set(strong["per_rank_ms"]) >= {"step_ms", "scan_ms", "exchange_ms", "fixed_ms"}
        assert strong["per_rank_ms"]["shard_docs"] == 40000 // gpus
        if not replicas:
            assert strong["value"] == rec["value"] and strong["ms_per_step"] == rec["ms_per_step"]
        port += 1


def test_tool_layer_over_a_gpu_index_client(T, golden):
    """8f.2 on the GPU: the agent tool ``search_knowledge_base()`` (reference
    src/voice_agent/tools/crm_knowledge.py:26-182; tests/test_rag2_tool_connection.py:75-330) with
    NOTHING injected -- tenant discovery through the registered GpuIndexClient, the default
    embedder / planner, RAG2Retriever.retrieve over the kernels, MaxSim rerank -- returns the
    dict schema of tests/golden/tool_layer.json with the oracle pipeline's chunks, ranks and
    rounded scores; refusals and a category (= collection) filter map as in the reference."""
    from triple_hybrid_rag_amd import synth
    from triple_hybrid_rag_amd import backend
    from triple_hybrid_rag_amd.backend import CorpusStore, GpuIndexClient
    from triple_hybrid_rag_amd.config import SETTINGS
    from triple_hybrid_rag_amd.rag2.embedder import HashEmbedder
    from triple_hybrid_rag_amd.tools import crm_knowledge as crm

    n, d = 20000, 768
    x = synth.dense_rows(0, n, d)
    csr, idf, avgdl, v = lexical_fixture(T, n)
    dtok = synth.doc_tokens(0, n, 32, 64)
    idx = (T.GpuIndex().set_dense(x)
           .set_lexical(csr.rowptr, csr.post_doc, csr.post_tf, csr.doclen, idf, avgdl).set_tokens(dtok))
    store = CorpusStore.synthetic(n, vocab_size=v)
    store.collections = ["faq" if i % 5 == 0 else "pricing" for i in range(n)]

    class TokEmb:
        def embed_query_tokens(self, query):
            return synth.query_tokens(1, 32, 64)[0]

    client = GpuIndexClient(idx, store, org_id="org-7", token_embedder=TokEmb())
    saved = dict(SETTINGS.__dict__)
    gold = golden("tool_layer.json")
    ok_keys = set(next(c["out"] for c in gold if c["out"].get("results")).keys())
    row_keys = set(next(c["out"]["results"][0] for c in gold if c["out"].get("results")).keys())
    refused_keys = set(next(c["out"] for c in gold if c["out"].get("refused")).keys())
    try:
        backend.set_default_client(client)
        SETTINGS.rag2_enabled = True
        SETTINGS.rag2_rerank_enabled = True
        SETTINGS.rag2_graph_enabled = False
        SETTINGS.rag2_embed_dim_store = d
        SETTINGS.rag2_safety_threshold = 0.0
        SETTINGS.rag2_denoise_alpha = 0.0
        text = "t100 t2000 t77"
        out = crm.search_knowledge_base(text, None, 5)
        assert set(out.keys()) == ok_keys and out["success"] and out["search_type"] == "rag2_triple_hybrid"
        assert out["query"] == text and out["category"] is None and out["result_count"] == len(out["results"]) == 5
        assert all(set(r.keys()) == row_keys for r in out["results"])
        # (the keys retrieve() sets, reference rag2/retrieval.py:153-191)
        assert set(out["timings_ms"]) == {"planning", "retrieval", "fusion", "expansion", "rerank", "safety"}
        assert all(isinstance(t, float) and t >= 0 for t in out["timings_ms"].values())
        # the oracle pipeline for the same query: dense + BM25 -> RRF -> top 20 -> MaxSim order
        qv = np.asarray(HashEmbedder(store_dim=d).embed_query(text), dtype=np.float32)
        _, Id, _ = CO.dense_topk_exact(x, qv[None], 100)
        tids = [store.vocab[w] for w in text.split()]
        _, Il = O.bm25_topk(csr.rowptr, csr.post_doc, csr.post_tf, csr.doclen, idf, avgdl, [tids], n, 50)
        f20, s20 = O.fused_topk_ids(list(Il[0]), list(Id[0]), None, SETTINGS.rag2_rerank_top_k)
        ms = CO.maxsim(synth.query_tokens(1, 32, 64), dtok, np.array([f20], dtype=np.int32))[0] / 32.0
        order = O.rerank_order([float(s) for s in ms])[:5]
        gaps = np.abs(np.diff(np.sort(ms)[::-1][:6]))
        if gaps.min() > 1e-4:   # (the order is pinned wherever the oracle's scores are apart)
            assert [r["chunk_id"] for r in out["results"]] == [f"c{f20[j]}" for j in order]
        for pos, r in enumerate(out["results"]):
            doc = int(r["chunk_id"][1:])
            j = f20.index(doc)
            assert r["relevance_rank"] == pos + 1 and r["parent_id"] == f"p{doc // 4}"
            assert r["similarity_score"] == round(s20[j], 4)
            assert abs(r["rerank_score"] - round(float(ms[j]), 4)) <= 1.01e-4
            assert r["content"] == f"parent text {doc // 4}" and r["title"] == f"Section {doc // 4}"
            assert r["lexical_rank"] == (list(Il[0]).index(doc) + 1 if doc in Il[0] else None)
            assert r["semantic_rank"] == (list(Id[0]).index(doc) + 1 if doc in Id[0] else None)
            assert r["graph_rank"] is None and r["is_table"] is False and r["category"] is None
        assert abs(out["max_rerank_score"] - round(float(ms.max()), 4)) <= 1.01e-4
        # category = the collection filter of the RPCs, applied on the device before the ranking
        faq = crm.search_knowledge_base(text, "faq", 3)
        assert faq["category"] == "faq" and faq["result_count"] == 3
        assert all(int(r["chunk_id"][1:]) % 5 == 0 and r["category"] == "faq" for r in faq["results"])
        # below the threshold -> refusal mapping (success stays True)
        SETTINGS.rag2_safety_threshold = 0.999
        ref = crm.search_knowledge_base(text, None, 5)
        assert set(ref.keys()) == refused_keys and ref["refused"] and ref["success"] and ref["results"] == []
        assert ref["refusal_reason"].startswith("Max score") and "below threshold 0.999" in ref["refusal_reason"]
        # the tenant the tool discovers is the index's: another client id sees nothing
        SETTINGS.rag2_safety_threshold = 0.0
        other = crm._search_knowledge_base_rag2(text, None, 5, org_id="someone-else")
        assert other["refused"] and other["refusal_reason"] == "No candidates found"
        # never raises into the agent
        backend.set_default_client(None)
        err = crm.search_knowledge_base(text, None, 5)
        assert set(err.keys()) == {"error", "query", "category"} and err["error"].startswith("Database error")
    finally:
        SETTINGS.__dict__.update(saved)
        backend.set_default_client(None)


def test_standalone_fusion_on_the_device(T, golden):
    """8f.3: the standalone package's RRFFusion as device ops -- thr_rrf_fuse_standalone (``fuse``
    and ``fuse_two_channels`` arithmetic: weight * (1 / (60 + rank)), per-occurrence adds) and
    thr_fuse_post (safety threshold, numpy-percentile denoise, [:top_k], min-max normalise) --
    against the reference's own outputs (tests/golden/standalone_fusion.json <-
    triple-hybrid-rag/src/triple_hybrid_rag/core/fusion.py:52-318), bit for bit."""
    from triple_hybrid_rag_amd.config import RAGConfig
    from triple_hybrid_rag_amd.core.fusion import RRFFusion
    g = golden("standalone_fusion.json")

    def table(rows_lists):
        ids = {}
        for rows in rows_lists:
            for r in rows:
                ids.setdefault(r["chunk_id"], len(ids) + 100)
        return ids

    def tensors(rows, ids, key):
        if not rows:
            return None, None
        i = torch.tensor([[ids[r["chunk_id"]] for r in rows]], dtype=torch.int64, device="cuda")
        s = torch.tensor([[float(r.get(key, 0.0)) for r in rows]], dtype=torch.float64, device="cuda")
        return i, s

    for c in g["fuse"]:
        ids = table([c["lexical"], c["semantic"], c["graph"]])
        back = {v: k for k, v in ids.items()}
        li, ls = tensors(c["lexical"], ids, "lexical_score")
        si, ss = tensors(c["semantic"], ids, "semantic_score")
        gi, gs = tensors(c["graph"], ids, "graph_score")
        fus = RRFFusion(RAGConfig(rag_safety_threshold=c["safety_threshold"],
                                  rag_denoise_enabled=c["denoise_enabled"],
                                  rag_denoise_alpha=c["denoise_alpha"]))
        oi, os_, oc = fus.fuse_batch(li, si, gi, ls, ss, gs, weights=c["weights"], top_k=c["top_k"])
        m = int(oc[0])
        assert [back[int(i)] for i in oi[0, :m]] == [r["chunk_id"] for r in c["out"]]
        assert [float(s) for s in os_[0, :m]] == [r["rrf_score"] for r in c["out"]]
        assert bool((oi[0, m:] == -1).all())
    for c in g["two"]:
        ids = table([c["a"], c["b"]])
        back = {v: k for k, v in ids.items()}
        ai, _ = tensors(c["a"], ids, "x")
        bi, _ = tensors(c["b"], ids, "x")
        oi, os_, _, oc = T._native.rrf_fuse_standalone(ai, bi, None, c["top_k"] or 256, c["wa"], c["wb"], 0.0,
                                                       two_channels=True)
        m = int(oc[0])
        assert [{"chunk_id": back[int(i)], "rrf_score": float(s)} for i, s in zip(oi[0, :m], os_[0, :m])] == c["out"]
    for c in g["normalize"]:
        n = len(c["in"])
        if n == 0:
            continue
        sc = torch.tensor([c["in"]], dtype=torch.float64, device="cuda")
        di = torch.arange(n, dtype=torch.int64, device="cuda")[None]
        _, os_, oc = T._native.fuse_post(di, sc, normalize=True)
        assert int(oc[0]) == n and [float(s) for s in os_[0]] == c["out"]
    # a batch of lists with padding: every row equals its own one-row call
    ci = torch.tensor([[5, 9, 7, -1], [3, -1, -1, -1], [8, 6, 4, 2]], dtype=torch.int64, device="cuda")
    cj = torch.tensor([[9, 1, -1], [3, 2, 1], [-1, -1, -1]], dtype=torch.int64, device="cuda")
    bi, bs, br, bc = T._native.rrf_fuse_standalone(ci, cj, None, 7, 0.7, 0.8, 1.0)
    for q in range(3):
        oi, os_, _, oc = T._native.rrf_fuse_standalone(ci[q:q + 1].contiguous(), cj[q:q + 1].contiguous(), None, 7,
                                                       0.7, 0.8, 1.0)
        assert torch.equal(bi[q], oi[0]) and torch.equal(bs[q], os_[0]) and int(bc[q]) == int(oc[0])
