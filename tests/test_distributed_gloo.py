"""N > 1 path on CPU: two gloo ranks shard a corpus by document, each produces its exact
per-shard top-k (CPU oracle standing in for the kernels, which need a GPU), the product's
``gather_topk`` all-gathers them, and the merged result must equal the unsharded top-k --
i.e. the sharding + exchange step is correct by construction."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n, d, k, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import thr_oracle as O
    from triple_hybrid_rag_amd import synth
    from triple_hybrid_rag_amd.distributed import gather_topk, shard_range
    lo, hi = shard_range(n, rank, world)
    docs = synth.dense_rows(lo, hi - lo, d)          # shard == same rows of the global corpus
    q = synth.dense_queries(5, d, n)
    S, I = O.dense_topk_exact(docs, q, k, doc_id_base=lo)
    S = torch.from_numpy(np.stack(S))
    I = torch.from_numpy(np.stack(I))
    Sg, Ig = gather_topk(S, I)
    assert Sg.shape == (world, 5, k)
    # several channels in ONE collective: the same lists as one collective per channel gives
    from triple_hybrid_rag_amd.distributed import gather_topk_many
    S2, I2 = S[:, : k // 2].contiguous() * 2.0, I[:, : k // 2].contiguous() + 7
    S3, I3 = S[:, :3].contiguous() - 1.0, I[:, :3].contiguous()
    many = gather_topk_many([(S, I), (S2, I2), (S3, I3)])
    for (ms_, mi_), (s_, i_) in zip(many, [(S, I), (S2, I2), (S3, I3)]):
        es_, ei_ = gather_topk(s_, i_)
        assert ms_.shape == es_.shape and torch.equal(ms_, es_) and torch.equal(mi_, ei_)
        assert ms_.stride(2) == 1 and ms_.stride(1) == s_.shape[1] and ms_.stride(0) == mi_.stride(0)
    if rank == 0:
        np.save(os.path.join(out_dir, "S.npy"), Sg.numpy())
        np.save(os.path.join(out_dir, "I.npy"), Ig.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_gather_merge(tmp_path):
    n, d, k, world = 3001, 64, 20, 2
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(world, port, n, d, k, str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, ROOT)
    from oracle import thr_oracle as O
    from triple_hybrid_rag_amd import synth
    from triple_hybrid_rag_amd.distributed import shard_range
    Sg, Ig = np.load(tmp_path / "S.npy"), np.load(tmp_path / "I.npy")
    q = synth.dense_queries(5, d, n)
    Se, Ie = O.dense_topk_exact(synth.dense_rows(0, n, d), q, k)
    for qi in range(5):
        ms, mi = O.topk_desc(Sg[:, qi].ravel(), k, Ig[:, qi].ravel())
        assert np.array_equal(mi, Ie[qi]) and np.array_equal(ms, Se[qi])
    # shard ranges tile the corpus exactly
    cover = [shard_range(1_000_003, r, 8) for r in range(8)]
    assert cover[0][0] == 0 and cover[-1][1] == 1_000_003
    assert all(a[1] == b[0] for a, b in zip(cover, cover[1:]))


def _worker_2d(rank, world, port, n, d, k, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import thr_oracle as O
    from triple_hybrid_rag_amd import synth
    from triple_hybrid_rag_amd.distributed import gather_topk, layout_2d, replica_groups, shard_range
    doc_shards = 2
    shard, replica, members = layout_2d(rank, world, doc_shards)
    group = replica_groups(world, doc_shards)[replica]
    assert rank in members and len(members) == doc_shards
    lo, hi = shard_range(n, shard, doc_shards)
    docs = synth.dense_rows(lo, hi - lo, d)
    q = synth.dense_queries(6, d, n)[replica::world // doc_shards]   # this replica's queries
    S, I = O.dense_topk_exact(docs, q, k, doc_id_base=lo)
    Sg, Ig = gather_topk(torch.from_numpy(np.stack(S)), torch.from_numpy(np.stack(I)), group)
    assert Sg.shape == (doc_shards, 3, k)
    if shard == 0:
        np.save(os.path.join(out_dir, f"S{replica}.npy"), Sg.numpy())
        np.save(os.path.join(out_dir, f"I{replica}.npy"), Ig.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_doc_shards_times_replicas_layout(tmp_path):
    """4 ranks = 2 document shards x 2 replicas: each replica answers its own queries from the
    whole corpus; the exchange stays inside the replica."""
    n, d, k, world = 2501, 64, 15, 4
    port = 31500 + os.getpid() % 2000
    mp.spawn(_worker_2d, args=(world, port, n, d, k, str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, ROOT)
    from oracle import thr_oracle as O
    from triple_hybrid_rag_amd import synth
    from triple_hybrid_rag_amd.distributed import layout_2d
    q = synth.dense_queries(6, d, n)
    Se, Ie = O.dense_topk_exact(synth.dense_rows(0, n, d), q, k)
    for replica in range(2):
        Sg, Ig = np.load(tmp_path / f"S{replica}.npy"), np.load(tmp_path / f"I{replica}.npy")
        for j, qi in enumerate(range(replica, 6, 2)):
            ms, mi = O.topk_desc(Sg[:, j].ravel(), k, Ig[:, j].ravel())
            assert np.array_equal(mi, Ie[qi]) and np.array_equal(ms, Se[qi])
    assert layout_2d(5, 8, 2) == (1, 2, [4, 5]) and layout_2d(3, 8, 8) == (3, 0, list(range(8)))
    with pytest.raises(ValueError):
        layout_2d(0, 8, 3)


def _worker_rerank(rank, world, port, n, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import thr_oracle as O
    from triple_hybrid_rag_amd import synth
    from triple_hybrid_rag_amd.distributed import gather_rows, shard_range
    lo, hi = shard_range(n, rank, world)
    dtok = synth.doc_tokens(lo, hi - lo, 32, 32)            # this shard's token matrices
    qtok = synth.query_tokens(4, 32, 32)
    cand = np.random.default_rng(5).integers(0, n, (4, 20)).astype(np.int64)   # fused ids: same on every rank
    local = np.where((cand >= lo) & (cand < hi), cand - lo, -1)
    ms = O.maxsim_scores(qtok, dtok, local).astype(np.float32)   # -inf where another shard owns the doc
    g = gather_rows(torch.from_numpy(ms))
    assert g.shape == (world, 4, 20)
    if rank == 0:
        np.save(os.path.join(out_dir, "ms.npy"), g.numpy())
        np.save(os.path.join(out_dir, "cand.npy"), cand)
    dist.barrier()
    dist.destroy_process_group()


def test_rerank_exchange_second_all_gather(tmp_path):
    """configs[4]'s rerank leg: candidates are scored on their owning shard, a second
    all-gather brings the [nq, n] score lists together, and the maximum per candidate is the
    unsharded MaxSim score -- so the stable rerank sort gives the unsharded order."""
    n, world = 600, 3
    port = 33500 + os.getpid() % 2000
    mp.spawn(_worker_rerank, args=(world, port, n, str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, ROOT)
    from oracle import thr_oracle as O
    from triple_hybrid_rag_amd import synth
    g, cand = np.load(tmp_path / "ms.npy"), np.load(tmp_path / "cand.npy")
    full = O.maxsim_scores(synth.query_tokens(4, 32, 32), synth.doc_tokens(0, n, 32, 32), cand)
    merged = g.max(axis=0)
    assert np.isfinite(merged).all() and (np.isfinite(g).sum(axis=0) == 1).all()
    assert np.array_equal(merged, full.astype(np.float32))
    for q in range(4):
        assert O.rerank_order(list(merged[q])) == O.rerank_order(list(full[q].astype(np.float32)))


def _worker_floor_agreement(rank, world, port, out_dir):
    import os
    import sys
    import types
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from triple_hybrid_rag_amd.distributed import ShardedIndex
    from triple_hybrid_rag_amd.index import floor_width
    out = {}
    # every shard on an f16 scan: the floor is on, and the exchange hands back all the shards' tiles
    sh = ShardedIndex(types.SimpleNamespace(shortlist="f16", device=torch.device("cpu")))
    out["all_f16"] = sh.floor and sh._floor_exchange() is not None
    fx, g = sh._floor_exchange()
    lb = torch.full((5, floor_width(100, world)), float(rank), dtype=torch.float32)
    got = fx(lb)
    out["tiles"] = list(got.shape) == [world, 5, floor_width(100, world)] and \
        all(bool(torch.all(got[r] == float(r))) for r in range(world)) and g == world
    # ONE shard fell back to the float32 scan (rows outside the float16 range): nobody uses the
    # floor -- a rank that skipped the exchange while the others entered it would hang the group
    sh = ShardedIndex(types.SimpleNamespace(shortlist="f32" if rank == 1 else "f16", device=torch.device("cpu")))
    out["one_f32"] = (not sh.floor) and sh._floor_exchange() is None
    # switched off by the caller: no collective at construction either
    sh = ShardedIndex(types.SimpleNamespace(shortlist="f16", device=torch.device("cpu")), floor=False)
    out["off"] = sh._floor_exchange() is None
    import json
    with open(os.path.join(out_dir, f"floor_{rank}.json"), "w") as f:
        json.dump(out, f)
    dist.barrier()
    dist.destroy_process_group()


def test_shards_agree_on_the_floor_exchange(tmp_path):
    """ShardedIndex splits the dense channel around one more all-gather only if EVERY shard of the
    group can (the scan a shard runs depends on its own rows): decided once, collectively."""
    import json
    world = 3
    port = 29500 + os.getpid() % 1000 + 7
    mp.spawn(_worker_floor_agreement, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        got = json.load(open(tmp_path / f"floor_{r}.json"))
        assert got == {"all_f16": True, "tiles": True, "one_f32": True, "off": True}, (r, got)


def test_floor_width():
    from triple_hybrid_rag_amd.index import floor_width
    assert floor_width(100, 8) == 26 and floor_width(100, 2) == 100 and floor_width(10, 8) == 16
    assert floor_width(256, 1) == 256                       # never above THR_DENSE_MAX_K
    for g in (1, 2, 3, 8, 64, 256, 1000):
        for k in (1, 10, 100, 256):
            m = floor_width(k, g)
            assert 1 <= m <= 256 and g * m <= max(4096, g)  # the band kernel's LDS holds the G * m values
