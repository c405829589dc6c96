#!/usr/bin/env python3
"""Headline benchmark: queries/sec for the fused top-10 of the RAG 2.0 retrieval hot path.

    python bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one batch of ``--queries`` synthetic queries:
dense brute-force cosine top-100 over the HBM-resident corpus -> (all-gather + merge when
N > 1) -> weighted RRF -> fused top-10.  Workload at N = 1 is BASELINE.json configs[1]
(1M-doc / 768-d dense-only top-10).  For N > 1 the SAME corpus is sharded by document
across the ranks (strong scaling, one process per GPU, RCCL all-gather of the per-shard
top-100).  Inputs are resident in HBM before the timed region.

The JSON line also carries
  roofline      the dense scan kernel (dense_scan<MODE_FILTER>) timed with HIP events on its
                own stream through thr_dense_scan_probe: algorithmic bytes
                (tiles * n_docs * dim * 4 per launch) / average launch time vs 8 TB/s
  cpu_baseline  the CPU oracle's fast path (float32 BLAS shortlist + float64 rescoring,
                oracle/thr_oracle.py dense_topk_fast) timed on this host's cores on a bounded
                sample of the same workload (rank 0, N = 1 only)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--docs", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--queries", type=int, default=1024, help="queries per step (batch)")
    ap.add_argument("--top-k", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-queries", type=int, default=64)
    ap.add_argument("--probe-reps", type=int, default=5)
    ap.add_argument("--no-f16-extra", action="store_true",
                    help="skip the additional float16-shortlist measurement reported next to the "
                         "default float32-scan one")
    ap.add_argument("--shortlist", choices=("f32", "f16", "f16-inline"), default="f32",
                    help="f16: opt-in float16 shortlist copy for the streaming pass (results stay "
                         "float64-exact; see DESIGN.md 4.1b)")
    return ap.parse_args()


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    import triple_hybrid_rag_amd as T
    from triple_hybrid_rag_amd import synth
    from triple_hybrid_rag_amd.distributed import ShardedIndex, shard_range

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with torch.distributed.run (one process per GPU)")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))

    # ---- inputs (deterministic, identical for every world size) ----
    lo, hi = shard_range(args.docs, rank, world)
    t0 = time.time()
    docs = synth.dense_rows(lo, hi - lo, args.dim)
    queries = synth.dense_queries(args.queries, args.dim, args.docs)
    gen_s = time.time() - t0
    index = T.GpuIndex(doc_base=lo).set_dense(docs, shortlist=args.shortlist)
    sharded = ShardedIndex(index)
    qd = torch.from_numpy(queries).cuda()
    torch.cuda.synchronize()

    def step():
        return sharded.retrieve_batch(qd, top_k=args.top_k)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def measure():
        """W warm-up steps, then K timed steps between barriers; max over ranks."""
        res, rescued = None, 0
        for _ in range(args.warmup):
            res = step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            res = step()
            rescued += res.rescued
        barrier()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return res, rescued, elapsed

    def probe_scan(f16):
        """Average launch time of the streaming scan kernel alone (HIP events on its stream)."""
        stream = torch.cuda.current_stream()
        index.scan_probe(qd)  # warm; same workspace (and tau) as the timed searches
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record(stream)
        for _ in range(args.probe_reps):
            index.scan_probe(qd)
        ev1.record(stream)
        torch.cuda.synchronize()
        ms = ev0.elapsed_time(ev1) / args.probe_reps
        if f16:
            qt, elem, name = (64 if args.dim <= 768 else 32), 2, "dense_scan_f16"
            if index.docs16 is None:
                elem, name = 4, "dense_scan_f16<F32IN>"
        else:
            qt = 16 if os.environ.get("THR_DENSE_QT") == "16" else 32
            elem = 4
            name = "dense_scan" if os.environ.get("THR_DENSE_IMPL", "m")[0] == "v" else "dense_scan_mfma2"
        tiles = (args.queries + qt - 1) // qt
        alg = tiles * (hi - lo) * args.dim * elem
        ach = alg / (ms * 1e-3) / 1e9
        return {"bound": "hbm", "kernel": f"{name}<dim={args.dim},MODE_FILTER> ({qt} queries/pass)",
                "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBPS, 4), "traffic": None, "launch_ms": round(ms, 4),
                "tile_passes_per_launch": tiles, "algorithmic_bytes_per_launch": alg}, qt, tiles

    res, rescued, elapsed = measure()
    qps = args.steps * args.queries / elapsed

    # ---- roofline of the dominant kernel: HIP events around the scan alone ----
    n_local = hi - lo
    f16 = args.shortlist != "f32"
    roofline, qt, tiles = probe_scan(f16)

    def pmc_traffic(kind, qt_expected, ntiles):
        """HBM bytes per launch from the committed PMC pass (FETCH_SIZE x2 on gfx950), scaled per
        tile pass; only valid for the shape it was measured on."""
        try:
            with open(os.path.join(ROOT, "profiles", "r1_dense_scan_traffic.json")) as f:
                pm = json.load(f)
            if pm["n_docs"] == n_local and pm["dim"] == args.dim and qt_expected:
                return round(pm[kind]["hbm_bytes_per_tile_pass"] * ntiles)
        except (OSError, KeyError, ValueError):
            pass
        return None

    roofline["traffic"] = pmc_traffic("f16" if f16 else "f32", qt == (64 if f16 else 32), tiles)

    # ---- the opt-in float16 shortlist copy, measured in the same run for comparison ----
    extra = None
    if not f16 and not args.no_f16_extra and args.dim in (512, 768, 1024):
        ids_f32 = res.ids.clone()
        index.docs16, index.doc_rel_err = T._native.dense_quantize_f16(index.docs)
        res16, rescued16, elapsed16 = measure()
        roof16, qt16, tiles16 = probe_scan(True)
        roof16["traffic"] = pmc_traffic("f16", qt16 == 64, tiles16)
        extra = {"value": round(args.steps * args.queries / elapsed16, 1), "unit": "queries/s",
                 "ms_per_step": round(1e3 * elapsed16 / args.steps, 3),
                 "rescued_queries": rescued16, "roofline": roof16,
                 "fused_top10_identical_to_f32_scan": bool(torch.equal(res16.ids, ids_f32)),
                 "note": "float16 copy of the rows for the streaming pass only; scores are the "
                         "float64 rescoring of float32 rows, certificate covers quantisation"}
        index.docs16, index.doc_rel_err = None, 0.0

    # ---- exactness of what was timed + CPU baseline (rank 0, N = 1) ----
    cpu = None
    check = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import thr_oracle as O
        torch.set_num_threads(os.cpu_count() or 1)
        nq_cpu = min(args.cpu_queries, args.queries)
        dn = index.dnorm.cpu().numpy()
        t0 = time.perf_counter()
        Sc, Ic = O.dense_topk_fast(docs, queries[:nq_cpu], 100, dnorm=dn)
        fused = [O.fused_topk_ids(None, list(i), None, args.top_k)[0] for i in Ic]
        cpu_s = time.perf_counter() - t0
        cpu = {"value": round(nq_cpu / cpu_s, 2), "unit": "queries/s", "cores": os.cpu_count(),
               "kind": "port",
               "sample": f"{nq_cpu} of the {args.queries} queries over the full "
                         f"{args.docs}x{args.dim} corpus (fp32 BLAS shortlist + fp64 rescoring "
                         f"+ RRF), {cpu_s:.1f} s"}
        ids = res.ids.cpu().numpy()
        same = sum(list(ids[i]) == fused[i] for i in range(nq_cpu))
        check = {"fused_top10_identical": f"{same}/{nq_cpu}", "recall_at_10": round(float(np.mean(
            [len(set(ids[i]) & set(fused[i])) / args.top_k for i in range(nq_cpu)])), 4)}

    if rank == 0:
        out = {
            "metric": "queries/sec (fused top-10)", "value": round(qps, 1), "unit": "queries/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None,
            "dtype": ("f16 shortlist scan" if f16 else "f32 scan") + " + f64 rescoring of f32 rows",
            "data": "synthetic",
            "config": {"workload": f"{args.docs}-doc / {args.dim}-d dense-only brute-force cosine "
                                   f"top-{args.top_k} (BASELINE.json configs[1])",
                       "docs": args.docs, "dim": args.dim, "queries_per_step": args.queries,
                       "semantic_top_k": 100, "fused_top_k": args.top_k, "shortlist": args.shortlist,
                       "parallelism": f"doc-shard x{world}" if world > 1 else "single GPU",
                       "rescued_queries": rescued, "input_gen_s": round(gen_s, 1)},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        if extra:
            out["f16_shortlist"] = extra
        if check:
            out["parity_check"] = check
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
