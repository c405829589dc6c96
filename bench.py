#!/usr/bin/env python3
"""Headline benchmark: queries/sec for the fused top-10 of the RAG 2.0 retrieval hot path.

    python bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one batch of ``--queries`` (default 1536) synthetic queries:
dense brute-force cosine top-100 over the HBM-resident corpus -> (all-gather + merge when
N > 1) -> weighted RRF -> fused top-10.  Workload at N = 1 is BASELINE.json configs[1]
(1M-doc / 768-d dense-only top-10).  For N > 1 the SAME corpus is sharded by document
across the ranks (strong scaling, one process per GPU, RCCL all-gather of the per-shard
top-100).  Inputs are resident in HBM before the timed region.

The JSON line also carries
  roofline         the dense scan kernel of the timed path, launched alone through
                   thr_dense_scan_probe[_f16] and timed with HIP events on its own stream:
                   2*N*D*Q flops per launch / average launch time against the dense MFMA peak of
                   its input type, the PMC HBM bytes per launch (`traffic`), and SURVEY 8(d)'s
                   passes*N*D*bytes/t figure against 8 TB/s (`hbm_formula`)
  other_shortlists the same step with the other shortlist scans (same bits out)
  cpu_baseline     the CPU oracle's fast path (float32 BLAS shortlist + float64 rescoring,
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

# MI355X peaks (MI355X_MICROARCH.md): HBM3E spec; dense f16 MFMA; f32-input MFMA (= f32 vector)
HBM_PEAK_GBPS = 8000.0
MFMA_F16_PEAK_TFLOPS = 2500.0
MFMA_F32_PEAK_TFLOPS = 157.3


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--docs", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--queries", type=int, default=2048,
                    help="queries per step (batch): 2048 = 8 workgroup tiles of 256 queries x 4 "
                         "row slices per XCD, one full round of the 256 CUs for the default scan")
    ap.add_argument("--top-k", type=int, default=10)
    ap.add_argument("--doc-shards", type=int, default=0,
                    help="N > 1: split the corpus into this many document shards (default N: the "
                         "pure document-sharded layout); the N / doc-shards groups are replicas "
                         "that serve different query batches (distributed.layout_2d)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-queries", type=int, default=2048)
    ap.add_argument("--probe-reps", type=int, default=5)
    ap.add_argument("--no-extras", "--no-f16-extra", dest="no_extras", action="store_true",
                    help="skip the measurements of the other shortlist flavours reported next to "
                         "the primary one")
    ap.add_argument("--shortlist", choices=("auto", "f16-inline", "f16", "f32"), default="auto",
                    help="shortlist scan of the timed path (GpuIndex.set_dense): auto = f16 copy; "
                         "results are the same float64-exact bits in every flavour (DESIGN.md 4.1)")
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
    # THR_BENCH_REHEARSAL=1: all ranks on cuda:0 over gloo -- exercises the N > 1 control flow
    # (sharding, groups, exchange through the host, merge) on a one-GPU box; not a measurement
    rehearsal = os.environ.get("THR_BENCH_REHEARSAL") == "1"
    torch.cuda.set_device(0 if rehearsal else local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))

    # ---- inputs (deterministic, identical for every world size) ----
    from triple_hybrid_rag_amd.distributed import layout_2d, replica_groups
    doc_shards = args.doc_shards or world
    shard, replica, _ = layout_2d(rank, world, doc_shards)
    n_replicas = world // doc_shards
    group = None
    if world > 1 and n_replicas > 1:
        group = replica_groups(world, doc_shards)[replica]
    lo, hi = shard_range(args.docs, shard, doc_shards)
    t0 = time.time()
    docs = synth.dense_rows(lo, hi - lo, args.dim)
    queries = synth.dense_queries(args.queries * n_replicas, args.dim, args.docs)
    queries = np.ascontiguousarray(queries[replica::n_replicas])   # this replica's batch
    gen_s = time.time() - t0
    index = T.GpuIndex(doc_base=lo).set_dense(docs, shortlist=args.shortlist)
    index.reserve(args.queries, 100)   # workspaces are part of the resident index, not of a step
    sharded = ShardedIndex(index, group=group)
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
        res, counters = None, []
        for _ in range(args.warmup):
            res = step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            res = step()
            counters.append(res.rescued)   # device counters: a step has no host read-back
        barrier()
        elapsed = time.perf_counter() - t0
        rescued = sum(int(c) for c in counters)
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return res, rescued, elapsed

    FLAVOURS = ("f16-inline", "f16", "f32")
    KERNEL = {"f32": f"dense_scan_mfma2<dim={args.dim},MODE_FILTER>",
              "f16": f"dense_scan_f16q{'s' if args.dim <= 768 else ''}<dim={args.dim},MODE_FILTER> "
                     "(queries in registers, fragment-major f16 rows through LDS-DMA)",
              "f16-inline": f"dense_scan_f16<dim={args.dim},MODE_FILTER> (float32 rows rounded in flight)"}

    def set_flavour(name):
        """Switch the shortlist scan of the (already resident) index in place."""
        index.shortlist = name
        if name == "f32":
            index.docs16, index.doc_rel_err = None, 0.0
        else:
            index.docs16, index.doc_rel_err = T._native.dense_quantize_f16(
                index.docs, keep_copy=name == "f16")

    def pmc(name):
        """Counters of the scan kernel from the committed --pmc passes (profiles/): HBM-side
        bytes per launch (FETCH_SIZE x2 on gfx950) and the MFMA pipe's busy share.  Only
        valid for the shape they were measured on."""
        try:
            with open(os.path.join(ROOT, "profiles", "r1_dense_scan_traffic.json")) as f:
                pm = json.load(f)
            e = pm["flavours"][name]
            if pm["n_docs"] == n_local and pm["dim"] == args.dim and pm["queries"] == args.queries:
                return e
        except (OSError, KeyError, ValueError):
            pass
        return {}

    def probe_scan(name):
        """The streaming scan kernel alone: average launch time from HIP events on its stream."""
        stream = torch.cuda.current_stream()
        index.scan_probe(qd)  # warm; same workspace (and tau) as the timed searches
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record(stream)
        for _ in range(args.probe_reps):
            index.scan_probe(qd)
        ev1.record(stream)
        torch.cuda.synchronize()
        ms = ev0.elapsed_time(ev1) / args.probe_reps
        f16 = name != "f32"
        qt = 32 if name == "f32" else T._native.dense_f16_query_tile(args.dim, name == "f16",
                                                                      args.queries)
        passes = (args.queries + qt - 1) // qt
        flops = 2.0 * n_local * args.dim * args.queries
        peak = MFMA_F16_PEAK_TFLOPS if f16 else MFMA_F32_PEAK_TFLOPS
        ach = flops / (ms * 1e-3) / 1e12
        row_bytes = args.dim * (2 if name == "f16" else 4)
        alg = passes * n_local * row_bytes
        hbm = alg / (ms * 1e-3) / 1e9
        c = pmc(name)
        return {"bound": "mfma", "kernel": f"{KERNEL[name]} ({qt} queries/pass)",
                "achieved": round(ach, 1), "peak": peak, "unit": "TFLOP/s",
                "frac": round(ach / peak, 4), "traffic": c.get("hbm_bytes_per_launch"),
                "launch_ms": round(ms, 4), "flops_per_launch": flops,
                "mfma_busy": c.get("mfma_busy"),
                # SURVEY 8(d)'s dense-scan figure (passes * N * D * bytes / t against 8 TB/s).  The
                # query tiles of a row slice now share each row through their XCD's L2, so the
                # bytes that reach HBM are `traffic`, not this product: the figure can exceed 1.
                "hbm_formula": {"passes": passes, "algorithmic_bytes_per_launch": alg,
                                "achieved": round(hbm, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                "frac": round(hbm / HBM_PEAK_GBPS, 4)}}

    n_local = hi - lo
    primary = index.shortlist
    res, rescued, elapsed = measure()
    qps = args.steps * args.queries * n_replicas / elapsed   # every replica serves its own batch
    roofline = probe_scan(primary)

    # ---- the other shortlist flavours, measured in the same run for comparison ----
    extras = {}
    if not args.no_extras and args.dim in (512, 768, 1024):
        ids0 = res.ids.clone()
        for name in FLAVOURS:
            if name == primary:
                continue
            set_flavour(name)
            r2, resc2, el2 = measure()
            extras[name] = {"value": round(args.steps * args.queries * n_replicas / el2, 1),
                            "unit": "queries/s",
                            "ms_per_step": round(1e3 * el2 / args.steps, 3),
                            "rescued_queries": resc2, "roofline": probe_scan(name),
                            "fused_top10_identical_to_primary": bool(torch.equal(r2.ids, ids0))}
        set_flavour(primary)

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
            "dtype": ("f32" if primary == "f32" else "f16") + " MFMA shortlist scan (f32 accumulate) + "
                     "f64 rescoring of the f32 rows",
            "data": "synthetic",
            "config": {"workload": f"{args.docs}-doc / {args.dim}-d dense-only brute-force cosine "
                                   f"top-{args.top_k} (BASELINE.json configs[1])",
                       "docs": args.docs, "dim": args.dim, "queries_per_step": args.queries,
                       "semantic_top_k": 100, "fused_top_k": args.top_k, "shortlist": primary,
                       "parallelism": (f"doc-shard x{doc_shards}" + (f" x {n_replicas} replicas"
                                       if n_replicas > 1 else "")) if world > 1 else "single GPU",
                       "rescued_queries": rescued, "input_gen_s": round(gen_s, 1)},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        if extras:
            out["other_shortlists"] = extras
        if check:
            out["parity_check"] = check
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
