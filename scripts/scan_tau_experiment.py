#!/usr/bin/env python3
"""How much of the f16 scan is the emit's rare path: time the scan alone with the thresholds of a
real search and with tau = +inf (nothing passes).  python3 scripts/scan_tau_experiment.py [queries]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import triple_hybrid_rag_amd as T
    from triple_hybrid_rag_amd import synth
    nq = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    n, d = 1_000_000, 768
    x = torch.from_numpy(synth.dense_rows(0, n, d)).cuda()
    q = torch.from_numpy(synth.dense_queries(nq, d, n)).cuda()
    idx = T.GpuIndex().set_dense(x, shortlist="f16")
    idx.dense_search(q, 100, rescue=False)

    def timed(reps=8):
        idx.scan_probe(q)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            idx.scan_probe(q)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    out = {"queries": nq, "scan_ms_real_tau": round(timed(), 4)}
    # tau is the first array of the workspace (make_plan: off_tau = 0), qpad floats
    qpad = (nq + 127) // 128 * 128
    tau = idx._ws[: 4 * qpad].view(torch.float32)
    keep = tau.clone()
    tau.fill_(float("inf"))
    out["scan_ms_tau_inf"] = round(timed(), 4)
    tau.copy_(keep - 0.02)        # ~4x more rows pass
    out["scan_ms_tau_lower"] = round(timed(), 4)
    tau.copy_(keep)
    out["scan_ms_real_tau_again"] = round(timed(), 4)
    flops = 2.0 * n * d * nq
    out["tflops"] = {k: round(flops / (v * 1e-3) / 1e12, 1) for k, v in out.items() if k.startswith("scan_ms")}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
