#!/usr/bin/env python3
"""Where the float16-copy scan spends its time: per-phase s_memtime sums of every wave
(thr_dense_scan_stamps_f16), at the bench shape.  Only the 4-wave-block kernel dense_scan_f16q has a
stamped build: run with THR_DENSE_F16=q at dim <= 768 (the default staggered kernel answers
"unsupported shape").  THR_DENSE_F16=q python3 scripts/scan_stamps.py [queries] [docs] [dim]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import triple_hybrid_rag_amd as T
    from triple_hybrid_rag_amd import synth
    nq = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
    d = int(sys.argv[3]) if len(sys.argv) > 3 else 768
    x = torch.from_numpy(synth.dense_rows(0, n, d)).cuda()
    q = torch.from_numpy(synth.dense_queries(nq, d, n)).cuda()
    idx = T.GpuIndex().set_dense(x, shortlist="f16")
    idx.dense_search(q, 100, rescue=False)
    st = T._native.dense_scan_stamps_f16(idx.docs16, n, nq, idx._ws)
    torch.cuda.synchronize()
    st = st.cpu().numpy().astype(np.float64)
    tiles = st[:, 5]
    names = ["wait_own_dma", "barrier", "ring_fill", "k_loop", "emit"]
    # s_memtime ticks are shader cycles (MI355X_MICROARCH.md); sums are per HALF tile
    out = {"waves": int(st.shape[0]), "half_tiles_per_wave": float(tiles.mean()),
           "cycles_per_half_tile": {nm: round(float((st[:, j] / tiles).mean()), 2) for j, nm in enumerate(names)},
           "cycles_per_half_tile_max_wave": {nm: round(float((st[:, j] / tiles).max()), 2) for j, nm in enumerate(names)},
           "loop_cycles_per_half_tile": round(float((st[:, 6] / tiles).mean()), 2)}
    lo, hi = st[:st.shape[0] // 2], st[st.shape[0] // 2:]
    w = np.arange(st.shape[0]) % 4
    out["by_wave_in_block"] = {int(k): {nm: round(float((st[w == k, j] / tiles[w == k]).mean()), 2)
                                        for j, nm in enumerate(names)} for k in range(4)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
