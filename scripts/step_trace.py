#!/usr/bin/env python3
"""One timed step of a rocprofv3 --kernel-trace CSV of bench.py, kernel by kernel: start / end
relative to the step's first kernel (thr_embed_postproc), duration, HIP stream.
    python3 scripts/step_trace.py gpurun_out/x/t_kernel_trace.csv [step index, default the middle one]"""
import csv
import re
import sys


def load(path):
    out = []
    for r in csv.DictReader(open(path)):
        n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("thr::", "")
        out.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, r.get("Queue_Id"), r.get("Stream_Id")))
    out.sort()
    return out


def main():
    rows = load(sys.argv[1])
    idx = [i for i, r in enumerate(rows) if "embed_postproc" in r[2]]
    j = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) // 2
    i0, i1 = idx[j], idx[j + 1]
    t0 = rows[i0][0]
    print(f"step {j} of {len(idx)}: {(rows[i1][0] - t0) / 1e3:.1f} us from its first kernel to the next step's")
    print("| start us | end us | duration us | stream | kernel |\n|---|---|---|---|---|")
    for r in rows[i0:i1]:
        print(f"| {(r[0] - t0) / 1e3:.1f} | {(r[1] - t0) / 1e3:.1f} | {(r[1] - r[0]) / 1e3:.1f} | {r[4]} | {r[2][:80]} |")


if __name__ == "__main__":
    main()
