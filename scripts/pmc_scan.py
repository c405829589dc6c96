#!/usr/bin/env python3
"""PMC passes over the dense scan kernels (how profiles/r1_dense_scan_traffic.json is made).

  run FLAVOUR            launch the scan of one shortlist flavour a few times at the bench shape
                         (1M x 768, 1536 queries); meant to sit behind rocprofv3, one counter set
                         per pass (MI355X_MICROARCH.md, HBM / rocprofv3):
      rocprofv3 --pmc FETCH_SIZE --kernel-trace -d D -o fetch_F -- python3 scripts/pmc_scan.py run F
      rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace \
                -d D -o sq_F -- python3 scripts/pmc_scan.py run F
  parse DIR OUT.json     fold the passes under DIR (fetch_F_* and sq_F_*, rocpd .db or csv) into the
                         JSON that bench.py reads, plus OUT_counters.csv with every raw counter row
"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
N_DOCS, DIM, NQ = 1_000_000, 768, 1536
FLAVOURS = ("f16-inline", "f16", "f32")
KERNEL_TAG = {"f32": "dense_scan_mfma2", "f16": "dense_scan_f16p", "f16-inline": "dense_scan_f16<"}


def run(flavour, nq=NQ):
    import torch
    import triple_hybrid_rag_amd as T
    from triple_hybrid_rag_amd import synth
    x = torch.from_numpy(synth.dense_rows(0, N_DOCS, DIM)).cuda()
    q = torch.from_numpy(synth.dense_queries(nq, DIM, N_DOCS)).cuda()
    idx = T.GpuIndex().set_dense(x, shortlist=flavour)
    idx.dense_search(q, 100, rescue=False)       # sets tau in the workspace
    for _ in range(3):
        idx.scan_probe(q)
    torch.cuda.synchronize()


def rows(path):
    """(dispatch_id, kernel_name, counter_name, value, start, end) of one rocprofv3 pass
    (the rocpd database's counters_collection view, or a counter_collection.csv)."""
    if path.endswith(".db"):
        import sqlite3
        cur = sqlite3.connect(path).cursor()
        for r in cur.execute("select dispatch_id, kernel_name, counter_name, value, start, end "
                             "from counters_collection order by dispatch_id"):
            yield r
    else:
        with open(path, newline="") as f:
            for r in csv.DictReader(f):
                yield (int(r["Dispatch_Id"]), r["Kernel_Name"], r["Counter_Name"],
                       float(r["Counter_Value"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"]))


def filter_scan(path, flavour, dump=None):
    """The probe launches: the last three MODE_FILTER dispatches of the flavour's scan kernel
    (MODE is the second template argument, "<768, 1, ..." / "<96, 1, ...")."""
    out = {}
    for disp, name, cname, value, start, end in rows(path):
        if dump is not None:
            dump.append((os.path.basename(path), disp, name, cname, value, end - start))
        if KERNEL_TAG[flavour] not in name or "<" not in name:
            continue
        if name.split("<")[1].split(",")[1].strip() != "1":
            continue
        e = out.setdefault((disp, name), {})
        e[cname] = float(value)
        e["_ns"] = end - start
    return sorted(out.items())[-3:]


def parse(d, out_path):
    res = {"source": "rocprofv3 --pmc passes of scripts/pmc_scan.py (separate runs per counter set)",
           "n_docs": N_DOCS, "dim": DIM, "queries": NQ,
           "gfx950_correction": "FETCH_SIZE is in KiB counted at 64 B per 128-B request: bytes = "
                                "value * 1024 * 2 (MI355X_MICROARCH.md, HBM)",
           "flavours": {}}
    dump = []
    for fl in FLAVOURS:
        e = {}
        for path in glob.glob(os.path.join(d, "**", f"fetch_{fl}_*"), recursive=True):
            ls = filter_scan(path, fl, dump)
            if ls:
                kb = sum(v["FETCH_SIZE"] for _, v in ls) / len(ls)
                e["kernel"] = ls[-1][0][1].split("(")[0]
                e["FETCH_SIZE_KB_avg_per_launch"] = kb
                e["hbm_bytes_per_launch"] = round(kb * 1024 * 2)
        for path in glob.glob(os.path.join(d, "**", f"sq_{fl}_*"), recursive=True):
            ls = filter_scan(path, fl, dump)
            if ls:
                busy = sum(v["SQ_VALU_MFMA_BUSY_CYCLES"] for _, v in ls) / len(ls)
                sq = sum(v["SQ_BUSY_CYCLES"] for _, v in ls) / len(ls)
                gui = sum(v["GRBM_GUI_ACTIVE"] for _, v in ls) / len(ls)
                ns = sum(v["_ns"] for _, v in ls) / len(ls)
                e["SQ_VALU_MFMA_BUSY_CYCLES"] = busy
                e["SQ_BUSY_CYCLES"] = sq
                e["GRBM_GUI_ACTIVE"] = gui
                # GRBM_GUI_ACTIVE sums the 8 XCDs; MFMA busy is summed over the SIMDs of all CUs
                e["effective_clock_ghz"] = round(gui / 8 / ns, 3)
                e["mfma_busy"] = round(busy / (gui / 8 * 256 * 4) , 4) if gui else None
                e["launch_ms_under_pmc"] = round(ns / 1e6, 4)
        passes = -(-NQ // {"f32": 32, "f16-inline": 64, "f16": 96}[fl])   # bench shape: 1536 queries
        e["passes"] = passes
        e["algorithmic_bytes_per_launch_8d"] = passes * N_DOCS * DIM * (2 if fl == "f16" else 4)
        e["one_corpus_pass_bytes"] = N_DOCS * DIM * (2 if fl == "f16" else 4)
        if "hbm_bytes_per_launch" in e:
            e["traffic_over_one_corpus_pass"] = round(e["hbm_bytes_per_launch"] / e["one_corpus_pass_bytes"], 3)
        res["flavours"][fl] = e
    with open(out_path, "w") as f:
        json.dump(res, f, indent=1)
    with open(out_path.replace(".json", "_counters.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["pass", "dispatch_id", "kernel_name", "counter_name", "value", "duration_ns"])
        w.writerows(dump)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else NQ)
    else:
        parse(sys.argv[2], sys.argv[3])
