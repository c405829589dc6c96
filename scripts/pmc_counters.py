#!/usr/bin/env python3
"""Fold rocprofv3 --pmc passes (csv output) into one JSON: per counter the average over the
dispatches of the kernels whose name contains TAG and whose template arguments contain MODE.

    python3 scripts/pmc_counters.py DIR TAG OUT.json [ARGS_SUBSTRING]

DIR holds *_counter_collection.csv files of separate passes (one counter set each, as the
MI355X guide prescribes).  Derived figures (gfx950): FETCH_SIZE is KiB at 64 B per 128-B request
(bytes = value * 2048); SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles;
SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs; GRBM_GUI_ACTIVE sums the 8 XCDs.
"""
import csv
import glob
import json
import os
import sys


def main():
    d, tag, out = sys.argv[1:4]
    sub = sys.argv[4] if len(sys.argv) > 4 else ""
    acc = {}
    names = set()
    for path in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        per = {}
        with open(path, newline="") as f:
            for r in csv.DictReader(f):
                kn = r["Kernel_Name"]
                if tag not in kn or sub not in kn:
                    continue
                names.add(kn.split("(")[0])
                e = per.setdefault(int(r["Dispatch_Id"]), {})
                e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                e["_ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
                e["_vgpr"] = int(r["VGPR_Count"]) + int(r.get("Accum_VGPR_Count", 0) or 0)
                e["_lds"] = int(r["LDS_Block_Size"])
                e["_grid"] = int(r["Grid_Size"])
                e["_wg"] = int(r["Workgroup_Size"])
        disp = [per[k] for k in sorted(per)][-3:]   # the probe launches are the last three
        for e in disp:
            for k, v in e.items():
                acc.setdefault(k + "@" + os.path.basename(path), []).append(v)
    res = {"kernels": sorted(names), "counters": {}}
    for k, v in sorted(acc.items()):
        name, src = k.split("@")
        res["counters"].setdefault(name, {})[src] = sum(v) / len(v)
    c = {k: list(v.values())[0] for k, v in res["counters"].items()}
    der = {}
    if "GRBM_GUI_ACTIVE" in c and "_ns" in c:
        ns = res["counters"]["_ns"]
        gui_src = list(res["counters"]["GRBM_GUI_ACTIVE"].keys())[0]
        der["clock_ghz"] = c["GRBM_GUI_ACTIVE"] / 8 / ns[gui_src]
        der["kernel_cycles"] = c["GRBM_GUI_ACTIVE"] / 8
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            der["mfma_pipe_busy"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (der["kernel_cycles"] * 1024)
    if "SQ_WAVE_CYCLES" in c:
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS",
                  "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM",
                  "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC"):
            if k in c:
                der[k + "/WAVE_CYCLES"] = c[k] / c["SQ_WAVE_CYCLES"]
    if "FETCH_SIZE" in c:
        der["hbm_read_bytes_per_launch"] = c["FETCH_SIZE"] * 2048
    if "SQ_LDS_BANK_CONFLICT" in c and "SQ_LDS_IDX_ACTIVE" in c:
        der["lds_bank_conflict_share"] = c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_LDS_IDX_ACTIVE"], 1)
    if "TCC_HIT_sum" in c:
        der["l2_hit_rate"] = c["TCC_HIT_sum"] / max(c["TCC_HIT_sum"] + c["TCC_MISS_sum"], 1)
    res["derived"] = {k: round(v, 4) for k, v in der.items()}
    with open(out, "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
    print(json.dumps(res["derived"], indent=1))
    print({k: round(v) for k, v in c.items()})


if __name__ == "__main__":
    main()
