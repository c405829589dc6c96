#!/usr/bin/env python3
"""Per-kernel statistics of a rocprofv3 rocpd database (``rocprofv3 --kernel-trace`` writes
``*_results.db`` by default): calls, total / average / min / max duration.  The same table the
``--stats`` CSV holds, for runs whose output format was left at the default.
    python3 scripts/rocpd_stats.py FILE.db [substring of the kernel name] [--seq]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    pat = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else ""
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else "kernel_name"
    rows = cur.execute(f"select {name}, start, end from kernels order by start").fetchall()
    if "--seq" in sys.argv:   # launch by launch
        for n, s, e in rows:
            if pat in n:
                print(f"{(e - s) / 1e3:10.1f} us  {n[:110]}")
        return
    agg = {}
    for n, s, e in rows:
        if pat in n:
            a = agg.setdefault(n, [0, 0, 1 << 62, 0])
            a[0] += 1
            a[1] += e - s
            a[2] = min(a[2], e - s)
            a[3] = max(a[3], e - s)
    print("calls,total_us,avg_us,min_us,max_us,kernel")
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{a[0]},{a[1] / 1e3:.1f},{a[1] / a[0] / 1e3:.2f},{a[2] / 1e3:.2f},{a[3] / 1e3:.2f},\"{n[:140]}\"")


if __name__ == "__main__":
    main()
