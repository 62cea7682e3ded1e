#!/usr/bin/env python3
"""Summarise rocprofv3 (rocpd sqlite) outputs: per-kernel stats and per-kernel mean PMC counter values."""
import sqlite3
import sys


def main(path):
    import glob, os
    if os.path.isdir(path):   # rocprofv3 -d <dir>: <dir>/<host>/<pid>_results.db
        dbs = sorted(glob.glob(os.path.join(path, "**", "*.db"), recursive=True), key=os.path.getsize)
        path = dbs[-1]
    db = sqlite3.connect(path)
    cur = db.cursor()
    rows = cur.execute("select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start) from kernels group by name order by 6 desc").fetchall()
    tot = sum(r[5] for r in rows) or 1
    print("%-60s %6s %12s %12s %12s %7s" % ("kernel", "calls", "avg_us", "min_us", "max_us", "pct"))
    for r in rows:
        print("%-60s %6d %12.2f %12.2f %12.2f %6.1f%%" % (r[0][:60], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, 100.0 * r[5] / tot))
    try:
        cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
        kn = "kernel_name" if "kernel_name" in cols else "name"
        rows = cur.execute(f"select {kn}, counter_name, count(*), avg(value) from counters_collection group by {kn}, counter_name order by {kn}").fetchall()
        if rows:
            print("\n%-50s %-28s %6s %18s" % ("kernel", "counter", "n", "mean_per_dispatch"))
            for r in rows:
                print("%-50s %-28s %6d %18.1f" % (r[0][:50], r[1], r[2], r[3]))
    except sqlite3.Error as e:
        print("no counters:", e, cols)


if __name__ == "__main__":
    main(sys.argv[1])
