#!/usr/bin/env python3
"""Per-kernel summary (calls, total, average, min, max in microseconds) of this
library's kernels from a rocprofv3 --kernel-trace result database (rocpd sqlite)."""
import sqlite3
import sys


def main(path, prefix="ddc"):
    db = sqlite3.connect(path)
    cur = db.cursor()
    rows = cur.execute(
        "select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 "
        "from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    print("%-34s %7s %12s %12s %12s %12s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"))
    other = 0.0
    for name, n, t, a, lo, hi in rows:
        short = name.split("(")[0]
        if prefix in short:
            print("%-34s %7d %12.1f %12.2f %12.2f %12.2f %6.2f" % (short[-34:], n, t, a, lo, hi, 100 * t / tot))
        else:
            other += t
    print("%-34s %7s %12.1f %12s %12s %12s %6.2f" % ("(all non-library kernels)", "", other, "", "", "", 100 * other / tot))


if __name__ == "__main__":
    main(*sys.argv[1:])
