#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite result (--kernel-trace --stats) as a
per-kernel table: calls, total / average / min / max duration."""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = cur.execute(
        "select %s, count(*), sum(end - start), avg(end - start), min(end - start), "
        "max(end - start) from kernels group by %s order by 3 desc" % (name_col, name_col)).fetchall()
    total = sum(r[2] for r in rows) or 1
    print("%-90s %8s %14s %12s %12s %12s %7s" % ("Name", "Calls", "TotalNs", "AvgNs", "MinNs", "MaxNs", "Pct"))
    for name, calls, tot, avg, mn, mx in rows:
        print("%-90s %8d %14d %12.0f %12d %12d %6.2f%%" % (name[:90], calls, tot, avg, mn, mx, 100.0 * tot / total))


if __name__ == "__main__":
    main(sys.argv[1])
