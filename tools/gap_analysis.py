#!/usr/bin/env python3
"""Timeline gaps from a rocprofv3 kernel trace CSV: where the GPU idles
between kernels (host synchronisation bubbles vs launch gaps)."""
import csv, sys, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:]))
rows.sort()
# keep the steady part: last 60%
rows = rows[len(rows) * 4 // 10:]
busy = sum(e - s for s, e, _ in rows)
span = rows[-1][1] - rows[0][0]
print("kernels %d span %.1f ms busy %.1f ms (%.1f%%)" % (len(rows), span / 1e6, busy / 1e6, 100.0 * busy / span))
gaps = collections.defaultdict(list)
for (s0, e0, k0), (s1, e1, k1) in zip(rows, rows[1:]):
    gaps[(k0, k1)].append(s1 - e0)
tot = sorted(gaps.items(), key=lambda kv: -sum(kv[1]))
for (k0, k1), g in tot[:14]:
    print("%-42s -> %-42s n=%5d mean %7.2f us total %8.1f us" % (k0, k1, len(g), sum(g) / len(g) / 1e3, sum(g) / 1e3))
dur = collections.defaultdict(list)
for s, e, k in rows:
    dur[k].append(e - s)
for k, d in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print("%-42s n=%5d mean %8.2f us total %9.1f us" % (k, len(d), sum(d) / len(d) / 1e3, sum(d) / 1e3))
