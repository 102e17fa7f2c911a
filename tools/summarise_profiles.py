#!/usr/bin/env python3
"""Condense the rocprofv3 passes of tools/collect_profiles.sh into the files
kept under profiles/: per-kernel statistics, per-kernel HBM traffic and the
traffic JSON bench.py reads.

HBM bytes follow MI355X_MICROARCH.md's rocprofv3 section: FETCH_SIZE and
WRITE_SIZE are collected in separate passes, both count KiB, and on gfx950
FETCH_SIZE reports half of the bytes read:
    bytes = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024.
"""
import collections
import csv
import glob
import json
import os
import sys

SHORT = [("gn_patch_kernel", "patch"), ("gn_assemble_kernel", "assemble"),
         ("cg_resident_kernel", "cg_resident"), ("live_patch_list_kernel", "live_patch_list"),
         ("cg_spmv_kernel", "cg_spmv"), ("cg_update_kernel", "cg_update"),
         ("cg_init_kernel", "cg_init"), ("reactivate_kernel", "reactivate"),
         ("apply_update_kernel", "apply_update"), ("prepare_update_kernel", "prepare_update"),
         ("finish_step_kernel", "finish_step")]


def short(name):
    for key, s in SHORT:
        if key in name:
            return s
    return name.split("(")[0].split("::")[-1][:40]


def counter(dirname, cname):
    per = collections.defaultdict(list)
    for fn in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(fn)):
            if r["Counter_Name"] == cname:
                per[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return per


def main(src, dst, tag):
    os.makedirs(dst, exist_ok=True)
    # ---- kernel statistics (pass 1: --kernel-trace --stats) ----
    stats = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)
    lines = []
    if stats:
        rows = list(csv.DictReader(open(stats[0])))
        with open(os.path.join(dst, "%s_kernel_stats.csv" % tag), "w") as f:
            w = csv.writer(f)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
            for r in rows:
                w.writerow([r["Name"][:100], r["Calls"], r["TotalDurationNs"], r["AverageNs"],
                            r["Percentage"], r["MinNs"], r["MaxNs"]])
    # ---- HBM traffic (passes 2, 3) ----
    fetch = counter(os.path.join(src, "fetch"), "FETCH_SIZE")
    write = counter(os.path.join(src, "write"), "WRITE_SIZE")
    traffic = {}
    lines.append("%-16s %8s %14s %14s %14s %14s" % ("kernel", "launches", "read MB (avg)",
                                                    "write MB (avg)", "total MB (avg)", "total MB (max)"))
    for k in sorted(set(fetch) | set(write)):
        fr = [2.0 * 1024.0 * v for v in fetch.get(k, [])]
        wr = [1024.0 * v for v in write.get(k, [])]
        # launches after CG convergence are no-ops: average over launches that moved data
        work_f = [v for v in fr if v > 1e6] or fr or [0.0]
        work_w = [v for v in wr if v > 2.5e5] or wr or [0.0]
        avg = sum(work_f) / len(work_f) + sum(work_w) / len(work_w)
        mx = (max(fr) if fr else 0.0) + (max(wr) if wr else 0.0)
        traffic[k] = int(avg)
        lines.append("%-16s %8d %14.1f %14.1f %14.1f %14.1f" % (
            k, len(work_f), sum(work_f) / len(work_f) / 1e6, sum(work_w) / len(work_w) / 1e6,
            avg / 1e6, mx / 1e6))
    with open(os.path.join(dst, "%s_hbm_traffic.txt" % tag), "w") as f:
        f.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of\n"
                "# `python bench.py --steps 20 --warmup 3 --repeats 3 --no-cpu-baseline --no-peaks`; bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024\n"
                "# averages are over launches that moved data (CG launches after convergence are no-ops)\n")
        f.write("\n".join(lines) + "\n")
    with open(os.path.join(dst, "traffic_%s.json" % tag), "w") as f:
        json.dump(traffic, f, indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "r1")
