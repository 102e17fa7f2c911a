#!/usr/bin/env python3
"""Summary of the rocprofv3 --pmc passes of tools/kernel_pmc.sh: per kernel-name
substring the mean of every counter over the matching launches (and per wave),
the wave-lifetime split, instruction mix and HBM traffic.
Usage: pmc_summary.py <dir with s*/ passes> <substring[,substring...]> [command text]"""
import collections
import csv
import glob
import sys

out_dir, kernels = sys.argv[1], sys.argv[2].split(",")
cmd = sys.argv[3] if len(sys.argv) > 3 else ""
rows = []
for fn in glob.glob(out_dir + "/s*/**/*counter_collection.csv", recursive=True):
    with open(fn) as f:
        rows += list(csv.DictReader(f))
print("# rocprofv3 --pmc, one counter set per pass, --kernel-trace only (tools/kernel_pmc.sh) over: " + cmd)
for kern in kernels:
    acc = collections.defaultdict(list)
    names = set()
    for r in rows:
        if kern in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            names.add(r["Kernel_Name"][:90])
    if not acc:
        print("\n## no launch matches '%s'" % kern)
        continue
    m = {k: sum(v) / len(v) for k, v in acc.items()}
    n_launch = max(len(v) for v in acc.values())
    waves = m.get("SQ_WAVES", 1.0) or 1.0
    print("\n## kernels matching '%s' (%s): mean over %d launches of a pass, and per wave"
          % (kern, "; ".join(sorted(names)), n_launch))
    print("%-36s %14s %12s" % ("counter", "per launch", "per wave"))
    for k in sorted(m):
        print("%-36s %14.5g %12.1f" % (k, m[k], m[k] / waves))
    g = lambda k: m.get(k, 0.0) / waves
    wc = g("SQ_WAVE_CYCLES")
    if wc > 0:
        print("wave lifetime split (quad-cycle counters): issuing %.1f %%, waiting on s_waitcnt / barrier %.1f %%, "
              "issue stalls %.1f %%" % (100 * g("SQ_ACTIVE_INST_ANY") / wc, 100 * g("SQ_WAIT_ANY") / wc,
                                        100 * g("SQ_WAIT_INST_ANY") / wc))
        iv, isa, il = g("SQ_INSTS_VALU"), g("SQ_INSTS_SALU"), g("SQ_INSTS_LDS")
        ivm = g("SQ_INSTS_VMEM_RD") + g("SQ_INSTS_VMEM_WR")
        print("instructions per wave: %.0f VALU (FP64: %.0f FMA + %.0f MUL + %.0f ADD + %.0f transcendental; "
              "FP32: %.0f FMA + %.0f MUL + %.0f ADD; MFMA %.0f), %.0f SALU, %.0f LDS (bank-conflict cycles %.0f), "
              "%.0f VMEM (%.0f rd + %.0f wr); wave lifetime %.0f cycles = %.1f cycles per instruction"
              % (iv, g("SQ_INSTS_VALU_FMA_F64"), g("SQ_INSTS_VALU_MUL_F64"), g("SQ_INSTS_VALU_ADD_F64"),
                 g("SQ_INSTS_VALU_TRANS_F64"), g("SQ_INSTS_VALU_FMA_F32"), g("SQ_INSTS_VALU_MUL_F32"),
                 g("SQ_INSTS_VALU_ADD_F32"), g("SQ_INSTS_MFMA"), isa, il, g("SQ_LDS_BANK_CONFLICT"), ivm,
                 g("SQ_INSTS_VMEM_RD"), g("SQ_INSTS_VMEM_WR"), 4 * wc, 4 * wc / max(iv + isa + il + ivm, 1)))
    if "FETCH_SIZE" in m or "WRITE_SIZE" in m:
        # MI355X_MICROARCH.md, HBM / rocprofv3: both count KiB, and on gfx950
        # FETCH_SIZE reports half of the bytes read
        rd = 2.0 * 1024.0 * m.get("FETCH_SIZE", 0.0)
        wr = 1024.0 * m.get("WRITE_SIZE", 0.0)
        print("HBM traffic per launch: read %.2f MB, written %.2f MB, total %.2f MB "
              "(bytes = 2 x FETCH_SIZE x 1024 + WRITE_SIZE x 1024, separate passes)"
              % (rd / 1e6, wr / 1e6, (rd + wr) / 1e6))
