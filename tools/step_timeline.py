#!/usr/bin/env python3
"""Per-kernel durations and the gaps between consecutive kernels inside the
Newton steps of the bench's timed region, from a rocprofv3 kernel trace:
    rocprofv3 --kernel-trace --output-format csv -d DIR -o run -- python bench.py ...
    python tools/step_timeline.py DIR/**/run_kernel_trace.csv"""
import collections, csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].split("(")[0].split("::")[-1]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name.replace("void ", "")))
rows.sort()
# steps of the fused loop: patch -> resident -> reactivate -> finish_step
steps = []
i = 0
while i + 3 < len(rows):
    names = [rows[i + k][2] for k in range(4)]
    if (names[0].startswith("gn_patch_kernel") and names[1].startswith("cg_resident_kernel")
            and names[2] == "reactivate_kernel" and names[3] == "finish_step_kernel"):
        # (a step enqueued behind the end of its loop does nothing: leave it out)
        if rows[i + 1][1] - rows[i + 1][0] > 20000:
            steps.append(rows[i:i + 4] + [i])
        i += 4
    else:
        i += 1
print("fused steps found:", len(steps))
dur = collections.defaultdict(list)
gap = collections.defaultdict(list)
index = [st[4] for st in steps]
steps = [st[:4] for st in steps]
loops = 1
for k, st in enumerate(steps):
    for s, e, n in st:
        dur[n].append(e - s)
    for a, b in zip(st, st[1:]):
        gap[a[2] + " -> " + b[2]].append(b[0] - a[1])
    if k + 1 < len(steps):
        if index[k + 1] == index[k] + 4:
            # the next step of the same loop (nothing launched in between)
            gap["finish_step_kernel -> next step of the loop"].append(steps[k + 1][0][0] - st[-1][1])
        else:
            loops += 1
            if steps[k + 1][0][0] - st[-1][1] < 500000:
                gap["end of a loop -> first patch kernel of the next (restore, loop begin, skipped step)"].append(
                    steps[k + 1][0][0] - st[-1][1])
tot = 0.0
print("loops: %d (%.2f steps per loop)" % (loops, len(steps) / max(loops, 1)))
for n, d in dur.items():
    print("%-28s n=%4d mean %8.2f us" % (n, len(d), sum(d) / len(d) / 1e3)); tot += sum(d) / len(d)
for n, g in gap.items():
    print("%-60s n=%4d mean %7.2f us" % (n, len(g), sum(g) / len(g) / 1e3))
    if not n.startswith("end of a loop"):
        tot += sum(g) / len(g)
print("sum of the in-loop means: %.1f us per step" % (tot / 1e3))
