#!/usr/bin/env python3
"""Kernel timeline of ONE DepthOptimizer::optimize (configs[1]: 1920x1080, 8
neighbours, --no-sgm unless --sgm) for profiles/: run under

    rocprofv3 --kernel-trace --output-format csv -d DIR -o run -- \
        python tools/optimize_timeline.py run [--sgm]

(two optimize() calls 0.7 s apart; the second is the warm one; the maps a caller
may fetch afterwards with get_depth() / get_normals() are not part of it), then

    python tools/optimize_timeline.py report DIR/**/run_kernel_trace.csv

prints, for the warm call, GPU time and launches per kernel and per phase
(between two Newton loops = topology / grid surgery; inside a loop = the four
kernels of a step), the idle time of the GPU and the longest gaps."""
import collections, csv, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(sgm):
    from smvs_amd import synth, host
    inp = synth.pipeline_inputs("sphere", 1920, 1080, 8, flen=1.2)
    sd = host.sgm_depth(inp, sgm_scale=1) if sgm else None
    for rep in range(2):
        time.sleep(0.7)
        t = time.perf_counter()
        out = host.optimize(inp, regularization=0.01, num_iterations=5, min_scale=2,
                            sgm_depth=sd, want_maps=False)
        wall = time.perf_counter() - t
        loops = sum(e["loop_seconds"] for e in out["log"])
        aps = sum(e["active_patch_steps"] for e in out["log"])
        print("optimize %d: wall %.1f ms, loops %.2f ms, %d active patch-steps -> %.1f M/s"
              % (rep, 1e3 * wall, 1e3 * loops, aps, aps / loops / 1e6))
        for e in out["log"]:
            print("   scale %d iter %d: %d steps, %d CG its, %d aps, valid %d, loop %.3f ms"
                  % (e["scale"], e["iter"], e["newton_steps"], e["cg_iterations"],
                     e["active_patch_steps"], e["valid_patches"], 1e3 * e["loop_seconds"]))


def report(path):
    rows = []
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0].split("::")[-1].replace("void ", "")
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
    rows.sort()
    # the warm call: everything after the last gap > 0.5 s
    cut = 0
    for i in range(1, len(rows)):
        if rows[i][0] - rows[i - 1][1] > 500_000_000:
            cut = i
    rows = rows[cut:]
    t0, t1 = rows[0][0], rows[-1][1]
    busy = sum(e - s for s, e, _ in rows)
    print("warm optimize(): %d launches, first to last %.2f ms, GPU busy %.2f ms (%.0f %%)"
          % (len(rows), (t1 - t0) / 1e6, busy / 1e6, 100.0 * busy / (t1 - t0)))
    per = collections.defaultdict(lambda: [0, 0])
    for s, e, n in rows:
        per[n][0] += e - s
        per[n][1] += 1
    print("%-44s %6s %10s %9s" % ("kernel", "calls", "total us", "avg us"))
    for n, (t, c) in sorted(per.items(), key=lambda kv: -kv[1][0]):
        print("%-44s %6d %10.1f %9.2f" % (n[:44], c, t / 1e3, t / c / 1e3))
    # phases: a Newton loop = maximal run of {patch, resident/cg, reactivate, finish_step}
    loop_k = ("gn_patch_kernel", "cg_resident_kernel", "reactivate_kernel", "finish_step_kernel",
              "gn_assemble_kernel", "cg_", "coarse_step_kernel", "live_patch_list_kernel")
    phases = []
    for s, e, n in rows:
        kind = "loop" if n.startswith(loop_k) else "between"
        if not phases or phases[-1][0] != kind:
            phases.append([kind, s, e, 0, 0])
        phases[-1][2] = e
        phases[-1][3] += e - s
        phases[-1][4] += 1
    print("\nphases (kind, start ms, span us, busy us, launches):")
    for kind, s, e, b, c in phases:
        print("  %-8s %8.3f %9.1f %9.1f %5d" % (kind, (s - t0) / 1e6, (e - s) / 1e3, b / 1e3, c))
    tot = collections.defaultdict(float)
    for kind, s, e, b, c in phases:
        tot[kind + " span"] += (e - s) / 1e3
        tot[kind + " busy"] += b / 1e3
    print({k: round(v, 1) for k, v in tot.items()})
    gaps = sorted(((rows[i][0] - rows[i - 1][1], rows[i - 1][2], rows[i][2])
                   for i in range(1, len(rows))), reverse=True)[:25]
    print("\nlongest gaps (us, after, before):")
    for g, a, b in gaps:
        print("  %9.1f  %s -> %s" % (g / 1e3, a[:36], b[:36]))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run("--sgm" in sys.argv)
    else:
        report(sys.argv[2])
