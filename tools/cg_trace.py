#!/usr/bin/env python3
"""Where the resident PCG spends a solve: 100 MHz wall-clock stamps (SMVS_CG_TRACE) of
workgroup 0 per iteration and of every workgroup in the prologue, on the three Newton
steps of a batch of the bench workload (SMVS_LOOP_TEST=unpipelined: the host waits
for each solve).  Usage: cg_trace.py [out.txt [scale]] -- scale 6 / 5 / 4: the same
view at a coarse scale (1 / 4 / 16 tiles of the resident solver)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
parse_only = len(sys.argv) > 2 and sys.argv[1] == "--parse"   # cg_trace.py --parse raw.txt
path = sys.argv[2] if parse_only else sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/cg_trace.txt"
os.environ["SMVS_CG_TRACE"] = path
os.environ["SMVS_LOOP_TEST"] = "unpipelined"   # the host waits for each solve
import numpy as np
if not parse_only:
    if os.path.exists(path):
        os.remove(path)
    import bench, smvs_amd
    scale = int(sys.argv[2]) if len(sys.argv) > 2 else bench.SCALE
    if scale == bench.SCALE:
        prob = bench.make_problem(0, False)
    else:
        from smvs_amd import synth
        prob = synth.make_problem(bench.W, bench.H, bench.NSUBS, scale, noise=bench.NOISE, seed=2000)
    surf = prob["surf"]
    ctx = smvs_amd.ViewContext(surf["width"], surf["height"], bench.NSUBS)
    ctx.set_views(prob["views"]); ctx.set_surface(surf)
    ctx.run_loop(bench.REG, max_newton_steps=3, reset_active=True)
    ctx.close()
names2 = ["start", "d+halo", "spmv", "lower", "allreduce A", "update", "allreduce B"]
names1 = ["start", "product", "P q + sums", "allreduce (+ halo q)", "update"]
for block in open(path).read().split("solve")[1:]:
    lines = block.strip().split("\n")
    blocks = np.array([[int(x) for x in l.split()[1:]] for l in lines if l.startswith("block")],
                      dtype=np.int64)
    skew = np.array([[int(x) for x in l.split()[1:]] for l in lines if l.startswith("skew")],
                    dtype=np.int64)
    waves = np.array([[int(x) for x in l.split()[1:]] for l in lines if l.startswith("wave")],
                     dtype=np.int64)
    lines = [l for l in lines if not l.startswith(("block", "skew", "wave"))]
    if len(waves) and waves[:, 1].min() > 0:
        t0 = waves[:, 1].min()
        print("  iteration 5, the waves of workgroup %s (us after the first one started the iteration):"
              % os.environ.get("SMVS_CG_TRACE_WG", "0"))
        # stamps 16 .. 21 of the kernel: start, product done, sums formed, rim's stores issued (before the
        # arithmetic of the sums), partial sums in LDS, (the wave that stores the sums) all eight waves' seen
        print("    wave   start  product  rim stores issued  sums formed  partial sums in LDS  all partial sums seen"
              "  totals in  update done")
        cell = lambda x: "%6.2f" % ((x - t0) / 100.0) if x > 0 else "     -"
        for w in waves:
            print("    %4d  %s  %s  %s            %s       %s               %s              %s     %s"
                  % (w[0], cell(w[1]), cell(w[2]), cell(w[4]), cell(w[3]), cell(w[5]), cell(w[6]),
                     cell(w[7]), cell(w[8])))
    if len(skew) and skew[:, 1].min() > 0:
        t0 = skew[:, 1].min()
        print("  iteration 5 over the %d workgroups, us after the first one started it:" % len(skew))
        for name, col in (("iteration start", 1), ("sums published", 2), ("totals received", 3),
                          ("halo collected", 4)):
            v = (skew[:, col] - t0) / 100.0
            print("    %-16s min %6.2f  median %6.2f  max %6.2f  (last: workgroup %d)"
                  % (name, v.min(), np.median(v), v.max(), int(skew[np.argmax(v), 0])))
        lead = (skew[:, 0] % 16 == 0)   # (device-scope exchange: the first workgroup of 16)
        for name, m in (("group leaders", lead), ("members", ~lead)):
            if m.any():
                print("    %-13s start median %6.2f, sums median %6.2f, totals median %6.2f"
                      % (name, np.median((skew[m, 1] - t0) / 100.0), np.median((skew[m, 2] - t0) / 100.0),
                         np.median((skew[m, 3] - t0) / 100.0)))
        own = (skew[:, 2] - skew[:, 1]) / 100.0
        print("    start -> sums published inside a workgroup: min %.2f median %.2f max %.2f us"
              % (own.min(), np.median(own), own.max()))
    if len(blocks):
        # (wall_clock64: one 100 MHz counter for the whole device)
        rel = (blocks[:, 1:] - blocks[:, 1].min()) / 100.0
        print("  prologue over the %d workgroups, us after the first workgroup started:" % len(blocks))
        for name, col in (("start", 0), ("own blocks done", 1), ("rim blocks done", 2), ("before all-reduce", 3)):
            v = rel[:, col]
            print("    %-18s min %7.2f  median %7.2f  max %7.2f  (workgroup %d)"
                  % (name, v.min(), np.median(v), v.max(), int(blocks[np.argmax(v), 0])))
        own = (blocks[:, 4] - blocks[:, 1]) / 100.0
        print("    a workgroup's own prologue (start -> before all-reduce): median %.2f, max %.2f us"
              % (np.median(own), own.max()))
    print("solve", lines[0])
    names = names1 if "exchanges=1" in lines[0] else names2
    rows = np.array([[int(x) for x in l.split()] for l in lines[1:]], dtype=np.int64)
    t0 = rows[0, 0]
    print("  kernel start -> first iteration: %.2f us" % ((rows[1, 0] - t0) / 100.0 if rows[1, 0] else -1))
    p = rows[0]
    if p[2] and p[3] and p[4]:
        print("  prologue: own blocks (one exchange: halo, diagonal, P, upper blocks) %.2f  "
              "rim blocks %.2f  P/r/z %.2f  first all-reduce (one exchange: barrier) %.2f us"
              % ((p[2] - p[0]) / 100.0, (p[3] - p[2]) / 100.0, (p[4] - p[3]) / 100.0,
                 (p[1] - p[4]) / 100.0))
    for k in range(1, min(len(rows), 9)):
        r = rows[k]
        if "exchanges=1" in lines[0]:
            if r[4] == 0:
                break
            d = np.diff(r[:5])
            extra = "  [after the sums: group sum %.2f, total %.2f, halo %.2f]" % (
                (r[5] - r[2]) / 100.0 if r[5] else -1, (r[6] - r[2]) / 100.0 if r[6] else -1,
                (r[7] - r[2]) / 100.0 if r[7] else -1)
            if len(r) > 10 and r[8]:
                extra += "  [sweep wave: starts %.2f, second hop from %.2f, load rounds %d + %d]" % (
                    (r[8] - r[2]) / 100.0, (r[9] - r[2]) / 100.0 if r[9] else -1,
                    r[10] // 1000, r[10] % 1000)
            print("  it %2d: " % k + "  ".join("%s %.2f" % (n, v / 100.0) for n, v in zip(names[1:], d))
                  + "  | total %.2f us" % ((r[4] - r[0]) / 100.0) + extra)
            continue
        if r[6] == 0:
            break
        d = np.diff(r[:7])
        print("  it %2d: " % k + "  ".join("%s %.2f" % (n, v / 100.0) for n, v in zip(names[1:], d))
              + "  | total %.2f us" % ((r[6] - r[0]) / 100.0))
