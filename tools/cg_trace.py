#!/usr/bin/env python3
"""Where an iteration of the resident PCG spends its time: cycle stamps of
workgroup 0 (SMVS_CG_TRACE) on the first Newton step of the bench workload."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/cg_trace.txt"
os.environ["SMVS_CG_TRACE"] = path
os.environ["SMVS_LOOP_TEST"] = "unpipelined"   # the host waits for each solve
if os.path.exists(path):
    os.remove(path)
import numpy as np
import bench, smvs_amd
prob = bench.make_problem(0, False)
surf = prob["surf"]
ctx = smvs_amd.ViewContext(surf["width"], surf["height"], bench.NSUBS)
ctx.set_views(prob["views"]); ctx.set_surface(surf)
ctx.run_loop(bench.REG, max_newton_steps=3, reset_active=True)
ctx.close()
names = ["start", "d+halo", "spmv", "lower", "allreduce A", "update", "allreduce B"]
for block in open(path).read().split("solve")[1:]:
    lines = block.strip().split("\n")
    blocks = np.array([[int(x) for x in l.split()[1:]] for l in lines if l.startswith("block")],
                      dtype=np.int64)
    lines = [l for l in lines if not l.startswith("block")]
    if len(blocks):
        t00 = blocks[:, 1].min()
        rel = (blocks[:, 1:] - t00) / 100.0
        print("  prologue over the %d workgroups (units of this tool, relative to the first start):" % len(blocks))
        for name, col in (("start", 0), ("own blocks done", 1), ("rim blocks done", 2), ("before all-reduce", 3)):
            v = rel[:, col]
            print("    %-18s min %8.2f  median %8.2f  max %8.2f  (workgroup %d)"
                  % (name, v.min(), np.median(v), v.max(), int(blocks[np.argmax(v), 0])))
    print("solve", lines[0])
    rows = np.array([[int(x) for x in l.split()] for l in lines[1:]], dtype=np.int64)
    t0 = rows[0, 0]
    print("  kernel start -> first iteration: %.2f us" % ((rows[1, 0] - t0) / 100.0 if rows[1, 0] else -1))
    p = rows[0]
    if p[2] and p[3] and p[4]:
        print("  prologue: own blocks %.2f  rim blocks %.2f  P/r/z %.2f  first all-reduce %.2f us"
              % ((p[2] - p[0]) / 100.0, (p[3] - p[2]) / 100.0, (p[4] - p[3]) / 100.0,
                 (p[1] - p[4]) / 100.0))
    for k in range(1, min(len(rows), 9)):
        r = rows[k]
        if r[6] == 0:
            break
        d = np.diff(r[:7])
        print("  it %2d: " % k + "  ".join("%s %.2f" % (n, v / 100.0) for n, v in zip(names[1:], d))
              + "  | total %.2f us" % ((r[6] - r[0]) / 100.0))
