#!/usr/bin/env python3
"""Where an iteration of the resident PCG spends its time: cycle stamps of
workgroup 0 (SMVS_CG_TRACE) on the first Newton step of the bench workload."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/cg_trace.txt"
os.environ["SMVS_CG_TRACE"] = path
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
    print("solve", lines[0])
    rows = np.array([[int(x) for x in l.split()] for l in lines[1:]], dtype=np.int64)
    t0 = rows[0, 0]
    print("  kernel start -> first iteration: %.2f us" % ((rows[1, 0] - t0) / 100.0 if rows[1, 0] else -1))
    for k in range(1, min(len(rows), 9)):
        r = rows[k]
        if r[6] == 0:
            break
        d = np.diff(r[:7])
        print("  it %2d: " % k + "  ".join("%s %.2f" % (n, v / 100.0) for n, v in zip(names[1:], d))
              + "  | total %.2f us" % ((r[6] - r[0]) / 100.0))
