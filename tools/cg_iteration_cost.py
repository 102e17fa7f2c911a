#!/usr/bin/env python3
"""Cost of one iteration of the resident PCG WITHOUT the stamps of cg_trace.py: the assembled
system of the bench problem's first Newton step solved with the iteration limit at 10, 20, 40 and
80 (tolerances off), kernel time from HIP events (smvs_profile_*): the slope is the iteration,
the intercept the prologue.  Usage: cg_iteration_cost.py [scale]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench, smvs_amd
scale = int(sys.argv[1]) if len(sys.argv) > 1 else bench.SCALE
if scale == bench.SCALE:
    prob = bench.make_problem(0, False)
else:
    from smvs_amd import synth
    prob = synth.make_problem(bench.W, bench.H, bench.NSUBS, scale, noise=bench.NOISE, seed=2000)
surf = prob["surf"]
ctx = smvs_amd.ViewContext(surf["width"], surf["height"], bench.NSUBS)
ctx.set_views(prob["views"]); ctx.set_surface(surf)
ctx.gn_construct(bench.REG)
ctx.profile(True)
rows = []
for limit in (10, 20, 40, 80, 10, 20, 40, 80):
    times = []
    for rep in range(6):
        ctx.profile_reset()
        its, info = ctx.cg_solve(max_iterations=limit, error_tolerance=0.0, q_tolerance=0.0)
        ms, n = ctx.profile_get()["cg_resident"]
        if n != 1:
            sys.exit("the resident solver did not run (%d launches)" % n)
        times.append(ms * 1000.0)
    rows.append((its, min(times), float(np.median(times))))
    print("limit %3d: %3d iterations, kernel min %.1f us, median %.1f us" % (limit, its, rows[-1][1], rows[-1][2]))
x = np.array([r[0] for r in rows], float); y = np.array([r[1] for r in rows])
slope, icpt = np.polyfit(x, y, 1)
print("nodes %d: %.2f us per iteration, %.1f us before the first one (least squares over the minima)"
      % (ctx.num_nodes, slope, icpt))
ctx.close()
