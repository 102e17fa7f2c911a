#!/usr/bin/env python3
"""What the solver costs ABOVE the resident limit (131,072 nodes): the assembled
system of one Newton step of a grid the resident kernel cannot hold, solved by
the streaming kernels (csrc/cg.hip: one SpMV + one update launch per iteration)
with the iteration limit at 10, 20, 40 and 80 (tolerances off); kernel time from
HIP events.  Slope = one iteration, with the algorithmic bytes of SURVEY 8(d)
(H upper half + vectors per node and iteration) beside it.
Usage: cg_streaming_cost.py <width> <height> <scale> [streaming|auto]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench, smvs_amd
from smvs_amd import synth

w, h, scale = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
mode = sys.argv[4] if len(sys.argv) > 4 else "auto"
prob = synth.make_problem(w, h, bench.NSUBS, scale, noise=bench.NOISE, seed=2000)
surf = prob["surf"]
ctx = smvs_amd.ViewContext(w, h, bench.NSUBS)
ctx.set_solver(mode)
ctx.set_views(prob["views"]); ctx.set_surface(surf)
n_patches = ctx.gn_construct(bench.REG)
ctx.profile(True)
rows = []
classes = ("cg_resident", "cg_spmv", "cg_update", "cg_init")
for limit in (10, 20, 40, 80, 10, 20, 40, 80):
    times = []
    for rep in range(4):
        ctx.profile_reset()
        its, info = ctx.cg_solve(max_iterations=limit, error_tolerance=0.0, q_tolerance=0.0)
        p = ctx.profile_get()
        times.append((sum(p[k][0] for k in classes) * 1000.0, {k: p[k] for k in classes}))
    best = min(times, key=lambda t: t[0])
    rows.append((its, best[0]))
    print("limit %3d: %3d iterations, solver kernels min %.1f us  %s" % (
        limit, its, best[0], {k: (round(v[0] * 1000, 1), int(v[1])) for k, v in best[1].items() if v[1]}))
x = np.array([r[0] for r in rows], float); y = np.array([r[1] for r in rows])
slope, icpt = np.polyfit(x, y, 1)
N = ctx.num_nodes
by = (bench.CG_BYTES["cg_spmv"] + bench.CG_BYTES["cg_update"]) * N
resident = p["cg_resident"][1] > 0
print("%dx%d scale %d: nodes %d patches %d solver %s: %.2f us per iteration, %.1f us before the first one"
      % (w, h, scale, N, n_patches, "RESIDENT" if resident else "streaming", slope, icpt))
if not resident:
    print("  algorithmic bytes per iteration %.1f MB (%d B per node: upper half of H, P, vectors) -> %.0f GB/s = %.3f of 8 TB/s"
          % (by / 1e6, by // N, by / (slope * 1e-6) / 1e9, by / (slope * 1e-6) / 8e12))
ctx.close()
