#!/usr/bin/env python3
"""Runs smvs_gn_construct N times on the bench workload (profiling helper)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench, smvs_amd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
small = "--small" in sys.argv
prob = bench.make_problem(0, small)
surf = prob["surf"]
ctx = smvs_amd.ViewContext(surf["width"], surf["height"], bench.NSUBS)
ctx.set_views(prob["views"]); ctx.set_surface(surf)
ctx.gn_construct(bench.REG)
ctx.profile(True); ctx.profile_reset()
t = time.perf_counter()
for _ in range(n):
    ctx.gn_construct(bench.REG)
dt = time.perf_counter() - t
p = ctx.profile_get()
print("construct wall %.1f us; patch %.1f us assemble %.1f us" % (1e6 * dt / n, 1e3 * p["patch"][0] / n, 1e3 * p["assemble"][0] / n))
