#!/usr/bin/env python3
"""Per-Newton-step cost inside one batch of the bench workload: active
patches, PCG iterations and wall time of every step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench, smvs_amd

prob = bench.make_problem(0, False)
surf = prob["surf"]
ctx = smvs_amd.ViewContext(surf["width"], surf["height"], bench.NSUBS)
ctx.set_views(prob["views"]); ctx.set_surface(surf); ctx.save_nodes()
bench.run_steps(ctx, prob, 20)
for rep in range(2):
    ctx.set_nodes(surf["nodes"]); ctx.set_active(None); ctx.synchronize()
    n_init = int(surf["node_valid"].sum()); n_act = n_init; step = 0
    tot = 0.0
    while n_act > n_init // 20 and step < 50:
        t = time.perf_counter()
        st = ctx.run_loop(bench.REG, max_newton_steps=1, reset_active=False)
        dt = time.perf_counter() - t
        step += 1; tot += dt
        n_act = st["final_active_nodes"]
        if rep == 1:
            print("step %2d: active patches %6d  cg its %3d  %.3f ms  -> active nodes %6d" % (
                step, st["active_patch_steps"], st["linear_iterations"], 1e3 * dt, n_act))
    print("batch: %d steps, %.3f ms" % (step, 1e3 * tot))
