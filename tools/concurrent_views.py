#!/usr/bin/env python3
"""Experiment: V reference views in flight on one GPU (one context, stream and
host thread each) running the bench's Newton loop concurrently."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench, smvs_amd

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
prob = bench.make_problem(0, False)
surf = prob["surf"]
for V in (1, 2, 3, 4):
    ctxs = []
    for v in range(V):
        c = smvs_amd.ViewContext(surf["width"], surf["height"], bench.NSUBS)
        c.set_views(prob["views"]); c.set_surface(surf); c.save_nodes()
        bench.run_steps(c, prob, 20); c.set_nodes(surf["nodes"])
        ctxs.append(c)
    res = [None] * V
    def work(i):
        res[i] = bench.run_steps(ctxs[i], prob, steps)
    for c in ctxs: c.synchronize()
    t = time.perf_counter()
    th = [threading.Thread(target=work, args=(i,)) for i in range(V)]
    [x.start() for x in th]; [x.join() for x in th]
    for c in ctxs: c.synchronize()
    dt = time.perf_counter() - t
    ps = sum(r[0] for r in res)
    print("views in flight %d: %.1f M active-patch-steps/s (%.3f ms per step per view)" % (V, ps / dt / 1e6, 1e3 * dt / steps))
    for c in ctxs: c.close()
