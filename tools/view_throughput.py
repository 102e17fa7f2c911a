#!/usr/bin/env python3
"""Whole-view throughput on one GPU through the C++ ViewQueue
(smvs_host_optimize_views): per view StereoView::create x 9, optional SGM front
end, DepthOptimizer::optimize (all scales), depth + normal maps -- the per-view
task of smvsrecon (app/smvsrecon.cc:658-733) on a synthetic 1920x1080 /
8-neighbour scene, for 1, 2, 3, 4 views in flight.

    python tools/view_throughput.py [W H NEIGHBOURS] [--sgm] [--shading]
"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from smvs_amd import synth, host

args = [a for a in sys.argv[1:] if not a.startswith("--")]
w = int(args[0]) if len(args) > 0 else 1920
h = int(args[1]) if len(args) > 1 else 1080
n = int(args[2]) if len(args) > 2 else 8
use_sgm = "--sgm" in sys.argv
shading = "--shading" in sys.argv
lighting = None
if shading:
    lighting = np.zeros(16); lighting[0] = 0.85; lighting[1:4] = (0.1, -0.15, 0.05)
t = time.perf_counter()
inp = synth.pipeline_inputs("sphere", w, h, n, flen=1.2, lighting=lighting)
print("inputs rendered in %.1f s" % (time.perf_counter() - t), flush=True)
kw = dict(min_scale=2, use_shading=shading, sgm_scale=1 if use_sgm else None)
host.optimize_views(inp, 2, views_in_flight=1, **kw)   # warm-up: library, pools
out = {}
for in_flight in (1, 2, 4, 6, 8):
    jobs = 3 * in_flight if in_flight > 4 else 4 * in_flight
    r = host.optimize_views(inp, jobs, views_in_flight=in_flight, **kw)
    out[in_flight] = dict(views=jobs, seconds=round(r["total_seconds"], 3),
                          views_per_s=round(r["views_per_s"], 2),
                          ms_per_view_latency=round(1e3 * float(np.mean(r["job_seconds"])), 1))
    print("in flight %d: %d views in %.2f s = %.2f views/s (mean task %.0f ms)"
          % (in_flight, jobs, r["total_seconds"], r["views_per_s"],
             1e3 * np.mean(r["job_seconds"])), flush=True)
print(json.dumps(dict(size=[w, h], neighbours=n, sgm=use_sgm, shading=shading, result=out)))
