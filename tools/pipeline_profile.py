#!/usr/bin/env python3
"""One reference view through the whole device pipeline at BASELINE.json's
size (1920x1080, 8 neighbours): SGM front end (4 x run_sgm, L/R check,
merge), bilateral upsample, scale space, topology kernels and Newton loops of
optimize() with SGM initialisation, then the cross-view depth-map cut over 9
views.  Run under rocprofv3 --kernel-trace --stats for profiles/."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from smvs_amd import synth, host
import smvs_amd

inp = synth.pipeline_inputs("sphere", 1920, 1080, 8, flen=1.2)
for rep in range(2):
    t = time.perf_counter()
    sd = host.sgm_depth(inp, sgm_scale=1)
    t1 = time.perf_counter()
    out = host.optimize(inp, min_scale=2, sgm_depth=sd)
    t2 = time.perf_counter()
    print("sgm front end %.3f s, optimize %.3f s (%d batches)" % (t1 - t, t2 - t1, len(out["log"])))
depths, normals = synth.depth_and_normal_maps(inp["scene"], inp["cams"])
t = time.perf_counter()
cut, _ = smvs_amd.cut_depth_maps(inp["cams"], depths, normals)
print("cut_depth_maps over %d views: %.3f s, kept %.1f%%" % (
    len(cut), time.perf_counter() - t,
    100.0 * sum((c > 0).sum() for c in cut) / sum((d > 0).sum() for d in depths)))
