#!/usr/bin/env python3
"""Wall clock of the host SGM initialisation (reconstruct_sgm_depth_for_view,
app/smvsrecon.cc:346-384) + SGM-initialised optimize() on a synthetic scene."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SMVS_HOST_TIMING"] = "1"
import numpy as np
from smvs_amd import synth, host

w = int(sys.argv[1]) if len(sys.argv) > 1 else 1920
h = int(sys.argv[2]) if len(sys.argv) > 2 else 1080
n = int(sys.argv[3]) if len(sys.argv) > 3 else 8
inp = synth.pipeline_inputs("sphere", w, h, n, flen=1.2)
for rep in range(2):
    t = time.perf_counter()
    sd = host.sgm_depth(inp, sgm_scale=1)
    t1 = time.perf_counter()
    out = host.optimize(inp, min_scale=2, sgm_depth=sd)
    t2 = time.perf_counter()
    m = sd > 0
    print("sgm_depth: %.3f s (valid %.1f%%), optimize with SGM init: %.3f s, %d batches"
          % (t1 - t, 100.0 * m.mean(), t2 - t1, len(out["log"])))
