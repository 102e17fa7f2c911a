#!/usr/bin/env python3
"""Wall-clock breakdown of the C++ host DepthOptimizer::optimize on a
synthetic scene (SMVS_HOST_TIMING prints the per-phase split on stderr)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SMVS_HOST_TIMING"] = "1"
import numpy as np
from smvs_amd import synth, host

kind = sys.argv[1] if len(sys.argv) > 1 else "sphere"
w = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
h = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
n = int(sys.argv[4]) if len(sys.argv) > 4 else 8
t = time.perf_counter()
inp = synth.pipeline_inputs(kind, w, h, n, flen=1.2 if kind == "sphere" else 1.0)
print("inputs rendered in %.1f s" % (time.perf_counter() - t))
for rep in range(2):
    t = time.perf_counter()
    out = host.optimize(inp, min_scale=2)
    dt = time.perf_counter() - t
    d = out["depth"]; m = d > 0
    err = np.sqrt(np.mean((d[m] - inp["truth"][m]) ** 2)) if m.any() else float("nan")
    print("optimize: %.3f s, %d batches, valid px %.1f%%, depth rms %.4g" % (dt, len(out["log"]), 100.0 * m.mean(), err))
print(out["log"])
