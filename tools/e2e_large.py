#!/usr/bin/env python3
"""End-to-end validation at a chosen size: C++ host optimize (HIP) vs the
oracle's optimize() on the same inputs.
    e2e_large.py [sphere|plane] W H N [--sgm] [--shading]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from smvs_amd import synth, host
from oracle import pyoracle as oracle

args = [a for a in sys.argv[1:] if not a.startswith("--")]
kind = args[0] if len(args) > 0 else "sphere"
w = int(args[1]) if len(args) > 1 else 960
h = int(args[2]) if len(args) > 2 else 540
n = int(args[3]) if len(args) > 3 else 4
use_sgm = "--sgm" in sys.argv
shading = "--shading" in sys.argv
oracle.lib().orc_set_threads(max(1, min(os.cpu_count() or 1, 64)))
lighting = None
if shading:
    rng = np.random.default_rng(3000)
    lighting = np.zeros(16); lighting[0] = 0.9
    lighting[1:4] = rng.uniform(-0.2, 0.2, 3)
inp = synth.pipeline_inputs(kind, w, h, n, flen=1.2 if kind == "sphere" else 1.0,
                            lighting=lighting)
sgm = sgm_o = None
if use_sgm:
    t = time.perf_counter(); sgm = host.sgm_depth(inp, sgm_scale=1); t1 = time.perf_counter()
    sgm_o = oracle.sgm_depth_for_view(inp, sgm_scale=1, roundtrip=True); t2 = time.perf_counter()
    print("SGM front end: HIP %.3f s, oracle %.1f s" % (t1 - t, t2 - t1))
t = time.perf_counter()
got = host.optimize(inp, min_scale=2, use_shading=shading, sgm_depth=sgm)
t1 = time.perf_counter()
want = oracle.optimize(inp, regularization=0.01, num_iterations=5, min_scale=2,
                       use_shading=shading, sgm_depth=sgm_o)
t2 = time.perf_counter()
if use_sgm:
    print("sgm roundtrip identical:", np.array_equal(got["sgm_roundtrip"], sgm_o))
key = lambda e: (e["scale"], e["iter"], e["newton_steps"], e["valid_patches"])
same = [key(e) for e in got["log"]] == [key(e) for e in want["log"]]
rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
both = (got["depth"] > 0) & (want["depth"] > 0)
print("%s %dx%d, %d neighbours%s%s: HIP %.2f s, oracle %.1f s; batches %d / %d, identical control "
      "flow %s, valid pixels differ at %d of %d, depth rel L2 (common pixels) %.3e, normals %.3e"
      % (kind, w, h, n, " +sgm" if use_sgm else "", " +shading" if shading else "",
         t1 - t, t2 - t1, len(got["log"]), len(want["log"]), same,
         int(((got["depth"] > 0) != (want["depth"] > 0)).sum()), got["depth"].size,
         rel(got["depth"][both], want["depth"][both]),
         rel(got["normals"][both], want["normals"][both])))
for a, b in zip(got["log"], want["log"]):
    print(key(a), a["cg_iterations"], "|", key(b), b["cg_iterations"],
          "" if (key(a), a["cg_iterations"]) == (key(b), b["cg_iterations"]) else "  <--")
