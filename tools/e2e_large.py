#!/usr/bin/env python3
"""One-off end-to-end validation at a larger size than the test suite uses:
C++ host optimize (HIP) vs the oracle's optimize() on the same inputs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from smvs_amd import synth, host
from oracle import pyoracle as oracle

kind = sys.argv[1] if len(sys.argv) > 1 else "sphere"
w = int(sys.argv[2]) if len(sys.argv) > 2 else 960
h = int(sys.argv[3]) if len(sys.argv) > 3 else 540
n = int(sys.argv[4]) if len(sys.argv) > 4 else 4
inp = synth.pipeline_inputs(kind, w, h, n, flen=1.2 if kind == "sphere" else 1.0)
t = time.perf_counter(); got = host.optimize(inp, min_scale=2); t1 = time.perf_counter()
want = oracle.optimize(inp, regularization=0.01, num_iterations=5, min_scale=2); t2 = time.perf_counter()
key = lambda e: (e["scale"], e["iter"], e["newton_steps"], e["valid_patches"])
same = [key(e) for e in got["log"]] == [key(e) for e in want["log"]]
rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
print("%s %dx%d, %d neighbours: HIP %.2f s, oracle %.1f s; batches %d, identical control flow %s, "
      "same valid pixels %s, depth rel L2 %.3e, normals rel L2 %.3e"
      % (kind, w, h, n, t1 - t, t2 - t1, len(got["log"]), same,
         np.array_equal(got["depth"] > 0, want["depth"] > 0),
         rel(got["depth"], want["depth"]), rel(got["normals"], want["normals"])))
if not same:
    for a, b in zip(got["log"], want["log"]):
        print(key(a), key(b))
