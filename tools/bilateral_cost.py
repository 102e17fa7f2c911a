#!/usr/bin/env python3
"""smvs_ctx_sgm_init_depth at 1920x1080 (a 960x540 map, RGB guidance, sigma 5,
half width 5): run under rocprofv3 --kernel-trace --stats to compare the two
forms of the colour-weight lookup (default: the triangle of all byte pairs in
LDS; SMVS_BILATERAL=compressed: 511 x 4 values + selector)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import smvs_amd

rng = np.random.default_rng(1)
W, H = 1920, 1080
img = rng.integers(0, 256, size=(H, W, 3)).astype(np.uint8)
low = (2.0 + rng.random((H // 2, W // 2))).astype(np.float32)
low[rng.random(low.shape) < 0.1] = 0.0
ctx = smvs_amd.ViewContext(W, H, 1)
ctx.upload_image(-1, img)
best = 1e9
for rep in range(6):
    t = time.perf_counter()
    ctx.sgm_init_depth(low)
    best = min(best, time.perf_counter() - t)
print("sgm_init_depth (%s): best host wall %.3f ms" % (os.environ.get("SMVS_BILATERAL", "triangle"), 1e3 * best))
ctx.close()
