#!/usr/bin/env python3
"""StereoView::set_scale on the device (csrc/scale.hip): the planes of nine
1920x1080 RGB views at every scale of an optimize() (6 .. 2), with the y pass of
the blur, the luminance and the quadratic fit in one kernel (default) and as
blur_y_kernel + gradients_kernel (SMVS_SCALE_FUSED=0).  Host wall time of
smvs_ctx_set_scale (it ends with a wait for the context's stream), best of 9,
both forms alternating in one process."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import smvs_amd

w, h, n = 1920, 1080, 8
rng = np.random.default_rng(0)
ctx = smvs_amd.ViewContext(w, h, n)
for v in range(n + 1):
    ctx.upload_image(v - 1, rng.integers(0, 256, size=(h, w, 3)).astype(np.uint8))
best = {}
for rep in range(9):
    for fused in ("1", "0"):
        os.environ["SMVS_SCALE_FUSED"] = fused
        for scale in (6, 5, 4, 3, 2):
            ctx.synchronize()
            t = time.perf_counter()
            ctx.set_scale(scale)
            dt = time.perf_counter() - t
            key = (fused, scale)
            best[key] = min(best.get(key, 1e9), dt)
print("set_scale of 9 views 1920x1080x3, ms (best of 9)")
print("scale   separate    fused")
for scale in (6, 5, 4, 3, 2):
    print("%5d   %8.3f %8.3f" % (scale, 1e3 * best[("0", scale)], 1e3 * best[("1", scale)]))
print("sum     %8.3f %8.3f" % (1e3 * sum(best[("0", s)] for s in (6, 5, 4, 3, 2)),
                               1e3 * sum(best[("1", s)] for s in (6, 5, 4, 3, 2))))
ctx.close()
