#!/usr/bin/env python3
"""Times smvs_sgm_run on the 960x540x128 workload of BASELINE.json configs[2]
(profiling helper; use under rocprofv3 for per-kernel numbers)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import smvs_amd
from smvs_amd import synth

w, h = (1920, 1080) if "--small" not in sys.argv else (480, 270)
main, subs = synth.ring_cameras(w, h, 1)
scene = synth.SphereScene(px_size=3.0 / (1.2 * w))
imgs = [synth.render(scene, c) for c in (main, subs[0])]
half = [im[::2, ::2].copy() for im in imgs]  # stand-in for rescale_half_size
hw, hh = half[0].shape[1], half[0].shape[0]
cm = synth.Camera(main.R, main.t, main.flen, hw, hh)
cs = synth.Camera(subs[0].R, subs[0].t, subs[0].flen, hw, hh)
M, t = synth.reprojection(cm, cs)
for i in range(3):
    t0 = time.perf_counter()
    out = smvs_amd.sgm_run(half[0], half[1], M.astype(np.float32), t.astype(np.float32), 2.0, 10.0, 128)
    dt = time.perf_counter() - t0
    print("sgm_run %dx%dx128: %.1f ms (incl. H2D/D2H), valid %.2f" % (hw, hh, 1e3 * dt, (out["depth"] > 0).mean()))
