#!/usr/bin/env python3
"""Randomised sweep of the grid surgery (not part of the suite): random image
sizes, initial scales, bundle / depth-map initialisation with random holes and
steps, random scripts of operations (1 expand, 2 subdivide, 3 fill from depth,
4 remove isolated patches, 5 delete every k-th patch) -- the surface of a
device context (csrc/surface.hip) against the C++ host mirror (bit for bit)
and against the oracle's restatement of lib/surface.cc (validity identical,
nodes identical or, once a subdivision evaluated patches, to 1e-12).
Usage: fuzz_surface.py [cases [seed [device|cpu]]] -- "cpu": host mirror vs
oracle only (no GPU needed)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from smvs_amd import synth, host
from oracle import pyoracle as oracle

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 50
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
use_device = not (len(sys.argv) > 3 and sys.argv[3] == "cpu")
host.load()


def same(a, b, exact, what):
    for k in ("scale", "npx", "npy", "start_x", "start_y"):
        assert a[k] == b[k], (what, k, a[k], b[k])
    assert np.array_equal(a["patch_valid"], b["patch_valid"]), (what, "patch_valid")
    assert np.array_equal(a["node_valid"], b["node_valid"]), (what, "node_valid")
    m = a["node_valid"].astype(bool)
    if exact:
        assert np.array_equal(a["nodes"][m], b["nodes"][m]), (what, "nodes")
    else:
        assert np.allclose(a["nodes"][m], b["nodes"][m], rtol=1e-12, atol=1e-13), (what, "nodes")


done = 0
for case in range(n_cases):
    w = int(rng.integers(96, 700)); h = int(rng.integers(72, 500))
    init_scale = int(rng.integers(2, 7))
    while ((min(w, h) - 2) >> init_scale) < 2:
        init_scale -= 1
    scene = synth.pipeline_inputs("sphere" if rng.random() < 0.7 else "plane", w, h, 2, flen=1.2)
    depth = None
    if rng.random() < 0.6:
        yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
        depth = (3.0 + 0.003 * xx + 0.002 * yy
                 + 0.02 * rng.standard_normal((h, w))).astype(np.float32)
        if rng.random() < 0.5:
            depth[:, int(w * rng.uniform(0.3, 0.7)):] += np.float32(rng.uniform(0.3, 1.5))
        for _ in range(int(rng.integers(0, 4))):
            y0, x0 = int(rng.integers(0, h - 8)), int(rng.integers(0, w - 8))
            depth[y0:y0 + int(rng.integers(4, h // 2)), x0:x0 + int(rng.integers(4, w // 2))] = 0.0
        depth[rng.random((h, w)) < rng.choice([0.0, 0.02, 0.3])] = 0.0
    n_ops = int(rng.integers(0, 7))
    ops, scale = [], init_scale
    for _ in range(n_ops):
        op = int(rng.choice([1, 2, 3, 4, 5]))
        if op == 2:
            if scale <= 1:
                continue
            scale -= 1
        ops.append(op)
    every = int(rng.integers(2, 9))
    try:
        want = oracle.surface_script(scene, init_scale, ops, init_depth=depth, delete_every=every)
    except RuntimeError:
        print("case %d skipped (the oracle rejects it)" % case); continue
    mirror = host.surface_script(scene, init_scale, ops, init_depth=depth, delete_every=every)
    exact = not (2 in ops and len(ops) > 1)
    same(mirror, want, exact, "host mirror vs oracle")
    if use_device:
        got = host.surface_script(scene, init_scale, ops, init_depth=depth, delete_every=every,
                                  device=0)
        same(got, mirror, True, "device vs host mirror")
        assert got["valid_patches"] == int(mirror["patch_valid"].sum())
    done += 1
    print("case %3d: %3dx%-3d scale %d -> %d, %s, ops %s every %d: %d / %d patches valid"
          % (case, w, h, init_scale, mirror["scale"], "depth map" if depth is not None else "bundle",
             ops, every, int(mirror["patch_valid"].sum()), mirror["patch_valid"].size))
print("cases %d, compared %d: all identical%s" % (n_cases, done, "" if use_device else " (host mirror vs oracle only)"))
