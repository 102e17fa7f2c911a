"""GPU parity of the grid surgery on the device (csrc/surface.hip):
the scripted sequences of tests/test_host_surface_cpu.py -- Surface::create
from the bundle / from a depth map, expand, subdivide_patches,
fill_patches_from_depth, remove_isolated_patches, deletions -- run on the
surface of a device context and compared with

  * the C++ host mirror (csrc/host/surface.cc): geometry and validity
    identical, node values BIT-identical (both sides compile
    csrc/host/surface_math.h, contraction off);
  * the oracle's restatement of lib/surface.cc (oracle/smvs_oracle_opt.c):
    geometry and validity identical, node values as tight as the host mirror
    is against it (identical until a subdivision evaluates the bicubic patch in
    the other basis, then 1e-12).

Then the whole optimiser with the surface resident on the device against the
SMVS_HOST_SURGERY path (the host Surface with a round trip per batch).
"""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import smvs_amd
    if smvs_amd.device_count() < 1:
        pytest.fail("no HIP device visible: the GPU tests must run on a GPU")
    return smvs_amd


@pytest.fixture(scope="module")
def host(hip):
    from smvs_amd import host as h
    h.load()
    return h


@pytest.fixture(scope="module")
def scene():
    from smvs_amd import synth
    return synth.pipeline_inputs("sphere", 320, 240, 3, flen=1.2)


def _exact(ops):
    return not (2 in ops and len(ops) > 1)


def _same(a, b, exact=True):
    for k in ("scale", "npx", "npy", "start_x", "start_y"):
        assert a[k] == b[k], k
    assert np.array_equal(a["patch_valid"], b["patch_valid"])
    assert np.array_equal(a["node_valid"], b["node_valid"])
    m = a["node_valid"].astype(bool)
    if exact:
        assert np.array_equal(a["nodes"][m], b["nodes"][m])
    else:
        assert np.allclose(a["nodes"][m], b["nodes"][m], rtol=1e-12, atol=1e-13)


SCRIPTS = [[], [1], [1, 1], [2], [1, 2], [5], [5, 4], [1, 5, 4, 1], [2, 3],
           [1, 2, 5, 4, 1, 2, 3, 4], [2, 2, 5, 4, 1, 4], [2, 2, 2, 3, 1, 4]]


@pytest.mark.parametrize("ops", SCRIPTS)
def test_device_surface_script_from_bundle(host, oracle, scene, ops):
    """Surface::create from the bundle (scale 5 of a 320x240 view) and a
    script of operations."""
    got = host.surface_script(scene, 5, ops, delete_every=4, device=0)
    mirror = host.surface_script(scene, 5, ops, delete_every=4)
    want = oracle.surface_script(scene, 5, ops, delete_every=4)
    _same(got, mirror, exact=True)
    _same(got, want, exact=_exact(ops))
    assert got["valid_patches"] == int(want["patch_valid"].sum()) > 0


@pytest.mark.parametrize("delete_every", [3, 5])
def test_device_isolated_patch_removal_keeps_the_reference_order(host, oracle, scene,
                                                                 delete_every):
    """remove_isolated_patches deletes in place while it walks the grid column
    by column (surface.cc:887-927): with every third / fifth patch gone the
    deletions cascade and the visiting order is part of the result.  The
    device replays the walk as 2 npx + npy - 2 steps of independent cells."""
    ops = [2, 2, 5, 4, 1, 4]
    got = host.surface_script(scene, 5, ops, delete_every=delete_every, device=0)
    mirror = host.surface_script(scene, 5, ops, delete_every=delete_every)
    want = oracle.surface_script(scene, 5, ops, delete_every=delete_every)
    _same(got, mirror, exact=True)
    _same(got, want, exact=False)
    assert 0 < got["patch_valid"].sum() < got["patch_valid"].size


@pytest.mark.parametrize("init_scale", [6, 4, 3, 2, 1])
def test_device_surface_from_an_initial_depth_map(host, oracle, init_scale):
    """Surface::create with an initial depth map (surface.cc:46-50, the SGM
    path): a depth map with a step, a hole and scattered invalid pixels at
    every window size from 64 x 64 pixels per node (scale 6: the radix
    selection re-reads the window) to 2 x 2 (scale 1)."""
    from smvs_amd import synth
    scene = synth.pipeline_inputs("sphere", 416, 300, 2, flen=1.2)
    h, w = 300, 416
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    rng = np.random.default_rng(init_scale)
    depth = (4.0 + 0.002 * xx + 0.001 * yy + 0.01 * rng.standard_normal((h, w))
             ).astype(np.float32)
    depth[:, w // 2:] += 0.8
    depth[40:90, 60:140] = 0.0
    depth[::7, ::5] = 0.0
    depth[200:, 300:] = 0.0
    for ops in ([], [1], [2, 3], [1, 2, 3, 4]):
        if init_scale - ops.count(2) < 0:
            continue
        got = host.surface_script(scene, init_scale, ops, init_depth=depth, device=0)
        mirror = host.surface_script(scene, init_scale, ops, init_depth=depth)
        want = oracle.surface_script(scene, init_scale, ops, init_depth=depth)
        _same(got, mirror, exact=True)
        _same(got, want, exact=_exact(ops))
    assert got["patch_valid"].sum() > 0


def test_device_surface_full_size_script(host, oracle):
    """The grids of the bench workload: 1920x1080 from a depth map at scale 4,
    subdivided twice to scale 2 (479 x 269 nodes) with the fill and the
    isolated-patch walk (1,222 steps) in between -- against the host mirror
    (bit-identical) AND against the oracle's restatement of lib/surface.cc
    (validity identical, nodes to 1e-12 after the subdivisions)."""
    from smvs_amd import synth
    scene = synth.pipeline_inputs("sphere", 1920, 1080, 1, flen=1.2)
    truth = np.asarray(scene["truth"], dtype=np.float32).copy()
    rng = np.random.default_rng(7)
    truth[rng.random(truth.shape) < 0.05] = 0.0
    truth[300:420, 800:1000] = 0.0
    ops = [2, 3, 5, 4, 2, 3, 4]
    got = host.surface_script(scene, 4, ops, init_depth=truth, delete_every=7, device=0)
    mirror = host.surface_script(scene, 4, ops, init_depth=truth, delete_every=7)
    _same(got, mirror, exact=True)
    want = oracle.surface_script(scene, 4, ops, init_depth=truth, delete_every=7)
    _same(got, want, exact=False)
    assert int(want["patch_valid"].sum()) < want["patch_valid"].size   # deletions happened
    assert (got["scale"], got["npx"], got["npy"]) == (2, 478, 268)
    assert got["valid_patches"] == int(mirror["patch_valid"].sum()) > 10000


_AB_PROBE = r"""
import json, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from smvs_amd import synth, host
inputs = synth.pipeline_inputs("sphere", 480, 320, 3, flen=1.2)
sgm = host.sgm_depth(inputs, sgm_scale=1) if sys.argv[3] == "sgm" else None
out = host.optimize(inputs, regularization=0.01, num_iterations=4, min_scale=2,
                    sgm_depth=sgm, use_shading=(sys.argv[4] == "S"))
np.save(sys.argv[2], out["depth"])
print(json.dumps([[e["scale"], e["iter"], e["newton_steps"], e["valid_patches"],
                   e["cg_iterations"], int(e["active_patch_steps"])] for e in out["log"]]))
"""


@pytest.mark.parametrize("mode", [("sgm", "-"), ("bundle", "-"), ("sgm", "S")])
def test_optimize_with_the_surface_on_the_device_equals_host_surgery(hip, tmp_path, mode):
    """DepthOptimizer::optimize with the surface resident in the context (no
    upload / download per batch) against the same optimiser with
    SMVS_HOST_SURGERY=1 (host Surface, round trip per batch): the same batch
    log entry for entry -- Newton steps, valid patches, CG iterations, active
    patch-steps -- and bit-identical depth maps, with SGM initialisation, from
    the bundle (--no-sgm: expand + NCC) and shading-aware."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    logs, maps = [], []
    for tag, env_extra in (("dev", {}), ("host", {"SMVS_HOST_SURGERY": "1"})):
        env = dict(os.environ, **env_extra)
        env.pop("SMVS_HOST_SURGERY", None) if not env_extra else None
        out = str(tmp_path / ("depth_%s.npy" % tag))
        res = subprocess.run([sys.executable, "-c", _AB_PROBE, root, out, mode[0], mode[1]],
                             env=env, capture_output=True, text=True, timeout=900)
        assert res.returncode == 0, res.stderr[-3000:]
        logs.append(json.loads(res.stdout.strip().splitlines()[-1]))
        maps.append(np.load(out))
    assert logs[0] == logs[1], (logs[0], logs[1])
    assert len(logs[0]) >= 3
    assert np.array_equal(maps[0], maps[1])
    assert (maps[0] > 0).mean() > 0.2
