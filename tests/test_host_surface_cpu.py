"""Host logic without a GPU: the C++ mirror of smvs::Surface
(csrc/host/surface.cc; lib/surface.cc:19-130 create / depth from the bundle,
:155-330 node initialisation from depth, :472-628 expand, :630-651 fill_holes,
:887-955 isolated patches / nodes without patch, :983-1107 subdivide) against
the oracle's restatement (oracle/smvs_oracle_opt.c) on the same inputs: grid
geometry and validity identical; node values identical where they are copied
or initialised from depth, and within 1e-12 where subdivide_patches evaluates
the bicubic patch (the host evaluates it from the Hermite basis it shares with
the device kernels, the oracle through the reference's coefficient form:
a few ulps apart)."""
import numpy as np
import pytest

from smvs_amd import synth


@pytest.fixture(scope="module")
def host():
    from smvs_amd import host as h
    h.load()
    return h


@pytest.fixture(scope="module")
def scene():
    return synth.pipeline_inputs("sphere", 320, 240, 3, flen=1.2)


def _exact(ops):
    # bit-identical until a subdivision evaluates patches whose nodes went
    # through another operation first (mixed derivative information)
    return not (2 in ops and len(ops) > 1)


def _same(a, b, exact=True):
    for k in ("scale", "npx", "npy", "start_x", "start_y"):
        assert a[k] == b[k], k
    assert np.array_equal(a["patch_valid"], b["patch_valid"])
    assert np.array_equal(a["node_valid"], b["node_valid"])
    # invalid nodes may hold anything on either side
    m = a["node_valid"].astype(bool)
    if exact:
        assert np.array_equal(a["nodes"][m], b["nodes"][m])
    else:
        assert np.allclose(a["nodes"][m], b["nodes"][m], rtol=1e-12, atol=1e-13)


@pytest.mark.parametrize("init_scale", [5, 4, 3])
def test_surface_create_from_bundle(host, oracle, scene, init_scale):
    got = host.surface_script(scene, init_scale, [])
    want = oracle.surface_script(scene, init_scale, [])
    _same(got, want)
    assert got["patch_valid"].sum() > 0


@pytest.mark.parametrize("ops", [[1], [1, 1], [2], [1, 2], [5], [5, 4], [1, 5, 4, 1],
                                 [2, 3], [1, 2, 5, 4, 1, 2, 3, 4]])
def test_surface_operations_match_oracle(host, oracle, scene, ops):
    got = host.surface_script(scene, 5, ops, delete_every=4)
    want = oracle.surface_script(scene, 5, ops, delete_every=4)
    _same(got, want, exact=_exact(ops))
    assert got["scale"] == 5 - ops.count(2)


@pytest.mark.parametrize("delete_every", [3, 5])
def test_isolated_patch_removal_keeps_the_reference_order(host, oracle, scene, delete_every):
    """remove_isolated_patches deletes in place while it walks the grid column
    by column (surface.cc:887-927): with every second / third patch gone the
    deletions cascade, so the result depends on the visiting order -- the
    host mirror (which walks a transposed copy) must still equal the oracle."""
    ops = [2, 2, 5, 4, 1, 4]
    got = host.surface_script(scene, 5, ops, delete_every=delete_every)
    want = oracle.surface_script(scene, 5, ops, delete_every=delete_every)
    _same(got, want, exact=False)
    assert 0 < got["patch_valid"].sum() < got["patch_valid"].size


def test_surface_from_an_initial_depth_map(host, oracle, scene):
    """Surface::create with the SGM-style initial depth (surface.cc:46-50):
    a depth map with holes and a discontinuity."""
    h, w = scene["images"][0].shape[:2]
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    depth = (4.0 + 0.002 * xx + 0.001 * yy).astype(np.float32)
    depth[:, w // 2:] += 0.8                      # a step
    depth[40:90, 60:140] = 0.0                    # a hole
    depth[::7, ::5] = 0.0                         # scattered invalid pixels
    for ops in ([], [1], [2], [1, 2, 3, 4]):
        got = host.surface_script(scene, 4, ops, init_depth=depth)
        want = oracle.surface_script(scene, 4, ops, init_depth=depth)
        _same(got, want, exact=_exact(ops))
    assert got["patch_valid"].sum() > 0


# ------------------------------------------------ host pieces of the SGM front end
def test_depth_range_matches_oracle(host, oracle, scene):
    """SGMStereo::fill_depth_range_for_view (lib/sgm_stereo.cc:669-720)."""
    for i in range(len(scene["cams"])):
        got = host.depth_range(scene, i)
        want = oracle.sgm_depth_range(scene, i)
        assert np.array_equal(got, want), i
        assert 0 < got[0] < got[1]


def test_reprojection_matches_oracle(host, oracle, scene):
    """CameraInfo::fill_reprojection [MVE-unverified, README M14], float."""
    for src, dst in ((0, 1), (1, 0), (2, 3), (3, 0)):
        M, t = host.view_reprojection(scene, src, dst)
        Mo, to = oracle.view_reprojection(scene, src, dst)
        assert np.array_equal(M, Mo) and np.array_equal(t, to), (src, dst)


@pytest.mark.parametrize("halvings", [0, 1, 2])
def test_sgm_input_image_matches_oracle(host, oracle, halvings):
    """get_byte_image (desaturate<uint8_t>) + rescale_half_size on an odd-sized
    colour image: u8, bit-exact."""
    inputs = synth.pipeline_inputs("sphere", 331, 247, 1, flen=1.2)
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, size=(247, 331, 3), dtype=np.uint8)
    inputs["images"][0] = img
    got = host.sgm_image(inputs, 0, halvings)
    want = oracle.sgm_image(inputs, 0, halvings)
    assert got.shape == want.shape and np.array_equal(got, want)


def test_view_queue_runs_every_task_once_and_binds_workers_to_devices():
    """smvs_amd::ViewQueue (the reference's one-task-per-view ThreadPool,
    lib/thread_pool.h, app/smvsrecon.cc:658-733, cut for GPUs): every task runs
    exactly once, worker w drives device w % num_devices, all devices get
    work, a task's exception arrives through its own future only."""
    from smvs_amd import host
    dev, wrk = host.view_queue_selftest(64, 8, 2, throwing_task=17)
    assert dev.sum() == 64 and wrk.sum() == 64
    assert (dev > 0).all()          # views spread over the 8 GPUs
    assert (wrk > 0).sum() >= 8
    dev, wrk = host.view_queue_selftest(5, 1, 3)
    assert dev.tolist() == [5]
    dev, wrk = host.view_queue_selftest(0, 2, 1)
    assert dev.sum() == 0
    with pytest.raises(Exception):
        host.view_queue_selftest(4, 0, 1)
