"""GPU parity of the callers either side of the Newton loop: the SGM front end
(SGMStereo::reconstruct + merge, a19), the consumer of the depth maps
(MeshGenerator::cut_depth_maps, f-3), and the whole optimiser at
BASELINE.json's full size (configs[2] and [3]).  Integer / validity outputs
must match exactly; float depth maps of the bit-exact integer SGM path must be
array_equal; optimised depth within the north-star tolerance 1e-4 rel. L2.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300)


@pytest.fixture(scope="module")
def hip():
    import smvs_amd
    if smvs_amd.device_count() < 1:
        pytest.fail("no HIP device visible: the GPU tests must run on a GPU")
    return smvs_amd


@pytest.fixture()
def oracle_threads(oracle):
    """All host cores for the oracle's patch loops (results do not depend on
    the thread count: tests/test_oracle_core.py)."""
    n = max(1, min(os.cpu_count() or 1, 64))
    oracle.lib().orc_set_threads(n)
    yield n
    oracle.lib().orc_set_threads(1)


from parity_units import assert_same_units  # noqa: E402  (tests/parity_units.py)


# ------------------------------------------------------------ a19: SGM front
@pytest.mark.parametrize("size,fixed_range", [((384, 256), None), ((250, 170), None),
                                              ((384, 256), (2.0, 12.0))])
def test_sgm_front_end_matches_oracle(hip, oracle, size, fixed_range):
    """reconstruct_sgm_depth_for_view (app/smvsrecon.cc:346-384) through the
    C++ host mirror -> smvs_sgm_depth_for_view (4 x run_sgm, L/R check, merge
    on the device) against the oracle's own front end: depth range from the
    bundle (sgm_stereo.cc:669-720) or fixed, L/R check (:64-91), merge --
    array_equal (the SGM path is integer; the check's doubles follow the
    reference's operation order)."""
    from smvs_amd import synth, host
    inputs = synth.pipeline_inputs("sphere", size[0], size[1], 3, flen=1.2)
    lo, hi = fixed_range if fixed_range else (0.0, 0.0)
    got = host.sgm_depth(inputs, sgm_scale=1, min_depth=lo, max_depth=hi)
    want = oracle.sgm_depth_for_view(inputs, sgm_scale=1, min_depth=lo, max_depth=hi)
    assert got.shape == want.shape
    assert (want > 0).mean() > 0.3 and (want == 0).mean() > 0.01
    assert np.array_equal(got, want)


def test_device_lr_check_rejects_and_keeps(hip, oracle):
    """smvs_sgm_depth_for_view with one neighbour against the oracle's pieces
    (two run_sgm + orc_sgm_lr_check), on a pair whose second half is
    inconsistent (the neighbour image is scrambled there)."""
    rng = np.random.default_rng(5)
    w, h = 112, 72
    base = rng.integers(30, 220, size=(h + 8, w + 40)).astype(np.float32)
    base = (base + np.roll(base, 1, 0) + np.roll(base, 1, 1) + np.roll(base, -1, 1)) / 4
    main = base[4:4 + h, 8:8 + w].astype(np.uint8)
    nbr = base[4:4 + h, 11:11 + w].astype(np.uint8).copy()
    nbr[:, w // 2:] = rng.integers(30, 220, size=(h, w - w // 2)).astype(np.uint8)
    f = np.float32
    M = np.eye(3, dtype=f).reshape(9)
    t_fwd = np.array([-6.0, 0, 0], dtype=f)     # disparity 6 / depth
    t_bwd = np.array([6.0, 0, 0], dtype=f)
    rng_main, rng_nbr = (1.0, 12.0), (0.9, 11.0)
    got = hip.sgm_depth_for_view(main, [dict(image=nbr, M_fwd=M, t_fwd=t_fwd, M_bwd=M,
                                             t_bwd=t_bwd, range_main=rng_main,
                                             range_neighbor=rng_nbr)], num_steps=64)

    def run(a, b, t, lo, hi):
        depths = oracle.sgm_depths(lo, hi, 64)
        cost = oracle.sgm_cost_volume(a, b, M, t, depths)
        sgm = oracle.sgm_aggregate(cost, 6, 96)
        return oracle.sgm_depth_from_volume(sgm, a, depths)[0]

    d_main = run(main, nbr, t_fwd, *rng_main)
    d_neig = run(nbr, main, t_bwd, *rng_nbr)
    want = oracle.sgm_lr_check(d_main, d_neig, M, t_fwd)
    assert np.array_equal(got, want)
    kept = want > 0
    assert kept[:, : w // 2 - 12].mean() > 0.5          # consistent half survives
    assert (d_main > 0).sum() > kept.sum()               # the check removed pixels


# -------------------------------------------------------- f-3: cut_depth_maps
@pytest.mark.parametrize("n_views,size", [(3, (192, 128)), (5, (160, 120)), (1, (96, 64))])
def test_cut_depth_maps_matches_oracle(hip, oracle, n_views, size):
    """smvs_cut_depth_maps against the restated MeshGenerator::cut_depth_maps
    (mesh_generator.cc:24-158) on a synthetic sphere seen by n views, with a
    block of one view pushed off the surface, holes and noise: cut maps and
    world-space normals bit-identical."""
    from smvs_amd import synth
    inputs = synth.pipeline_inputs("sphere", size[0], size[1], max(n_views - 1, 1), flen=1.2)
    cams = inputs["cams"][:n_views]
    depths, normals = synth.depth_and_normal_maps(inputs["scene"], cams)
    rng = np.random.default_rng(9)
    for i in range(n_views):
        depths[i] *= (1.0 + 0.002 * rng.standard_normal(depths[i].shape)).astype(np.float32)
    depths[0][20:34, 40:60] *= np.float32(0.8)
    if n_views > 1:
        depths[1][5:9, 5:9] = 0.0
        depths[1][50:60, 30:50] *= np.float32(1.3)
    got_d, got_n = hip.cut_depth_maps(cams, depths, normals)
    want_d, want_n = oracle.cut_depth_maps(cams, depths, normals)
    for i in range(n_views):
        assert np.array_equal(got_n[i], want_n[i])
        assert np.array_equal(got_d[i], want_d[i])
    if n_views > 1:
        cut = sum(int(((d > 0) & (c == 0)).sum()) for d, c in zip(depths, want_d))
        kept = sum(int((c > 0).sum()) for c in want_d)
        assert cut > 100 and kept > cut
    else:
        assert np.array_equal(want_d[0], depths[0])


def test_cut_depth_maps_rejects_bad_arguments(hip):
    from smvs_amd._capi import SmvsError
    with pytest.raises(SmvsError):
        hip.cut_depth_maps([], [], [])


# -------------------------------------------- bilateral at its production size
def test_bilateral_upsample_full_size(hip, oracle, oracle_threads):
    """K17 at the size the optimiser uses it: 960x540 SGM depth -> 1920x1080,
    11x11 taps (depth_optimizer.cc:957-1004)."""
    rng = np.random.default_rng(17)
    dh, dw, h, w = 540, 960, 1080, 1920
    yy, xx = np.mgrid[0:dh, 0:dw].astype(np.float32)
    dm = (4.0 + 0.5 * np.sin(xx / 70.0) + 0.3 * np.cos(yy / 45.0)).astype(np.float32)
    dm[rng.random((dh, dw)) < 0.1] = 0.0
    dm[200:260, 300:420] = 0.0
    ci = rng.random((h, w, 3)).astype(np.float32)
    ci = (ci + np.roll(ci, 1, 0) + np.roll(ci, 1, 1)) / 3
    got = hip.bilateral_upsample(dm, ci)
    want = oracle.bilateral_upsample(dm, ci)
    assert np.array_equal(got > 0, want > 0)
    # exponentials rounded once from double on the device, glibc's expf on the
    # host, same float operation order: bit-identical except where one of the
    # ~500 exponentials of a pixel rounds differently (measured: 0.2 % of the
    # pixels, by one ulp)
    assert np.max(np.abs(got - want)) <= 1e-6 * np.max(np.abs(want))
    assert (got != want).mean() < 1e-2


# ------------------------------------ configs[2] and [3] end to end, full size
def _full_size_inputs(lighting=None):
    from smvs_amd import synth
    return synth.pipeline_inputs("sphere", 1920, 1080, 8, flen=1.2, lighting=lighting)


def test_full_size_optimize_with_sgm_matches_oracle(hip, oracle, oracle_threads):
    """BASELINE.json configs[2]: 1920x1080, 1 + 8 views, SGM initialisation
    (960x540 x 128 planes, 8 paths, two neighbours, L/R check, merge) feeding
    DepthOptimizer::optimize down to scale 2.  C++ host + HIP against the
    oracle running its own SGM front end and optimiser on the same images:
    the merged SGM depth is bit-identical, the batch log identical in the
    units the bench metric counts (scale, iteration, Newton steps, valid
    patches, ACTIVE PATCH-STEPS per batch exactly; CG iterations per batch
    within the written bound of tests/parity_units.py), the valid pixels
    identical and the depth within 1e-4 relative L2."""
    from smvs_amd import host
    inputs = _full_size_inputs()
    sgm = host.sgm_depth(inputs, sgm_scale=1)
    sgm_o = oracle.sgm_depth_for_view(inputs, sgm_scale=1)
    assert sgm.shape == (540, 960)
    assert np.array_equal(sgm, sgm_o)
    got = host.optimize(inputs, regularization=0.01, num_iterations=5, min_scale=2,
                        sgm_depth=sgm)
    want = oracle.optimize(inputs, regularization=0.01, num_iterations=5, min_scale=2,
                           sgm_depth=oracle.sgm_depth_for_view(inputs, sgm_scale=1,
                                                               roundtrip=True))
    assert_same_units(got["log"], want["log"], 1920, 1080, "configs2_sgm_1920x1080")
    assert np.array_equal(got["depth"] > 0, want["depth"] > 0)
    assert (want["depth"] > 0).mean() > 0.5
    assert _rel(got["depth"], want["depth"]) <= 1e-4
    truth = inputs["truth"]
    ok = want["depth"] > 0
    assert np.median(np.abs(got["depth"][ok] - truth[ok]) / truth[ok]) < 5e-3


def test_full_size_optimize_basic_matches_oracle(hip, oracle, oracle_threads):
    """BASELINE.json configs[1] as a whole -- the scene and options of
    `bench.py --workload optimize` / `secondary.optimize`: 1920x1080 textured
    sphere, 1 + 8 views, --no-sgm (surface from the bundle, NCC visibility
    test, expand), shading off, scales init .. 2, five iterations per scale.
    C++ host + HIP against the oracle's optimize(): identical batch log (scale,
    iteration, Newton steps, valid patches), identical valid pixels, depth
    within 1e-4 relative L2; the per-batch ACTIVE PATCH-STEPS -- the unit of the
    bench value -- are asserted equal and the CG iterations per batch within
    the written bound (tests/parity_units.py); the table goes to
    gpurun_out/r5_units_configs1_basic_1920x1080.txt."""
    from smvs_amd import host
    import bench
    inputs = _full_size_inputs()
    got = host.optimize(inputs, regularization=bench.REG, num_iterations=5,
                        min_scale=bench.SCALE)
    want = oracle.optimize(inputs, regularization=bench.REG, num_iterations=5,
                           min_scale=bench.SCALE)
    assert_same_units(got["log"], want["log"], 1920, 1080, "configs1_basic_1920x1080")
    assert np.array_equal(got["depth"] > 0, want["depth"] > 0)
    assert (want["depth"] > 0).mean() > 0.2
    print("configs[1] whole optimize: depth rel. L2 %.2e" % _rel(got["depth"], want["depth"]))
    assert _rel(got["depth"], want["depth"]) <= 1e-4
    truth = inputs["truth"]
    ok = want["depth"] > 0
    assert np.median(np.abs(got["depth"][ok] - truth[ok]) / truth[ok]) < 5e-3


def test_full_size_optimize_shading_aware_matches_oracle(hip, oracle, oracle_threads):
    """BASELINE.json configs[3]: 1920x1080, 1 + 8 views, -S (GlobalLighting SH
    fit at scales < 4 + shading residual), no SGM: identical batch log and
    valid pixels, lighting coefficients to 1e-3, depth within 1e-4 rel. L2."""
    from smvs_amd import host
    rng = np.random.default_rng(3000)
    lighting = np.zeros(16); lighting[0] = 0.9
    lighting[1:4] = rng.uniform(-0.2, 0.2, 3)
    inputs = _full_size_inputs(lighting)
    got = host.optimize(inputs, regularization=0.01, num_iterations=5, min_scale=2,
                        use_shading=True)
    want = oracle.optimize(inputs, regularization=0.01, num_iterations=5, min_scale=2,
                           use_shading=True)
    assert_same_units(got["log"], want["log"], 1920, 1080, "configs3_shading_1920x1080")
    assert got["lighting"] is not None and want["lighting"] is not None
    assert _rel(got["lighting"], want["lighting"]) < 1e-3
    assert np.array_equal(got["depth"] > 0, want["depth"] > 0)
    assert _rel(got["depth"], want["depth"]) <= 1e-4


def test_full_size_view_of_config5_sgm_and_shading_matches_oracle(hip, oracle, oracle_threads):
    """BASELINE.json configs[4]'s per-view task at its real size on one rank
    (app/smvsrecon.cc:693-720): 1920x1080, 1 + 8 views, the SGM initialisation
    (960x540 x 128 planes, two neighbours, L/R check, merge, bilateral
    upsample) AND -S (GlobalLighting SH fit per view + shading residual)
    together, down to scale 2.  C++ host + HIP against the oracle's own SGM
    front end and optimiser: SGM depth bit-identical, batch log identical in
    the metric's units (tests/parity_units.py), lighting to 1e-3, valid pixels
    identical, depth within 1e-4 relative L2.  (The 64-view / 8-GPU sharding
    of configs[4] repeats this task per view: test_shard_cpu.py,
    test_config5_whole_pipeline_round_robin.)"""
    from smvs_amd import host
    rng = np.random.default_rng(4242)
    lighting = np.zeros(16); lighting[0] = 0.85
    lighting[1:4] = rng.uniform(-0.2, 0.2, 3)
    inputs = _full_size_inputs(lighting)
    sgm = host.sgm_depth(inputs, sgm_scale=1)
    got = host.optimize(inputs, regularization=0.01, num_iterations=5, min_scale=2,
                        use_shading=True, sgm_depth=sgm)
    sgm_o = oracle.sgm_depth_for_view(inputs, sgm_scale=1, roundtrip=True)
    assert np.array_equal(got["sgm_roundtrip"], sgm_o)
    want = oracle.optimize(inputs, regularization=0.01, num_iterations=5, min_scale=2,
                           use_shading=True, sgm_depth=sgm_o)
    assert_same_units(got["log"], want["log"], 1920, 1080, "configs4_view_sgm_shading_1920x1080")
    assert got["lighting"] is not None and want["lighting"] is not None
    assert _rel(got["lighting"], want["lighting"]) < 1e-3
    assert np.array_equal(got["depth"] > 0, want["depth"] > 0)
    assert (want["depth"] > 0).mean() > 0.5
    assert _rel(got["depth"], want["depth"]) <= 1e-4


# ---- the reference's other operating points, whole optimize() against the oracle
# (app/smvsrecon.cc:43-46: six neighbours, output scale 1 by default; -o3;
# portrait images; -S with --gamma-srgb, lib/stereo_view.cc:64-84)
def _operating_point(oracle, tag, width, height, n_subs, min_scale, **kw):
    """Whole optimize() against the oracle at one operating point.  The batch
    log must be identical in every unit up to and including the first batch
    whose CG iteration counts differ (inside the written bound); behind such a
    batch the two surfaces differ by what one more or fewer PCG iteration
    leaves, a validity decision on its threshold can go the other way, and the
    later batches are held to the written drift (tests/parity_units.py:
    max(2, 5e-4 x count) patches) -- measured in round 6: 3 of 12,500 patches at
    -o3 portrait after a 66 / 65 solve, 2 of 120,295 at scale 1 after solves
    that ran into the iteration limit."""
    from smvs_amd import synth, host
    lighting = kw.pop("lighting", None)
    kw_drift = kw.pop("drift_after_divergence", True)
    inputs = synth.pipeline_inputs("sphere", width, height, n_subs, flen=1.2, lighting=lighting)
    got = host.optimize(inputs, regularization=0.01, num_iterations=5, min_scale=min_scale, **kw)
    want = oracle.optimize(inputs, regularization=0.01, num_iterations=5, min_scale=min_scale,
                           **kw)
    assert min(e["scale"] for e in want["log"]) == min_scale
    exact = assert_same_units(got["log"], want["log"], width, height, tag,
                              drift_after_divergence=kw_drift)
    assert (want["depth"] > 0).mean() > 0.2
    if exact or not kw_drift:
        assert np.array_equal(got["depth"] > 0, want["depth"] > 0)
        print("%s: depth rel. L2 %.2e" % (tag, _rel(got["depth"], want["depth"])))
        assert _rel(got["depth"], want["depth"]) <= 1e-4
    else:
        # solves ended apart at the iteration limit (see assert_same_units): the
        # masks may differ where a validity decision sat on its threshold, the
        # common pixels to the bound written for a solve that ended apart
        both = (got["depth"] > 0) & (want["depth"] > 0)
        differ = ((got["depth"] > 0) != (want["depth"] > 0)).mean()
        rel = _rel(got["depth"][both], want["depth"][both])
        print("%s: solves ended apart at the iteration limit; masks differ on %.2e of the "
              "pixels, depth rel. L2 on the common ones %.2e" % (tag, differ, rel))
        assert differ <= 5e-4 and rel <= 5e-4
    return got, want, inputs


def test_optimize_default_operating_point_six_neighbours_scale_1(hip, oracle, oracle_threads):
    """smvsrecon's defaults: six neighbours, optimisation down to scale 1
    (patches of 2 x 2 pixels, one sample per pixel): 960x540, so that the
    finest grid (479 x 269 patches, 129,600 nodes) is as large as the bench
    workload's and uses the resident solver at its capacity.  At scale 1 the
    block-Jacobi PCG runs into its 200-iteration limit (device and oracle
    alike: 200 / 200, 200 / 198, 200 / 189 in round 6's run), so the batches
    behind the first solve that ended apart are held to the written drift
    (tests/parity_units.py) instead of identity."""
    _operating_point(oracle, "defaults_6_neighbours_scale1_960x540", 960, 540, 6, 1,
                     drift_after_divergence=True)


def test_optimize_output_scale_3_portrait(hip, oracle, oracle_threads):
    """-o3 on a PORTRAIT image (height > width: fill_calibration's max(w, h),
    the tile plans of a tall node grid), four neighbours."""
    _operating_point(oracle, "scale3_portrait_720x1280", 720, 1280, 4, 3)


def test_optimize_scale_1_portrait_five_neighbours(hip, oracle, oracle_threads):
    """Scale 1 on a small portrait image with an odd neighbour count."""
    _operating_point(oracle, "scale1_portrait_300x420", 300, 420, 5, 1)


def test_optimize_shading_with_gamma_correction(hip, oracle, oracle_threads):
    """-S --gamma-srgb (app/smvsrecon.cc:52, 110, 669): the main view's linear
    image goes through gamma_correct_inv_srgb before it is desaturated into the
    shading image (lib/stereo_view.cc:64-84; tests/golden/README.md M9 -- the
    one path of StereoView no other test exercises).  The lighting fit and the
    shading rows then see a different image than without the flag."""
    rng = np.random.default_rng(77)
    lighting = np.zeros(16); lighting[0] = 0.9
    lighting[1:4] = rng.uniform(-0.2, 0.2, 3)
    got, want, inputs = _operating_point(oracle, "shading_gamma_640x480", 640, 480, 3, 2,
                                         lighting=lighting, use_shading=True,
                                         gamma_correction=True)
    assert got["lighting"] is not None and want["lighting"] is not None
    assert _rel(got["lighting"], want["lighting"]) < 1e-3
    # the flag changes the result (the test would pass vacuously otherwise)
    from smvs_amd import host
    plain = host.optimize(inputs, regularization=0.01, num_iterations=5, min_scale=2,
                          use_shading=True)
    assert _rel(plain["lighting"], got["lighting"]) > 1e-3


# ------------------------- configs[4] code path (views sharded, shading-aware)
def test_config5_lighting_round_on_device_buffers(hip, oracle):
    """The --config 5 path of bench.py on one rank with two views in a
    lock-step round (light_optimizer.cc:32-55): per-view mode
    (the reference's behaviour) = each view's own normal equations; shared
    mode = the device buffers of smvs_light_accumulate_dev summed in place by
    shard.allreduce_lighting_device (RCCL when world > 1, no host hop) against
    the oracle summing A, b over the round; then a shading-aware construction
    with the solved lighting against the oracle."""
    import torch
    from smvs_amd import synth, shard
    prob = synth.make_problem(256, 192, 3, 2, shading=True, noise=0.003)
    surfs = [prob["surf"], dict(prob["surf"])]
    rng = np.random.default_rng(77)
    nodes = prob["surf"]["nodes"].copy()
    nodes[:, 0] *= 1.0 + 0.002 * rng.standard_normal(nodes.shape[0])
    surfs[1]["nodes"] = nodes
    ctxs, refs = [], []
    for s in surfs:
        c = hip.ViewContext(256, 192, 3)
        c.set_views(prob["views"]); c.set_surface(s)
        ctxs.append(c)
        orc = oracle.OracleProblem(s, prob["views"])
        refs.append(oracle.light_accumulate(orc.normal_map(), prob["views"]["shading"]))
    device = torch.device("cuda", 0)
    # per-view lighting (parity mode)
    for c, (A_ref, b_ref) in zip(ctxs, refs):
        c.light_accumulate_dev()
        A, b = c.light_download()
        assert _rel(A, A_ref) < 1e-10 and _rel(b, b_ref) < 1e-10
    # shared lighting over the round
    ptrs = [c.light_accumulate_dev() for c in ctxs]
    shard.allreduce_lighting_device(ptrs, None, device)
    A_sum = refs[0][0] + refs[1][0]; b_sum = refs[0][1] + refs[1][1]
    for c in ctxs:
        A, b = c.light_download()
        assert _rel(A, A_sum) < 1e-10 and _rel(b, b_sum) < 1e-10
    lighting = shard.solve_lighting(A_sum, b_sum)
    want = oracle.light_solve(A_sum, b_sum)
    assert _rel(lighting, want) < 1e-3        # 16x16 SH system is ill-conditioned
    # the Newton step consumes it
    n = ctxs[1].gn_construct(0.01, 0.0, lighting)
    H9, g, P = ctxs[1].gn_download()
    ref = oracle.OracleProblem(surfs[1], prob["views"]).gn_construct(
        surfs[1]["node_valid"], 0.01, 0.0, lighting)
    assert n == ref["active_patches"]
    assert _rel(H9, ref["H9"]) < 1e-9 and _rel(g, ref["g"]) < 1e-9
    for c in ctxs:
        c.close()


def test_config5_whole_pipeline_round_robin(hip, oracle, oracle_threads):
    """configs[4] as one rank runs it: four reference views round-robin, each
    through the WHOLE per-view pipeline with -S -- SGM initialisation (the
    default of smvsrecon, app/smvsrecon.cc:693-709), optimize() with the scale
    loop, the SH lighting fit per view and the shading residual -- on one GPU,
    against the oracle's pipeline of the same view: SGM depth array_equal,
    identical batch logs, same valid pixels, depth within 1e-4, lighting
    within 1e-3 (ill-conditioned 16 x 16 system).
    (Scenes: SGM-initialised, where the oracle's SSE and scalar photometric
    branches -- 4e-16 apart in H -- produce the same batch log.  The small
    --no-sgm scenes first tried here do not have that property: the reference's
    own two branches take different numbers of Newton steps on them, so they
    cannot pin anything.)"""
    from smvs_amd import synth, host
    rng = np.random.default_rng(4100)
    for view in range(4):
        lighting = np.zeros(16); lighting[0] = 0.85
        lighting[1:4] = rng.uniform(-0.2, 0.2, 3)
        w, h = 384 + 32 * view, 256 + 16 * (view % 3)
        inputs = synth.pipeline_inputs("sphere", w, h, 3, flen=1.2, lighting=lighting)
        sgm = host.sgm_depth(inputs, sgm_scale=1)
        got = host.optimize(inputs, regularization=0.01, num_iterations=3, min_scale=2,
                            use_shading=True, sgm_depth=sgm)
        sgm_o = oracle.sgm_depth_for_view(inputs, sgm_scale=1, roundtrip=True)
        assert np.array_equal(got["sgm_roundtrip"], sgm_o), view
        want = oracle.optimize(inputs, regularization=0.01, num_iterations=3,
                               min_scale=2, use_shading=True, sgm_depth=sgm_o)
        assert_same_units(got["log"], want["log"], w, h, "config5_round_robin_view%d" % view)
        assert got["lighting"] is not None and want["lighting"] is not None
        assert _rel(got["lighting"], want["lighting"]) < 1e-3
        assert np.array_equal(got["depth"] > 0, want["depth"] > 0)
        assert _rel(got["depth"], want["depth"]) <= 1e-4, view


def test_surface_constructor_maps_without_optimize(hip, oracle):
    """DepthOptimizer(main, subs, Surface::Ptr, opts).get_depth() /
    get_normals() before any optimize() (lib/depth_optimizer.h:53-61): the maps
    of the surface as it was passed in."""
    from smvs_amd import synth, host
    inputs = synth.pipeline_inputs("plane", 320, 240, 2)
    depth, normals = host.surface_maps(inputs, init_scale=4)
    surf = oracle.surface_script(inputs, 4, [])
    surf.update(width=320, height=240,
                patch_vis=np.zeros(surf["npx"] * surf["npy"], np.uint32))
    views = dict(M=np.eye(3).reshape(1, 9), t=np.zeros((1, 3)), flen=1.0,
                 inv_flen=float(np.float32(1.0) / (np.float32(inputs["cams"][0].flen)
                                                   * np.float32(320))),
                 grad=np.zeros((240, 320, 2), np.float32),
                 subs=[(np.zeros((240, 320, 2), np.float32),
                        np.zeros((240, 320, 3), np.float32))])
    orc = oracle.OracleProblem(surf, views)
    want = orc.depth_map()
    assert (want > 0).mean() > 0.3
    assert np.array_equal(depth == 0, want == 0)
    assert _rel(depth, want) < 1e-6
    n_want = orc.normal_map()
    assert np.max(np.abs(normals - n_want)) < 1e-5


def test_view_queue_pipeline_matches_single_view(hip, oracle):
    """smvs_host_optimize_views: six per-view tasks (StereoViews, SGM front end,
    optimize, maps) on a ViewQueue with three views in flight on one GPU --
    every job has the batch log of the same view run alone, and the maps of a
    job in the middle of the queue are those of the single run, bit for bit
    (concurrent views share the GPU, not state)."""
    from smvs_amd import synth, host
    inputs = synth.pipeline_inputs("sphere", 384, 256, 3, flen=1.2)
    sgm = host.sgm_depth(inputs, sgm_scale=1)
    single = host.optimize(inputs, regularization=0.01, num_iterations=3, min_scale=2,
                           sgm_depth=sgm)
    many = host.optimize_views(inputs, 6, regularization=0.01, num_iterations=3,
                               min_scale=2, sgm_scale=1, views_in_flight=3, keep_job=4)
    assert len(many["logs"]) == 6 and many["total_seconds"] > 0
    strip = lambda log: [{k: v for k, v in e.items() if k != "loop_seconds"} for e in log]
    for log in many["logs"]:
        assert strip(log) == strip(single["log"])
    assert np.array_equal(many["depth"], single["depth"])
    assert np.array_equal(many["normals"], single["normals"])
    assert hip._capi.load().smvs_release_workspaces() >= 1


def test_reference_signature_classes_match_oracle(hip, oracle):
    """smvs_amd::GaussNewtonStep::construct(surface,
    subsurfaces, active, lighting, &H, &g, &P) and
    smvs_amd::ConjugateGradient::solve(H, -g, &x, &P) -- the reference's
    signatures (lib/gauss_newton_step.h:46-50, lib/conjugate_gradient.h:55-56)
    over smvs_gn_construct / smvs_gn_download / smvs_gn_upload / smvs_cg_solve --
    against the oracle on the planes and reprojections the classes used."""
    from smvs_amd import synth, host
    inputs = synth.pipeline_inputs("plane", 320, 240, 2)
    got = host.gn_solve_step(inputs, init_scale=4, regularization=0.01)
    surf = got["surf"]
    assert surf["patch_valid"].sum() > 50
    orc = oracle.OracleProblem(surf, got["views"])
    ref = orc.gn_construct(surf["node_valid"], 0.01)
    assert _rel(got["H9"], ref["H9"]) < 1e-9
    assert np.all(got["H9"][ref["present"] == 0] == 0.0)
    assert _rel(got["g"], ref["g"]) < 1e-9
    assert _rel(got["P"], ref["P"]) < 1e-6
    xr, itr, infor = orc.cg_solve(ref["H9"], ref["present"], ref["P"], -ref["g"], 200,
                                  0.01 * np.linalg.norm(ref["g"]), 1e-3)
    assert (got["iterations"], got["info"]) == (itr, infor)
    assert _rel(got["x"], xr) < 1e-7


def test_native_rccl_lighting_allreduce_world_1(hip, oracle):
    """include/smvs_rccl.h: one-rank communicator (ncclCommInitRank), the
    normal equations of three views of a lock-step round summed on the device
    and written back to every context -- against the oracle's sums.  (The
    cross-rank step is ncclAllReduce on the same buffer; the driver's
    multi-GPU run is what exercises it.)"""
    from smvs_amd import synth, shard
    prob = synth.make_problem(224, 160, 3, 2, shading=True, noise=0.003)
    rng = np.random.default_rng(12)
    ctxs, refs = [], []
    for v in range(3):
        s = dict(prob["surf"])
        nodes = prob["surf"]["nodes"].copy()
        nodes[:, 0] *= 1.0 + 0.002 * v * rng.standard_normal(nodes.shape[0])
        s["nodes"] = nodes
        c = hip.ViewContext(224, 160, 3)
        c.set_views(prob["views"]); c.set_surface(s)
        c.light_accumulate_dev()
        ctxs.append(c)
        orc = oracle.OracleProblem(s, prob["views"])
        refs.append(oracle.light_accumulate(orc.normal_map(), prob["views"]["shading"]))
    comm = shard.NativeComm(0, None)
    assert (comm.rank, comm.world) == (0, 1)
    assert comm.ranks() == (1, 0)          # ncclCommCount, ncclCommUserRank
    comm.allreduce_lighting(ctxs)
    A_sum = sum(r[0] for r in refs); b_sum = sum(r[1] for r in refs)
    for c in ctxs:
        A, b = c.light_download()
        assert _rel(A, A_sum) < 1e-10 and _rel(b, b_sum) < 1e-10
    # smvs_light_upload: the pattern bench.py --gpus N sends through the
    # all-reduce to prove that every rank's buffer was summed
    pattern = np.arange(1.0, 273.0)
    ctxs[0].light_upload(3.0 * pattern[:256], 3.0 * pattern[256:])
    comm.allreduce_lighting(ctxs[:1])
    A, b = ctxs[0].light_download()
    assert np.array_equal(A.reshape(-1), 3.0 * pattern[:256])
    assert np.array_equal(b, 3.0 * pattern[256:])
    comm.close()
    for c in ctxs:
        c.close()


_DEBUG_PROBE = r"""
import json, sys
import numpy as np
import torch  # noqa: F401  (one HIP runtime, see conftest.py)
sys.path.insert(0, sys.argv[1])
from smvs_amd import synth, host
inputs = synth.pipeline_inputs("sphere", 384, 256, 3, flen=1.2)
out = host.optimize(inputs, regularization=0.01, num_iterations=3, min_scale=2,
                    use_shading=True)
sys.stdout.flush()
print("LOG " + json.dumps([[e["scale"], e["iter"], e["newton_steps"], e["valid_patches"],
                            e["cg_iterations"]] for e in out["log"]]))
"""


def test_debug_lvl_prints_the_reference_report(hip):
    """DepthOptimizer::Options::debug_lvl = 1 (lib/depth_optimizer.cc:58-60,
    84-86, 92-94, 112-113, 132-134, 185-187, 306-316): the per-scale banner and
    time, the valid patches of a scale's first batch, and per batch the number
    of Newton steps and the average solver iterations -- the same numbers as
    get_log(), in the reference's wording."""
    import json, re, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SMVS_DEBUG_LVL="1")
    res = subprocess.run([sys.executable, "-c", _DEBUG_PROBE, root], env=env,
                         capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    text = res.stdout
    log = json.loads([l for l in text.splitlines() if l.startswith("LOG ")][-1][4:])
    scales = [int(m) for m in re.findall(r"#{11} Scale (\d+) #{11}", text)]
    assert scales == sorted({e[0] for e in log}, reverse=True) and scales[-1] == 2
    assert [int(m) for m in re.findall(r"Scale (\d+) took [0-9.e+-]+s", text)] == scales
    assert text.count("######## with Lighting ########") == sum(1 for k in scales[1:] if k < 4)
    assert [int(m) for m in re.findall(r"### Finished iteration: (\d+)", text)] \
        == [e[1] for e in log]
    assert [int(m) for m in re.findall(r"Number of Newton steps: (\d+)", text)] \
        == [e[2] for e in log]
    assert [int(m) for m in re.findall(r"Surface Status - Valid patches: (\d+)", text)] \
        == [e[3] for e in log if e[1] == 0]
    assert [int(m) for m in re.findall(r"Avg solver iterations: (\d+)", text)] \
        == [e[4] // e[2] if e[2] else 0 for e in log]
    times = re.findall(r"Avg construction time: ([0-9.e+-]+)ms", text)
    assert len(times) == len(log) and all(float(t) > 0 for t in times)
    # ... and nothing is printed without it
    res0 = subprocess.run([sys.executable, "-c", _DEBUG_PROBE, root],
                          env={k: v for k, v in os.environ.items() if k != "SMVS_DEBUG_LVL"},
                          capture_output=True, text=True, timeout=900)
    assert res0.returncode == 0 and "Scale" not in res0.stdout


_DEBUG2_PROBE = r"""
import json, sys
import numpy as np
import torch  # noqa: F401  (one HIP runtime, see conftest.py)
sys.path.insert(0, sys.argv[1])
from smvs_amd import synth, host
lighting = np.zeros(16); lighting[0] = 0.9; lighting[2] = 0.15
inputs = synth.pipeline_inputs("sphere", 384, 256, 3, flen=1.2, lighting=lighting)
sgm = host.sgm_depth(inputs, sgm_scale=1)
out = host.optimize(inputs, regularization=0.01, num_iterations=3, min_scale=2,
                    use_shading=True, sgm_depth=sgm)
emb = host.last_embeddings()
np.savez(sys.argv[2], depth=out["depth"], normals=out["normals"], lighting=out["lighting"],
         flen=inputs["cams"][0].flen, **{k.replace("-", "_"): v for k, v in emb.items()})
print("NAMES " + json.dumps(sorted(emb)))
"""


def test_debug_lvl_2_leaves_the_reference_embeddings(hip, tmp_path):
    """Options::debug_lvl = 2 (lib/depth_optimizer.cc:44-45, 68-70, 119-127,
    139-156, depth_optimizer.h:150-160): the filtered SGM map, the initial depth,
    one depth map per scale (and '-exp' after an expansion, which the SGM mode
    does not run), the rendered shading, the rendered sphere and the implicit
    albedo are left in the main view beside the results; the shading image is
    the lighting applied to the final normals (GlobalLighting::render_normal_map),
    the last scale's depth map is the result."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "emb.npz")
    res = subprocess.run([sys.executable, "-c", _DEBUG2_PROBE, root, out],
                         env=dict(os.environ, SMVS_DEBUG_LVL="2"), capture_output=True,
                         text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    names = json.loads([l for l in res.stdout.splitlines() if l.startswith("NAMES ")][-1][6:])
    for want in ("smvs", "smvsN", "smvs-sgm", "smvs-sgm-filtered", "smvs-initial", "smvs-L2",
                 "smvs-L3", "smvs-L4", "smvs-shaded", "smvs-shaded-sphere",
                 "smvs-implicit-albedo"):
        assert want in names, (want, names)
    z = np.load(out)
    assert z["smvs_shaded_sphere"].shape == (555, 555)
    # the scale-2 debug depth is the result embedding (one converted on the host,
    # one where it is computed)
    assert np.allclose(z["smvs_L2"], z["smvs"], rtol=1e-6, atol=0)
    assert not np.array_equal(z["smvs_L3"], z["smvs"]) and (z["smvs_initial"] > 0).any()
    # smvs-shaded = lighting . sh(normal) wherever there is a surface
    n = z["normals"].astype(np.float64)
    x, y, zz = n[..., 0], n[..., 1], n[..., 2]
    x2, y2, z2 = x * x, y * y, zz * zz
    sh = np.stack([np.ones_like(x), y, zz, x, x * y, y * zz, -x2 - y2 + 2 * z2, x * zz, x2 - y2,
                   (3 * x2 - y2) * y, x * y * zz, (4 * z2 - x2 - y2) * y,
                   (2 * z2 - 3 * x2 - 3 * y2) * zz, (4 * z2 - x2 - y2) * x, (x2 - y2) * zz,
                   (x2 - 3 * y2) * x], axis=-1)
    want = (sh * z["lighting"]).sum(-1)
    unit = np.abs(np.sqrt(x2 + y2 + z2) - 1.0) <= 1e-6
    assert unit.mean() > 0.3
    assert np.allclose(z["smvs_shaded"][unit], want[unit], rtol=1e-5, atol=1e-6)
    assert np.all(z["smvs_shaded"][~unit] == 0)
    # the stored depths are in MVE's ray-length convention: depth x |K^-1 (x + .5, y + .5, 1)|
    h, w = z["depth"].shape
    ax = float(z["flen"]) * max(w, h)
    ys, xs = np.mgrid[0:h, 0:w]
    length = np.sqrt(((xs + 0.5 - w / 2) / ax) ** 2 + ((ys + 0.5 - h / 2) / ax) ** 2 + 1.0)
    ok = z["depth"] > 0
    assert np.allclose(z["smvs"][ok], (z["depth"] * length)[ok], rtol=1e-5)
    # without the level nothing but the results (and the input) is written
    res0 = subprocess.run([sys.executable, "-c", _DEBUG2_PROBE, root, out],
                          env={k: v for k, v in os.environ.items() if k != "SMVS_DEBUG_LVL"},
                          capture_output=True, text=True, timeout=900)
    names0 = json.loads([l for l in res0.stdout.splitlines() if l.startswith("NAMES ")][-1][6:])
    assert sorted(names0) == ["smvs", "smvs-sgm", "smvsN"]


_DEVICE_MAP_PROBE = r"""
import json, sys
import numpy as np
import torch  # noqa: F401  (one HIP runtime, see conftest.py)
sys.path.insert(0, sys.argv[1])
import smvs_amd
from smvs_amd import synth, host
inputs = synth.pipeline_inputs("sphere", 384, 256, 3, flen=1.2)
many = host.optimize_views(inputs, 5, regularization=0.01, num_iterations=3, min_scale=2,
                           sgm_scale=1, num_devices=int(sys.argv[3]), views_in_flight=2,
                           keep_job=3)
np.save(sys.argv[2], many["depth"])
strip = lambda log: [{k: v for k, v in e.items() if k != "loop_seconds"} for e in log]
print(json.dumps(dict(devices=smvs_amd.device_count(), logs=[strip(l) for l in many["logs"]])))
"""


def test_device_map_runs_the_multi_device_code_on_one_gpu(hip, tmp_path):
    """SMVS_DEVICE_MAP=0,0 (csrc/common.h): two LOGICAL devices on the one GPU
    of this box, so that ViewQueue(num_devices = 2, views_in_flight = 2) --
    workers bound to different devices, per-device context / workspace / pinned
    pools, the tile budget shared by the logical devices of one GPU -- runs in
    the one-GPU suite (app/smvsrecon.cc:658-733 is the unit being
    parallelised).  Every job has the batch log and the kept job the depth map,
    bit for bit, of the same queue on one logical device.  (What this cannot
    rehearse: two physical GPUs, and RCCL across ranks -- the scaling curve
    stays unmeasured by the builder)"""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    replies, maps = [], []
    for tag, env_extra, ndev in (("one", {}, 1), ("two", {"SMVS_DEVICE_MAP": "0,0"}, 2)):
        env = dict(os.environ, SMVS_LOCK_DIR=str(tmp_path), **env_extra)
        out = str(tmp_path / (tag + ".npy"))
        res = subprocess.run([sys.executable, "-c", _DEVICE_MAP_PROBE, root, out, str(ndev)],
                             env=env, capture_output=True, text=True, timeout=900)
        assert res.returncode == 0, res.stderr[-3000:]
        replies.append(json.loads(res.stdout.strip().splitlines()[-1]))
        maps.append(np.load(out))
    assert replies[1]["devices"] == 2 and replies[0]["devices"] >= 1
    assert len(replies[1]["logs"]) == 5
    for log in replies[1]["logs"]:
        assert log == replies[0]["logs"][0]
    assert np.array_equal(maps[0], maps[1]) and (maps[0] > 0).mean() > 0.2


def test_bench_two_ranks_on_the_logical_devices_of_one_gpu(hip, tmp_path):
    """`bench.py --gpus 2` rehearsed on ONE GPU: SMVS_DEVICE_MAP=0,0 and the
    gloo backend for the launcher's barrier (RCCL refuses two ranks on one
    device).  Exercises what the driver's multi-GPU run does and the one-GPU
    bench does not: the self-spawn under torch.distributed.run, two ranks whose
    Newton loops take turns on the GPU through the cross-process lock file, the
    barrier-bracketed timing with max over ranks and summed units, then whole
    views as two processes and as one ViewQueue(2, in_flight).  The numbers mean
    nothing (two ranks share a GPU); the JSON contract and the code paths do."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SMVS_DEVICE_MAP="0,0", SMVS_BENCH_DIST_BACKEND="gloo",
               SMVS_LOCK_DIR=str(tmp_path))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--small", "--steps",
           "4", "--warmup", "1", "--repeats", "2", "--views-per-rank", "3"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, cwd=root, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    out = json.loads([l for l in res.stdout.splitlines() if l.strip().startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["scaling"] == "weak"
    v = out["secondary"]["views_per_s"]
    for mode in ("one_process_per_gpu", "one_process_view_queue", "single_gpu_reference"):
        assert "error" not in v[mode], v[mode]
    assert v["one_process_per_gpu"]["views"] == 6
    assert v["one_process_view_queue"]["views"] == 6
    assert any(f.startswith("smvs_hip_barrier_") for f in os.listdir(str(tmp_path)))


def test_reconstruct_scene_end_to_end(hip, oracle, oracle_threads, tmp_path):
    """an MVE scene directory (meta.ini, synth_0.out, .mvei
    images) through smvs_amd::reconstruct_scene -- smvsrecon's main between
    scene loading and mesh generation (app/smvsrecon.cc:400-745): view
    selection, one ViewQueue task per view (SGM front end + optimize), result
    embeddings written as .mvei.  The reference view's depth embedding is what
    the oracle's pipeline computes for the neighbours ViewSelection chose."""
    from smvs_amd import synth, host, mve_scene
    inputs = synth.pipeline_inputs("sphere", 384, 256, 3, flen=1.2)
    d = str(tmp_path)
    mve_scene.write_scene(d, inputs)
    done, skipped, secs = host.reconstruct_scene(d, view_ids=[0], num_neighbors=3,
                                                 min_neighbors=2, output_scale=2)
    assert done == [0] and skipped == 0 and secs > 0
    vdir = os.path.join(d, "views", "view_0000.mve")
    depth_mve = mve_scene.load_mvei(os.path.join(vdir, "smvs-B0.mvei"))
    normals = mve_scene.load_mvei(os.path.join(vdir, "smvs-B0N.mvei"))
    sgm = mve_scene.load_mvei(os.path.join(vdir, "smvs-sgm.mvei"))
    assert depth_mve.shape == (256, 384) and normals.shape == (256, 384, 3)
    assert sgm.shape == (128, 192)
    # the neighbours the scene run used: ViewSelection on the same scene
    scene = dict(views=[dict(id=i, flen=c.flen, rot=c.R, trans=c.t, width=384, height=256)
                        for i, c in enumerate(inputs["cams"])],
                 features=inputs["features"],
                 refs=[list(range(4))] * len(inputs["features"]))
    nb = host.select_neighbors(scene, 0, num_neighbors=3)
    assert len(nb) >= 2 and set(nb) <= {1, 2, 3}   # (the selection may reject a view)
    order = [0] + nb
    sel = dict(inputs, cams=[inputs["cams"][i] for i in order],
               images=[inputs["images"][i] for i in order], view_ids=order)
    sgm_o = oracle.sgm_depth_for_view(sel, sgm_scale=1, roundtrip=True)
    want = oracle.optimize(sel, regularization=0.01, num_iterations=5, min_scale=2,
                           sgm_depth=sgm_o)
    # the embedding is stored in MVE's ray-length convention (stereo_view.h:100-119)
    xs, ys = np.meshgrid(np.arange(384, dtype=np.float32) + np.float32(0.5),
                         np.arange(256, dtype=np.float32) + np.float32(0.5))
    f = np.float32(inputs["cams"][0].flen) * np.float32(384)
    vx = (xs - np.float32(192)) / f; vy = (ys - np.float32(128)) / f
    ray = np.sqrt(vx * vx + vy * vy + np.float32(1)).astype(np.float64)
    z = depth_mve.astype(np.float64) / ray
    assert np.array_equal(z > 0, want["depth"] > 0)
    assert _rel(z, want["depth"]) <= 1e-4
    # a second run finds the view reconstructed (smvsrecon.cc:544-555)
    done2, skipped2, _ = host.reconstruct_scene(d, view_ids=[0], num_neighbors=3,
                                                min_neighbors=2, output_scale=2)
    assert done2 == [] and skipped2 == 1


def test_reconstruct_scene_makescene_directory_with_automatic_input_scale(
        hip, oracle, oracle_threads, tmp_path):
    """SURVEY 8(f)-4, what round 3 refused: a view directory as makescene
    leaves it (undistorted.png) and smvsrecon's default automatic input scale
    (app/smvsrecon.cc:477-505): with --max-pixels below the images' size the
    run halves the inputs once with rescale_half_size_gaussian (:634-647),
    saves them as undist-L1.png, reads the views from that embedding and names
    its outputs smvs-B1.  Against the oracle's pipeline on the oracle's own
    half-size images ([MVE-unverified] M29)."""
    from smvs_amd import synth, host, mve_scene
    inputs = synth.pipeline_inputs("sphere", 768, 512, 3, flen=1.2)
    d = str(tmp_path)
    mve_scene.write_scene(d, inputs, container="png")
    done, skipped, secs, scale = host.reconstruct_scene(
        d, view_ids=[0], num_neighbors=3, min_neighbors=2, output_scale=2, input_scale=-1,
        max_pixels=150000, details=True)
    # 768 x 512 = 393,216 pixels: ceil(log2(393216 / 150000) / 2) = 1
    assert scale == 1 and done == [0] and skipped == 0
    vdir = os.path.join(d, "views", "view_0000.mve")
    half = host.load_byte_image(os.path.join(vdir, "undist-L1.png"))
    want_half = oracle.rescale_half_size_gaussian(np.asarray(inputs["images"][0], np.uint8))
    assert half.shape == (256, 384, 3) and np.array_equal(half, want_half)
    depth_mve = mve_scene.load_mvei(os.path.join(vdir, "smvs-B1.mvei"))
    assert depth_mve.shape == (256, 384)
    assert not os.path.exists(os.path.join(vdir, "smvs-B0.mvei"))
    # the oracle on the same half-size views (cameras are resolution independent)
    scene = dict(views=[dict(id=i, flen=c.flen, rot=c.R, trans=c.t, width=768, height=512)
                        for i, c in enumerate(inputs["cams"])],
                 features=inputs["features"],
                 refs=[list(range(4))] * len(inputs["features"]))
    nb = host.select_neighbors(scene, 0, num_neighbors=3)
    order = [0] + nb
    sel = dict(inputs, cams=[inputs["cams"][i] for i in order],
               images=[oracle.rescale_half_size_gaussian(np.asarray(inputs["images"][i], np.uint8))
                       for i in order], view_ids=order)
    sgm_o = oracle.sgm_depth_for_view(sel, sgm_scale=1, roundtrip=True)
    want = oracle.optimize(sel, regularization=0.01, num_iterations=5, min_scale=2,
                           sgm_depth=sgm_o)
    xs, ys = np.meshgrid(np.arange(384, dtype=np.float32) + np.float32(0.5),
                         np.arange(256, dtype=np.float32) + np.float32(0.5))
    f = np.float32(inputs["cams"][0].flen) * np.float32(384)
    vx = (xs - np.float32(192)) / f; vy = (ys - np.float32(128)) / f
    ray = np.sqrt(vx * vx + vy * vy + np.float32(1)).astype(np.float64)
    z = depth_mve.astype(np.float64) / ray
    assert np.array_equal(z > 0, want["depth"] > 0)
    assert _rel(z, want["depth"]) <= 1e-4
    # a second run reuses undist-L1 and finds the view reconstructed
    done2, skipped2, _, scale2 = host.reconstruct_scene(
        d, view_ids=[0], num_neighbors=3, min_neighbors=2, output_scale=2, input_scale=-1,
        max_pixels=150000, details=True)
    assert done2 == [] and skipped2 == 1 and scale2 == 1


def test_bench_contract_small(hip):
    """bench.py prints ONE JSON line with the fields the driver reads; run at
    the 480x270 debug size (the numbers mean nothing, the structure does).
    The headline is BASELINE.md's region: a step = one pass over all Newton
    loops of one optimize(), replayed from the recorded batch start states (the
    bench itself asserts that the replay reproduces the optimize()'s log)."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--small", "--steps", "3",
           "--warmup", "1", "--repeats", "2", "--no-cpu-baseline", "--no-peaks"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, res.stdout[-2000:]
    out = json.loads(lines[0])
    for key, want in (("unit", "active-patch-steps/s"), ("n_gpus", 1), ("steps", 3), ("warmup", 1),
                      ("higher_is_better", True), ("scaling", "weak"), ("vs_baseline", None),
                      ("dtype", "f64"), ("data", "synthetic")):
        assert out[key] == want, (key, out[key])
    assert out["metric"].startswith("Gauss-Newton iters/sec x active patches")
    assert out["value"] > 0 and out["ms_per_step"] > 0
    cfg = out["config"]
    assert "workload" in cfg and "model" not in cfg
    assert "all Newton loops of all scales" in cfg["workload"]
    assert cfg["replay_verified_against_optimize_log"] is True
    assert cfg["batches_per_step"] >= 9 and cfg["newton_steps_per_step"] >= cfg["batches_per_step"]
    roof = out["roofline"]
    for key in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "per_kernel",
                "region_share", "by_scale"):
        assert key in roof, key
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    assert roof["kernel"] == max(roof["region_share"], key=roof["region_share"].get)
    assert set(roof["per_kernel"]) >= {"patch", "cg_resident"}
    assert roof["per_kernel"]["patch"]["flop_per_patch_by_scale"]["2"] > 0
    # the headline and the in-place timers of secondary.optimize count the same units
    sec = out["secondary"]
    assert "error" not in sec, sec
    steps_units = sum(v["active_patch_steps"] for v in roof["by_scale"].values())
    assert steps_units == sec["optimize"]["active_patch_steps"]
    assert abs(out["value"] * out["ms_per_step"] * 1e-3 - steps_units) < 1e-6 * steps_units
    assert out["cpu_baseline"] is None                 # --no-cpu-baseline
    assert sec["optimize"]["value"] > 0 and sec["views_per_s"]["per_gpu"]["sgm_in_flight_8"]["views_per_s"] > 0
    assert sec["sgm_front_end"]["kernels"]["paths"]["frac"] > 0
    assert sec["scale2_replay"]["value"] > 0 and sec["scale2_replay"]["per_kernel"]["patch"]["frac"] > 0
    # the whole-optimize workload as its own line
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--small", "--workload",
                          "optimize", "--steps", "2", "--warmup", "1"], capture_output=True,
                         text=True, timeout=600, cwd=root)
    assert res.returncode == 0, res.stderr[-2000:]
    line = json.loads([l for l in res.stdout.splitlines() if l.strip()][-1])
    assert line["unit"] == "active-patch-steps/s" and line["value"] > 0 and line["steps"] == 2
    assert "optimize" in line["config"]["workload"]


def test_bench_contract_value_optimize_and_traffic_source(hip):
    """The honesty fields: `value_optimize` (the same region timed in place by
    the C++ loop timers) at the top level beside the headline, the committed
    source of `roofline.traffic`, HBM fractions also against the read peak
    measured on the box, and the CPU baseline on the same region."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--small", "--steps", "3",
           "--warmup", "1", "--repeats", "2"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root)
    assert res.returncode == 0, res.stderr[-2000:]
    out = json.loads([l for l in res.stdout.splitlines() if l.strip()][-1])
    assert out["value_optimize"] == out["secondary"]["optimize"]["value"] > 0
    assert "BASELINE.md" in out["value_optimize_note"]
    roof = out["roofline"]
    assert roof["traffic_source"].startswith("profiles/traffic_r")
    cg = roof["per_kernel"]["cg_resident"]
    assert cg["measured_read_peak_GBps"] > 1000
    assert abs(cg["frac_of_measured_read_peak"]
               - cg["achieved"] / cg["measured_read_peak_GBps"]) < 1e-3
    cpu = out["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["cores"] == 1 and cpu["value"] > 0
    assert "first batch of every scale" in cpu["sample"]
    assert cpu["all_cores"]["value"] > 0
    # both sides count the same batches: the oracle's first batches are the
    # device's (identical active patch-steps per batch is what the parity tests assert)
    assert sum(v["active_patch_steps"] for v in cpu["by_scale"].values()) == cpu["active_patch_steps"]


def test_bench_newton_steps_workload_is_still_there(hip):
    """`--workload newton_steps` (the headline of rounds 1-5, now
    secondary.scale2_replay): a step = one Newton step at scale 2."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--small", "--workload", "newton_steps",
           "--steps", "3", "--warmup", "1", "--repeats", "2", "--no-cpu-baseline", "--no-peaks",
           "--no-secondary"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root)
    assert res.returncode == 0, res.stderr[-2000:]
    out = json.loads([l for l in res.stdout.splitlines() if l.strip()][-1])
    assert out["steps"] == 3 and out["value"] > 0
    assert "scale 2" in out["config"]["workload"]
    assert "step_frac" in out["roofline"]


def test_bench_config5_runs_whole_views_in_child_processes(hip):
    """`bench.py --config 5` (configs[4] as one rank sees it): besides the
    headline, whole per-view tasks with -S -- SGM front end, optimize() of all
    scales with the SH lighting fit, maps -- through the ViewQueue, measured in
    fresh child processes both as one process per GPU and as ONE process
    driving every GPU (here: one GPU, so the two coincide in kind), plus the
    cost of a native RCCL lighting round."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--small", "--config", "5",
           "--views-per-rank", "2", "--shared-lighting", "--steps", "4", "--warmup", "1",
           "--repeats", "2", "--no-peaks", "--no-cpu-baseline"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, res.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["value"] > 0
    assert "configs[4]" in out["config"]["workload"]
    sec = out["secondary"]
    assert sec["rccl_lighting_round"]["us_per_round"] > 0
    v = sec["views_per_s"]
    assert v["shading"] is True and v["views_per_gpu"] == 2
    for mode in ("one_process_per_gpu", "one_process_view_queue"):
        assert "error" not in v[mode], v[mode]
        assert v[mode]["views"] == 2 and v[mode]["views_per_s"] > 0


_RCCL_WORLD2_PROBE = r"""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, sys.argv[1])
import smvs_amd
from smvs_amd import synth, shard
rank = int(os.environ["RANK"]); local = int(os.environ["LOCAL_RANK"])
torch.cuda.set_device(local)
dist.init_process_group("nccl", device_id=torch.device("cuda", local))
prob = synth.make_problem(224, 160, 3, 2, shading=True, noise=0.003)
rng = np.random.default_rng(100 + rank)
ctxs, mine = [], []
for v in range(2):
    s = dict(prob["surf"])
    nodes = prob["surf"]["nodes"].copy()
    nodes[:, 0] *= 1.0 + 0.003 * rng.standard_normal(nodes.shape[0])
    s["nodes"] = nodes
    c = smvs_amd.ViewContext(224, 160, 3, device=local)
    c.set_views(prob["views"]); c.set_surface(s)
    c.light_accumulate_dev()
    mine.append(np.concatenate([x.reshape(-1) for x in c.light_download()]))
    ctxs.append(c)
comm = shard.NativeComm(local, dist)
assert (comm.rank, comm.world) == (rank, dist.get_world_size())
comm.allreduce_lighting(ctxs)
box = [None] * dist.get_world_size()
dist.all_gather_object(box, mine)
want = sum(sum(m) for m in box)
for c in ctxs:
    got = np.concatenate([x.reshape(-1) for x in c.light_download()])
    err = np.linalg.norm(got - want) / np.linalg.norm(want)
    assert err < 1e-12, err
# and once more (the communicator is reusable, results deterministic)
for c in ctxs:
    c.light_accumulate_dev()
comm.allreduce_lighting(ctxs)
got2 = np.concatenate([x.reshape(-1) for x in ctxs[0].light_download()])
assert np.array_equal(got2, got)
comm.close()
for c in ctxs:
    c.close()
dist.barrier()
dist.destroy_process_group()
print("rank %d ok" % rank)
"""


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_native_rccl_lighting_allreduce_world_2(hip, tmp_path):
    """include/smvs_rccl.h across RANKS: two processes, one GPU each, two views
    per rank; smvs_light_allreduce (device sum over the rank's views ->
    ncclAllReduce over xGMI -> copy back) must leave the sum over all four
    views in every context.  Needs two GPUs: skipped on the one-GPU box, run by
    the driver's multi-GPU node."""
    import subprocess, sys
    if hip.device_count() < 2:
        pytest.skip("needs 2 GPUs (the driver's multi-GPU node)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    probe = tmp_path / "rccl_probe.py"
    probe.write_text(_RCCL_WORLD2_PROBE)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(probe), root]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, (res.stdout[-1500:], res.stderr[-3000:])
    assert "rank 0 ok" in res.stdout and "rank 1 ok" in res.stdout


def test_bench_two_gpus_reports_whole_views(hip):
    """`bench.py --gpus 2`: the headline as two ranks over RCCL plumbing, then
    whole views per second on both GPUs as two processes and as ONE process
    with ViewQueue(2, in_flight).  Skipped without two GPUs."""
    import json, subprocess, sys
    if hip.device_count() < 2:
        pytest.skip("needs 2 GPUs (the driver's multi-GPU node)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--small", "--steps",
           "4", "--warmup", "1", "--repeats", "2", "--views-per-rank", "3"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, cwd=root)
    assert res.returncode == 0, res.stderr[-3000:]
    out = json.loads([l for l in res.stdout.splitlines() if l.strip().startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["value"] > 0
    # what RCCL itself counted, and that the all-reduce summed both ranks
    assert out["n_ranks_seen_by_rccl"] == 2
    assert out["secondary"]["rccl"]["allreduce_summed_every_rank"] is True
    v = out["secondary"]["views_per_s"]
    for mode in ("one_process_per_gpu", "one_process_view_queue", "single_gpu_reference"):
        assert "error" not in v[mode], v[mode]
    assert v["one_process_per_gpu"]["views"] == 6
    assert v["one_process_view_queue"]["views"] == 6
