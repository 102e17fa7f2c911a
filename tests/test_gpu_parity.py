"""Parity of the HIP hot path (through the C ABI) against the CPU oracle on
the same seeded inputs.  FP64 path: tolerances are written at each assert;
integer / index outputs (active sets, iteration counts) must match exactly.
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


SOLVERS = ["auto", "resident_ref", "streaming"]


def _setup(hip, oracle, width, height, n_subs, scale, shading=False, noise=0.004,
           solver="auto"):
    from smvs_amd import synth
    prob = synth.make_problem(width, height, n_subs, scale, shading=shading,
                              noise=noise)
    ctx = hip.ViewContext(width, height, n_subs)
    ctx.set_solver(solver)
    ctx.set_views(prob["views"])
    ctx.set_surface(prob["surf"])
    orc = oracle.OracleProblem(prob["surf"], prob["views"])
    return prob, ctx, orc


@pytest.mark.parametrize("scale,size,n_subs", [(2, (192, 128), 3),
                                                (3, (256, 192), 4),
                                                (4, (320, 256), 2),
                                                (5, (512, 384), 8),
                                                (6, (704, 512), 5)])
def test_patch_systems_match_oracle(hip, oracle, scale, size, n_subs):
    """per-patch 16x16 J^T W J and 16-gradient (gauss_newton_step.cc:145-518);
    scales 2 / 3: four patches per wave, 4 / 5: one patch per wave, 6: 256
    samples per patch, four chunks on four waves of a workgroup."""
    prob, ctx, orc = _setup(hip, oracle, size[0], size[1], n_subs, scale)
    reg = 0.01
    ctx.gn_construct(reg)
    Hp, gp = ctx.gn_patch_systems()
    valid = np.flatnonzero(prob["surf"]["patch_valid"])
    rng = np.random.default_rng(0)
    pick = rng.choice(valid, size=min(40, valid.size), replace=False)
    worst_H = worst_g = 0.0
    for p in pick:
        g_ref, H_ref = orc.gn_patch(int(p), reg)
        H_gpu = np.triu(Hp[p])
        worst_H = max(worst_H, _rel(H_gpu, np.triu(H_ref)))
        worst_g = max(worst_g, _rel(gp[p], g_ref))
    # FP64, different (factored) summation order: 1e-10 relative
    assert worst_H < 1e-10, worst_H
    assert worst_g < 1e-10, worst_g
    ctx.close()


@pytest.mark.parametrize("shading,light_reg", [(False, 0.0), (True, 0.0), (True, 0.5)])
def test_assembled_system_matches_oracle(hip, oracle, shading, light_reg):
    """H (block stencil), g and the inverted diagonal blocks, incl. the
    shading term and the inactive-node rule (gauss_newton_step.cc:88-142)."""
    prob, ctx, orc = _setup(hip, oracle, 192, 128, 3, 2, shading=shading)
    lighting = prob["lighting"] if shading else None
    reg = 0.01
    surf = prob["surf"]
    rng = np.random.default_rng(1)
    active = surf["node_valid"].copy()
    active[rng.random(active.size) < 0.3] = 0  # partially active set
    ctx.set_active(active)
    n_gpu = ctx.gn_construct(reg, light_reg, lighting)
    H9, g, P = ctx.gn_download()
    ref = orc.gn_construct(active, reg, light_reg, lighting)
    assert n_gpu == ref["active_patches"]
    assert _rel(H9, ref["H9"]) < 1e-10
    assert _rel(g, ref["g"]) < 1e-10
    # blocks the reference does not hold are exactly zero
    assert np.all(H9[ref["present"] == 0] == 0.0)
    # inverted 4x4 blocks amplify by the block condition number
    assert _rel(P, ref["P"]) < 1e-7
    ctx.close()


@pytest.mark.parametrize("solver", SOLVERS)
@pytest.mark.parametrize("size,scale", [((256, 192), 2), ((128, 96), 4), ((96, 64), 3)])
def test_spmv_and_cg_match_oracle(hip, oracle, size, scale, solver):
    """ConjugateGradient::solve on identical systems: same iteration count,
    same return info, x within 1e-9 (conjugate_gradient.h:72-202); from a
    single-workgroup grid (77 nodes) to several thousand nodes; every solver
    implementation of smvs_ctx_set_solver."""
    prob, ctx, orc = _setup(hip, oracle, size[0], size[1], 4 if scale < 5 else 2, scale,
                            solver=solver)
    active = prob["surf"]["node_valid"]
    ref = orc.gn_construct(active, 0.01)
    # feed the ORACLE's system to the GPU solver so only the solver differs
    ctx.gn_upload(ref["H9"], ref["g"], ref["P"])
    tol = 0.01 * np.linalg.norm(ref["g"])
    for max_it, q_tol, etol in [(200, 1e-3, -1.0), (200, 1e-9, 1e-20), (6, 1e-9, 1e-30)]:
        it, info = ctx.cg_solve(max_it, etol, q_tol)
        x = ctx.cg_x()
        xr, itr, infor = orc.cg_solve(ref["H9"], ref["present"], ref["P"], -ref["g"],
                                      max_it, tol if etol < 0 else etol, q_tol)
        assert (it, info) == (itr, infor), (max_it, q_tol, it, info, itr, infor)
        assert _rel(x, xr) < 1e-9
    ctx.close()


def test_update_and_reactivate_matches_oracle(hip, oracle):
    """Surface::update_nodes + reprojection-based active set
    (depth_optimizer.cc:271-303): identical sets, identical node values."""
    prob, ctx, orc = _setup(hip, oracle, 256, 192, 4, 3, noise=0.02)
    active = prob["surf"]["node_valid"].copy()
    ref = orc.gn_construct(active, 0.01)
    x, _, _ = orc.cg_solve(ref["H9"], ref["present"], ref["P"], -ref["g"], 200,
                           0.01 * np.linalg.norm(ref["g"]), 1e-3)
    ctx.cg_set_x(x)
    n_gpu, _, nan = ctx.update_and_reactivate(0.15, False)
    new_active, n_ref, _ = orc.update_and_reactivate(x, active)
    a_gpu, cnt = ctx.get_active()
    assert nan == 0
    assert n_gpu == n_ref == cnt
    assert np.array_equal(a_gpu, new_active)
    assert 0 < n_ref < active.sum()  # the test actually exercises both outcomes
    assert np.max(np.abs(ctx.get_nodes() - orc.nodes)) == 0.0
    # full_optimization variant: mean reprojection delta
    ctx.set_active(active); orc2 = oracle.OracleProblem(prob["surf"], prob["views"])
    ctx.set_nodes(prob["surf"]["nodes"]); ctx.cg_set_x(x)
    _, mean_gpu, _ = ctx.update_and_reactivate(0.15, True)
    _, _, mean_ref = orc2.update_and_reactivate(x, active, full_optimization=True)
    assert abs(mean_gpu - mean_ref) < 1e-9 * max(1.0, mean_ref)
    ctx.close()


def test_nan_guard(hip, oracle):
    """delta[0] NaN leaves the surface untouched (depth_optimizer.cc:267-268)."""
    prob, ctx, orc = _setup(hip, oracle, 192, 128, 2, 2)
    x = np.zeros(4 * ctx.num_nodes); x[0] = np.nan
    before = ctx.get_nodes()
    ctx.cg_set_x(x)
    _, _, nan = ctx.update_and_reactivate(0.15, False)
    assert nan == 1
    assert np.array_equal(ctx.get_nodes(), before, equal_nan=True)
    ctx.close()


def test_depth_and_normal_maps_match_oracle(hip, oracle):
    """Surface::get_depth_map / get_normal_map (surface.cc:155-183)."""
    prob, ctx, orc = _setup(hip, oracle, 256, 192, 2, 3)
    d_gpu, d_ref = ctx.depth_map(), orc.depth_map()
    n_gpu, n_ref = ctx.normal_map(), orc.normal_map()
    assert np.array_equal(d_gpu == 0, d_ref == 0)
    assert _rel(d_gpu, d_ref) < 1e-6       # float32 outputs
    assert np.max(np.abs(n_gpu - n_ref)) < 1e-6
    ctx.close()


def test_maps_in_one_pass_pinned_buffers_and_mve_convention(hip, oracle):
    """smvs_get_maps: the two maps of smvs_get_depth_map / smvs_get_normal_map
    bit for bit, into pageable and into page-locked buffers (smvs_pinned_alloc);
    with the inverse calibration the depth comes back in MVE's ray-length
    convention -- StereoView::write_depth_to_view, stereo_view.h:100-119:
    `dm *= len` with len the float norm of the pixel's viewing ray widened to
    double (tests/golden/README.md M10) -- bit-identical with those float
    operations done in numpy."""
    prob, ctx, orc = _setup(hip, oracle, 352, 288, 3, 2, noise=0.01)
    d0, n0 = ctx.depth_map(), ctx.normal_map()
    for pinned in (False, True):
        d1, n1 = ctx.maps(pinned=pinned)
        assert np.array_equal(d1, d0) and np.array_equal(n1, n0)
    f = np.float32
    inv = _inverse_calibration(prob["main"].flen, 352, 288)
    xs, ys = np.meshgrid(np.arange(352, dtype=f) + f(0.5), np.arange(288, dtype=f) + f(0.5))
    v = [(inv[3 * r] * xs + inv[3 * r + 1] * ys) + inv[3 * r + 2] for r in range(3)]
    length = np.sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]).astype(f)
    want = (d0.astype(np.float64) * length.astype(np.float64)).astype(f)
    for pinned in (False, True):
        d2, n2 = ctx.maps(inv, pinned=pinned)
        assert np.array_equal(n2, n0)
        assert np.array_equal(d2, want)
    assert (want > d0).sum() > 0.2 * d0.size      # off-axis rays are longer
    ctx.close()


def test_light_fit_matches_oracle(hip, oracle):
    """LightOptimizer accumulation (light_optimizer.cc:32-49)."""
    prob, ctx, orc = _setup(hip, oracle, 256, 192, 2, 2, shading=True)
    A_gpu, b_gpu = ctx.light_accumulate()
    A_ref, b_ref = oracle.light_accumulate(orc.normal_map(), prob["views"]["shading"])
    assert _rel(A_gpu, A_ref) < 1e-10
    assert _rel(b_gpu, b_ref) < 1e-10
    ctx.close()


@pytest.mark.parametrize("solver", SOLVERS)
def test_gn_loop_tracks_oracle_loop(hip, oracle, solver):
    """The fused Newton loop (depth_optimizer.cc:219-304) against the same
    loop run step by step on the oracle: same step count, same active-set
    sizes, same CG iteration total, depth relative L2 <= 1e-4 (north_star
    tolerance) -- launch-ahead loop with either resident solver, and the
    one-step-at-a-time loop with the assembly kernel + streaming solver."""
    prob, ctx, orc = _setup(hip, oracle, 256, 192, 4, 2, noise=0.01, solver=solver)
    reg = 0.01
    stats = ctx.run_loop(reg, max_newton_steps=6)
    active = prob["surf"]["node_valid"].copy()
    n_init = int(active.sum()); n_act = n_init; steps = 0; patch_steps = 0; its = 0
    while steps < 6 and n_act > n_init // 20:
        steps += 1
        ref = orc.gn_construct(active, reg)
        patch_steps += ref["active_patches"]
        x, it, _ = orc.cg_solve(ref["H9"], ref["present"], ref["P"], -ref["g"], 200,
                                0.01 * np.linalg.norm(ref["g"]), 1e-3)
        its += it
        active, n_act, _ = orc.update_and_reactivate(x, active)
    assert stats["newton_steps"] == steps
    assert stats["active_patch_steps"] == patch_steps
    assert stats["final_active_nodes"] == n_act
    assert stats["linear_iterations"] == its
    d_gpu, d_ref = ctx.depth_map(), orc.depth_map()
    assert _rel(d_gpu, d_ref) <= 1e-4
    ctx.close()


# ---------------------------------------------------------------------- SGM
def _sgm_pair(w, h, seed):
    rng = np.random.default_rng(seed)
    base = rng.integers(20, 235, size=(h + 8, w + 40)).astype(np.float32)
    base = (base + np.roll(base, 1, 0) + np.roll(base, 1, 1) + np.roll(base, -1, 1)) / 4
    base[5:9, 20:30] = 0  # zero intensities exercise the census special case
    main = base[4:4 + h, 16:16 + w].astype(np.uint8)
    nbr = base[4:4 + h, 19:19 + w].astype(np.uint8)
    M = np.array([1.001, 0.002, 0.1, -0.001, 0.999, 0.2, 1e-6, -2e-6, 1.0], np.float32)
    t = np.array([-6.0, 0.3, 0.01], np.float32)
    return main, nbr, M, t


@pytest.mark.parametrize("w,h,D,p1,p2", [(96, 64, 128, 6, 96), (71, 45, 128, 6, 96),
                                          (64, 40, 37, 6, 96),       # odd plane count: one launch per direction
                                          (80, 56, 64, 10, 255),     # the largest penalty the byte form takes
                                          (80, 56, 64, 10, 300),     # above it: u16 volume, atomics
                                          (72, 48, 62, 6, 96),       # even, not a multiple of 4: atomics
                                          (150, 41, 20, 3, 40)])
def test_sgm_bit_exact(hip, oracle, w, h, D, p1, p2):
    """cost volume, aggregated volume, argmin and depth map are bit-exact
    with the oracle (sgm_stereo.cc:98-306), ragged sizes and odd plane counts
    included; every form of the aggregation (path bytes summed on the fly,
    u16 volume with atomics, one launch per direction)."""
    main, nbr, M, t = _sgm_pair(w, h, seed=w)
    out = hip.sgm_run(main, nbr, M, t, 1.0, 12.0, D, p1, p2, want_volumes=True)
    depths = oracle.sgm_depths(1.0, 12.0, D)
    cost = oracle.sgm_cost_volume(main, nbr, M, t, depths)
    assert np.array_equal(out["cost"], cost)
    sgm = oracle.sgm_aggregate(cost, p1, p2)
    assert np.array_equal(out["sgm"], sgm)
    depth, argmin = oracle.sgm_depth_from_volume(sgm, main, depths)
    assert np.array_equal(out["argmin"], argmin)
    assert np.array_equal(out["depth"], depth)
    assert (depth > 0).sum() > 0
    # the same call without the volumes (S is then never formed in memory)
    lean = hip.sgm_run(main, nbr, M, t, 1.0, 12.0, D, p1, p2)
    assert np.array_equal(lean["depth"], depth)


def test_sgm_rejects_bad_arguments(hip):
    from smvs_amd._capi import SmvsError
    main, nbr, M, t = _sgm_pair(64, 40, 1)
    with pytest.raises(SmvsError):
        hip.sgm_run(main, nbr, M, t, 1.0, 12.0, 300)     # too many planes
    with pytest.raises(SmvsError):
        hip.sgm_run(main, nbr, M, t, 5.0, 2.0, 64)       # inverted range
    with pytest.raises(SmvsError):
        hip.sgm_run(main, nbr, M, t, 1.0, 12.0, 64, p1=50, p2=10)


def test_bilateral_upsample_matches_oracle(hip, oracle):
    """depthmap_bilateral_filter (depth_optimizer.cc:957-1004); float path with
    device expf: 1e-5 relative."""
    rng = np.random.default_rng(9)
    dm = (2.0 + rng.random((24, 32))).astype(np.float32)
    dm[rng.random(dm.shape) < 0.2] = 0.0
    ci = rng.random((48, 64, 3)).astype(np.float32)
    got = hip.bilateral_upsample(dm, ci)
    want = oracle.bilateral_upsample(dm, ci)
    assert np.array_equal(got == 0, want == 0)
    assert np.max(np.abs(got - want)) <= 1e-5 * np.max(np.abs(want))


# ------------------------------------------------- whole optimizer (host C++)
from parity_units import assert_same_units, control_flow  # noqa: E402


def _same_control_flow(a, b):
    return control_flow(a) == control_flow(b)


def test_host_optimize_matches_oracle_config1(hip, oracle):
    """configs[0]: 640x480 planar scene, 1 + 2 views, -o2 --no-sgm, through smvs_amd::DepthOptimizer
    (C++ host + HIP kernels) against the oracle's optimize(): identical scale /
    iteration / patch-count trace, depth relative L2 <= 1e-4."""
    from smvs_amd import synth, host
    inputs = synth.pipeline_inputs("plane", 640, 480, 2)   # configs[0]'s size
    got = host.optimize(inputs, regularization=0.01, num_iterations=5, min_scale=2)
    oracle.lib().orc_set_threads(max(1, min(os.cpu_count() or 1, 64)))
    try:
        want = oracle.optimize(inputs, regularization=0.01, num_iterations=5, min_scale=2)
    finally:
        oracle.lib().orc_set_threads(1)
    assert_same_units(got["log"], want["log"], 640, 480, "configs0_plane_640x480")
    assert np.array_equal(got["depth"] > 0, want["depth"] > 0)
    print("config1 depth rel L2 %.3e, normals rel L2 %.3e max %.3e"
          % (_rel(got["depth"], want["depth"]), _rel(got["normals"], want["normals"]),
             np.max(np.abs(got["normals"] - want["normals"]))))
    assert _rel(got["depth"], want["depth"]) <= 1e-4
    # normals are derivatives of the surface: 1e-3 relative L2
    assert _rel(got["normals"], want["normals"]) <= 1e-3


def test_host_optimize_with_sgm_and_shading_matches_oracle(hip, oracle):
    """configs[2]/[3]-like: SGM initialisation (run_sgm x 4, L/R check and
    merge on the device) feeding the optimizer with the shading term on; each
    side runs its own SGM front end."""
    from smvs_amd import synth, host
    rng = np.random.default_rng(3000)
    lighting = np.zeros(16); lighting[0] = 0.9
    lighting[1:4] = rng.uniform(-0.2, 0.2, 3)
    inputs = synth.pipeline_inputs("sphere", 384, 256, 3, flen=1.2,
                                   lighting=lighting)
    sgm = host.sgm_depth(inputs, sgm_scale=1)
    assert (sgm > 0).mean() > 0.3
    got = host.optimize(inputs, regularization=0.01, num_iterations=3, min_scale=2,
                        use_shading=True, sgm_depth=sgm)
    # the oracle runs its OWN SGM front end (run_sgm x 4, L/R check, merge,
    # write_depth_to_view / get_sgm_depth round trip) on the same images
    sgm_o = oracle.sgm_depth_for_view(inputs, sgm_scale=1, roundtrip=True)
    assert np.array_equal(got["sgm_roundtrip"], sgm_o)
    want = oracle.optimize(inputs, regularization=0.01, num_iterations=3,
                           min_scale=2, use_shading=True, sgm_depth=sgm_o)
    assert_same_units(got["log"], want["log"], 384, 256, "sgm_shading_384x256")
    assert got["lighting"] is not None and want["lighting"] is not None
    print("lighting rel %.3e" % _rel(got["lighting"], want["lighting"]))
    # the 16x16 SH normal matrix is ill-conditioned: 1e-3 on the coefficients
    assert _rel(got["lighting"], want["lighting"]) < 1e-3
    both = (got["depth"] > 0) & (want["depth"] > 0)
    assert both.mean() > 0.3
    print("sgm+shading depth rel L2 %.3e" % _rel(got["depth"], want["depth"]))
    assert np.array_equal(got["depth"] > 0, want["depth"] > 0)
    assert _rel(got["depth"], want["depth"]) <= 1e-4


def test_empty_active_set_and_invalid_surface(hip, oracle):
    """No active node: nothing is evaluated, H, g and x are exactly zero and
    the loop does not start (depth_optimizer.cc:219-220)."""
    prob, ctx, orc = _setup(hip, oracle, 192, 128, 2, 2)
    ctx.set_active(np.zeros(ctx.num_nodes, np.uint8))
    assert ctx.gn_construct(0.01) == 0
    H9, g, P = ctx.gn_download()
    assert not H9.any() and not g.any() and not P.any()
    stats = ctx.run_loop(0.01, reset_active=False)
    assert stats["newton_steps"] == 0 and stats["active_patch_steps"] == 0
    # a surface without any valid patch
    surf = dict(prob["surf"])
    surf["patch_valid"] = np.zeros_like(surf["patch_valid"])
    surf["node_valid"] = np.zeros_like(surf["node_valid"])
    ctx.set_surface(surf)
    assert ctx.gn_construct(0.01) == 0
    assert not ctx.depth_map().any()
    ctx.close()


def test_call_order_errors(hip):
    """Call-order violations surface as status codes, not crashes."""
    from smvs_amd._capi import SmvsError
    ctx = hip.ViewContext(64, 48, 2)
    with pytest.raises(SmvsError):
        ctx.gn_construct(0.01)            # no cameras / surface
    with pytest.raises(SmvsError):
        ctx.cg_solve()                    # no system
    with pytest.raises(SmvsError):
        hip.ViewContext(64, 48, 17)       # more than SMVS_MAX_SUBS
    ctx.close()


def test_loop_argument_errors(hip, oracle):
    """smvs_gn_run_loop rejects what its kernels cannot carry (the iteration
    number travels in 16 bits of the solvers' tags) and a loop of zero steps
    only reports the active set."""
    from smvs_amd._capi import SmvsError
    prob, ctx, _ = _setup(hip, oracle, 128, 96, 2, 2)
    with pytest.raises(SmvsError):
        ctx.run_loop(0.01, cg_max_iterations=70000)
    with pytest.raises(SmvsError):
        ctx.run_loop(0.01, max_newton_steps=-1)
    before = ctx.get_nodes()
    stats = ctx.run_loop(0.01, max_newton_steps=0)
    assert stats["newton_steps"] == 0
    assert stats["final_active_nodes"] == int(prob["surf"]["node_valid"].sum())
    assert np.array_equal(ctx.get_nodes(), before)
    # one CG iteration allowed: conjugate_gradient.h:121-123 returns x = 0
    stats = ctx.run_loop(0.01, max_newton_steps=2, cg_max_iterations=1)
    assert stats["newton_steps"] >= 1 and np.array_equal(ctx.get_nodes(), before)
    ctx.close()


def test_device_bicubic_matches_reference_known_answers(hip):
    """The reference's own known-answer vectors for BicubicPatch
    (tests/gtest_bicubic_patch.cc:16-162, tests/golden/) through the device
    surface evaluation: a 1 x 1 patch grid at scale 0 samples the patch at
    (0.5, 0.5)."""
    import json, os
    with open(os.path.join(os.path.dirname(__file__), "golden",
                           "reference_known_answers.json")) as f:
        known = json.load(f)
    ctx = hip.ViewContext(16, 12, 1)
    views = dict(M=np.eye(3).reshape(1, 9), t=np.zeros((1, 3)), flen=10.0,
                 inv_flen=0.1, grad=np.zeros((12, 16, 2), np.float32),
                 subs=[(np.zeros((12, 16, 2), np.float32),
                        np.zeros((12, 16, 3), np.float32))])
    ctx.set_views(views)
    checked = 0
    for case in known["bicubic"]:
        want = [c[3] for c in case["checks"] if c[0] == "f" and c[1] == 0.5 and c[2] == 0.5]
        if not want:
            continue
        surf = dict(scale=0, npx=1, npy=1, start_x=5, start_y=4,
                    nodes=np.array(case["nodes"], dtype=float),
                    node_valid=np.ones(4, np.uint8), patch_valid=np.ones(1, np.uint8),
                    patch_vis=np.ones(1, np.uint32))
        ctx.set_surface(surf)
        depth = ctx.depth_map()
        assert depth[4, 5] == np.float32(want[0]), case["source"]
        assert np.count_nonzero(depth) == 1
        checked += 1
    assert checked == 4
    ctx.close()


@pytest.mark.parametrize("scale", [2, 4, 6])
def test_device_scale_space_is_bit_exact(hip, oracle, scale):
    """Byte -> float, Gaussian blur, luminance and the 3x3 quadratic-fit
    gradient / Hessian on the device (stereo_view.cc:16-46, 97-188) are
    bit-identical to the host restatement, RGB and grey, ragged sizes."""
    rng = np.random.default_rng(scale)
    main = rng.integers(0, 256, size=(61, 83, 3)).astype(np.uint8)
    sub0 = rng.integers(0, 256, size=(61, 83, 3)).astype(np.uint8)
    sub1 = rng.integers(0, 256, size=(47, 70)).astype(np.uint8)   # grey, other size
    ctx = hip.ViewContext(83, 61, 2)
    ctx.upload_image(-1, main); ctx.upload_image(0, sub0); ctx.upload_image(1, sub1)
    ctx.set_scale(scale)
    for view, img in ((-1, main), (0, sub0), (1, sub1)):
        g_ref, h_ref = oracle.scale_planes(img, scale)
        g, h = ctx.download_planes(view)
        assert np.array_equal(g, g_ref), (view, np.abs(g - g_ref).max())
        if view >= 0:
            assert np.array_equal(h, h_ref), (view, np.abs(h - h_ref).max())
    ctx.close()


@pytest.mark.parametrize("scale", [1, 3, 5])
def test_device_scale_space_is_bit_exact_across_tiles(hip, oracle, scale):
    """The same planes on images wider and taller than one workgroup's tile
    of the fused y pass / luminance / fit kernel (254 x 8 pixels, csrc/scale.hip):
    partial last tiles, grey and RGB, in both forms of the pass."""
    import os
    rng = np.random.default_rng(40 + scale)
    main = rng.integers(0, 256, size=(301, 521, 3)).astype(np.uint8)
    sub0 = rng.integers(0, 256, size=(259, 509, 3)).astype(np.uint8)
    sub1 = rng.integers(0, 256, size=(263, 771)).astype(np.uint8)
    want = [oracle.scale_planes(img, scale) for img in (main, sub0, sub1)]
    old = os.environ.get("SMVS_SCALE_FUSED")
    try:
        for fused in ("1", "0"):
            os.environ["SMVS_SCALE_FUSED"] = fused
            ctx = hip.ViewContext(521, 301, 2)
            ctx.upload_image(-1, main); ctx.upload_image(0, sub0); ctx.upload_image(1, sub1)
            ctx.set_scale(scale)
            for view, (g_ref, h_ref) in zip((-1, 0, 1), want):
                g, h = ctx.download_planes(view)
                assert np.array_equal(g, g_ref), (fused, view, np.abs(g - g_ref).max())
                if view >= 0:
                    assert np.array_equal(h, h_ref), (fused, view, np.abs(h - h_ref).max())
            ctx.close()
    finally:
        if old is None:
            os.environ.pop("SMVS_SCALE_FUSED", None)
        else:
            os.environ["SMVS_SCALE_FUSED"] = old


def test_fused_scale_space_equals_the_separate_kernels_at_full_size(hip):
    """1920 x 1080 RGB and a ragged grey neighbour, every scale of optimize()
    (6 .. 2): the fused pass leaves the planes of blur_y_kernel +
    gradients_kernel to the bit (those are the oracle's, tests above)."""
    import os
    rng = np.random.default_rng(77)
    main = rng.integers(0, 256, size=(1080, 1920, 3)).astype(np.uint8)
    sub0 = rng.integers(0, 256, size=(1080, 1920, 3)).astype(np.uint8)
    sub1 = rng.integers(0, 256, size=(997, 1501)).astype(np.uint8)
    old = os.environ.get("SMVS_SCALE_FUSED")
    ctx = hip.ViewContext(1920, 1080, 2)
    ctx.upload_image(-1, main); ctx.upload_image(0, sub0); ctx.upload_image(1, sub1)
    try:
        for scale in (6, 5, 4, 3, 2):
            planes = {}
            for fused in ("0", "1"):
                os.environ["SMVS_SCALE_FUSED"] = fused
                ctx.set_scale(scale)
                planes[fused] = [ctx.download_planes(v) for v in (-1, 0, 1)]
            for v in range(3):
                assert np.array_equal(planes["0"][v][0], planes["1"][v][0]), (scale, v)
                assert np.count_nonzero(planes["1"][v][0]) > 0
                if v > 0:
                    assert np.array_equal(planes["0"][v][1], planes["1"][v][1]), (scale, v)
    finally:
        ctx.close()
        if old is None:
            os.environ.pop("SMVS_SCALE_FUSED", None)
        else:
            os.environ["SMVS_SCALE_FUSED"] = old


def test_cloned_loop_state_runs_the_same_loop(hip):
    """smvs_ctx_clone_loop_state (what bench.py's timed region replays): the
    clone's Newton loop -- and its rerun after smvs_ctx_restore_nodes -- is the
    original's to the bit, with and without the shading term; the original is
    untouched by the clone's runs."""
    from smvs_amd import synth
    for shading in (False, True):
        prob = synth.make_problem(256, 192, 3, 2, noise=0.003, shading=shading)
        lighting = None
        if shading:
            lighting = np.zeros(16); lighting[0] = 0.9; lighting[2] = 0.1
        ctx = hip.ViewContext(256, 192, 3)
        ctx.set_views(prob["views"]); ctx.set_surface(prob["surf"])
        clone = ctx.clone_loop_state()
        start = ctx.get_nodes()
        a = clone.run_loop(0.01, lighting=lighting)
        nodes_a = clone.get_nodes()
        assert a["newton_steps"] >= 1 and not np.array_equal(nodes_a, start)
        assert np.array_equal(ctx.get_nodes(), start)        # the original did not move
        clone.restore_nodes()
        assert np.array_equal(clone.get_nodes(), start)
        b = clone.run_loop(0.01, lighting=lighting)
        assert a == b and np.array_equal(clone.get_nodes(), nodes_a)
        c = ctx.run_loop(0.01, lighting=lighting)
        assert c == a and np.array_equal(ctx.get_nodes(), nodes_a)
        clone.close(); ctx.close()


def test_async_image_upload_gives_the_same_planes(hip, oracle):
    """smvs_ctx_upload_image_async (page-locked source, DMA on the context's copy
    stream, conversion where the image is first needed) against the
    synchronous upload and the oracle: images of more than a megabyte (smaller
    ones take the synchronous path by design), one of them replaced by a second
    upload before it was ever read, set_scale twice (the second finds nothing
    pending), planes bit-identical."""
    rng = np.random.default_rng(11)
    w, h = 800, 600
    imgs = [rng.integers(0, 256, size=(h, w, 3)).astype(np.uint8) for _ in range(3)]
    stale = rng.integers(0, 256, size=(h, w, 3)).astype(np.uint8)
    ctx = hip.ViewContext(w, h, 2)
    ctx.upload_image_async(-1, imgs[0])
    ctx.upload_image_async(0, stale)        # never read: replaced below
    ctx.upload_image_async(0, imgs[1])
    ctx.upload_image_async(1, imgs[2])
    for scale in (3, 2):
        ctx.set_scale(scale)
        for view, img in ((-1, imgs[0]), (0, imgs[1]), (1, imgs[2])):
            g_ref, h_ref = oracle.scale_planes(img, scale)
            g, hs = ctx.download_planes(view)
            assert np.array_equal(g, g_ref), (scale, view)
            if view >= 0:
                assert np.array_equal(hs, h_ref), (scale, view)
    # a synchronous upload over an asynchronous one
    ctx.upload_image(1, imgs[0])
    ctx.set_scale(2)
    g, hs = ctx.download_planes(1)
    g_ref, h_ref = oracle.scale_planes(imgs[0], 2)
    assert np.array_equal(g, g_ref) and np.array_equal(hs, h_ref)
    ctx.close()


# ---------------------------------------------------------------------------
# BASELINE.json's full sizes (configs[1] / configs[2])
# ---------------------------------------------------------------------------
@pytest.fixture(scope="module")
def full_size_problem(hip, oracle):
    """The bench workload: 1920x1080, 8 neighbours, scale 2."""
    import bench
    prob = bench.make_problem(0, False)
    surf = prob["surf"]
    ctx = hip.ViewContext(surf["width"], surf["height"], bench.NSUBS)
    ctx.set_views(prob["views"])
    ctx.set_surface(surf)
    orc = oracle.OracleProblem(surf, prob["views"])
    yield prob, ctx, orc, bench.REG
    ctx.close()


def test_full_size_newton_step_matches_oracle(hip, oracle, full_size_problem):
    """One Newton step of the bench workload (1920x1080, 8 neighbours,
    130k nodes): sampled per-patch systems against the oracle's patch
    routine, then the oracle's PCG and active-set update on the GPU-built
    system against the GPU's."""
    prob, ctx, orc, reg = full_size_problem
    surf = prob["surf"]
    ctx.set_nodes(surf["nodes"]); ctx.set_active(None)
    n_active_patches = ctx.gn_construct(reg)
    assert n_active_patches == int(surf["patch_valid"].sum())
    Hp, gp = ctx.gn_patch_systems()
    valid = np.flatnonzero(surf["patch_valid"])
    pick = np.random.default_rng(5).choice(valid, size=64, replace=False)
    for p in pick:
        g_ref, H_ref = orc.gn_patch(int(p), reg)
        # FP64, factored summation order over 72 residual rows: 1e-9 relative
        assert _rel(np.triu(Hp[p]), np.triu(H_ref)) < 1e-9
        assert _rel(gp[p], g_ref) < 1e-9
    del Hp, gp
    H9, g, P = ctx.gn_download()
    # size-independent structure: H is symmetric block for block
    N = H9.shape[0]; stride = surf["npx"] + 1
    n = np.arange(N - stride - 1)
    for s, off in [(5, 1), (7, stride), (8, stride + 1)]:
        a = H9[n, s].reshape(-1, 4, 4)
        b = H9[n + off, 8 - s].reshape(-1, 4, 4).transpose(0, 2, 1)
        assert np.array_equal(a, b)
    # the solve, on the same (GPU-built) system
    it, info = ctx.cg_solve(200, -1.0, 1e-3)
    x = ctx.cg_x()
    present = (np.abs(H9).sum(axis=2) > 0).astype(np.uint8)
    xr, itr, infor = orc.cg_solve(H9, present, P, -g, 200, 0.01 * np.linalg.norm(g), 1e-3)
    assert (it, info) == (itr, infor)
    assert _rel(x, xr) < 1e-9
    # residual actually went down: |Hx + g| < |g| (energy property of PCG)
    r = orc.spmv(H9, present, x) + g
    assert np.linalg.norm(r) < 0.5 * np.linalg.norm(g)
    # node update + re-activation
    active = surf["node_valid"].copy()
    n_gpu, _, nan = ctx.update_and_reactivate(0.15, False)
    new_active, n_ref, _ = orc.update_and_reactivate(xr, active)
    a_gpu, cnt = ctx.get_active()
    assert nan == 0 and n_gpu == n_ref == cnt
    assert np.array_equal(a_gpu, new_active)
    assert np.max(np.abs(ctx.get_nodes() - orc.nodes)) < 1e-12


_FULL_SIZE_ORACLE = {}


@pytest.mark.parametrize("solver", SOLVERS)
def test_full_size_loop_matches_oracle_loop(hip, oracle, full_size_problem, solver):
    """The bench workload as a whole Newton batch (1920x1080, 8 neighbours,
    scale 2, 128,851 nodes, 256 tiles of the resident solver; launch-ahead,
    fused assembly) against the oracle's loop run step by step on all host
    cores (depth_optimizer.cc:219-304): the number of Newton steps, the active
    patches of every step (their sum), the CG iteration total and the final
    active set are the oracle's, the nodes agree and the depth map is within
    the north-star 1e-4 relative L2 (measured: 1e-8).  The run is
    deterministic (two runs, bit-identical nodes), and the batch moves the
    surface towards the analytic sphere."""
    from smvs_amd import synth
    prob, ctx, orc, reg = full_size_problem
    surf = prob["surf"]
    if "loop" not in _FULL_SIZE_ORACLE:
        oracle.lib().orc_set_threads(max(1, min(os.cpu_count() or 1, 64)))
        try:
            orc.nodes[:] = np.asarray(surf["nodes"], dtype=np.float64).reshape(-1, 4)
            act = surf["node_valid"].copy()
            n_init = int(act.sum()); n_act = n_init
            steps = its = psteps = 0
            while steps < 200 and n_act > n_init // 20:
                steps += 1
                ref = orc.gn_construct(act, reg)
                psteps += ref["active_patches"]
                xr, itr, _ = orc.cg_solve(ref["H9"], ref["present"], ref["P"], -ref["g"],
                                          200, 0.01 * np.linalg.norm(ref["g"]), 1e-3)
                its += itr
                act, n_act, _ = orc.update_and_reactivate(xr, act)
                del ref
        finally:
            oracle.lib().orc_set_threads(1)
        _FULL_SIZE_ORACLE["loop"] = (steps, psteps, its, n_act, orc.nodes.copy(),
                                     orc.depth_map())
    steps, psteps, its, n_act, want_nodes, want_depth = _FULL_SIZE_ORACLE["loop"]

    ctx.set_solver(solver)
    xs, ys = np.meshgrid(np.arange(surf["width"], dtype=float),
                         np.arange(surf["height"], dtype=float))
    gt = synth.depth_at(prob["scene"], prob["main"], xs, ys)
    ctx.set_nodes(surf["nodes"]); ctx.set_active(None)
    d0 = ctx.depth_map()

    def run():
        ctx.set_nodes(surf["nodes"])
        stats = ctx.run_loop(reg)
        return stats, ctx.get_nodes(), ctx.depth_map()

    s1, n1, d1 = run()
    s2, n2, d2 = run()
    ctx.set_solver("auto")
    print("full-size loop [%s]: steps %d, active patch-steps %d, CG iterations oracle %d "
          "device %d, depth %.2e" % (solver, steps, psteps, its, s1["linear_iterations"],
                                     _rel(d1, want_depth)))
    assert s1 == s2 and np.array_equal(n1, n2)
    assert steps >= 2 and s1["nan_break"] == 0
    assert s1["newton_steps"] == steps
    assert s1["active_patch_steps"] == psteps
    assert s1["linear_iterations"] == its
    assert s1["final_active_nodes"] == n_act
    assert np.array_equal(d1 > 0, want_depth > 0)
    assert _rel(d1, want_depth) <= 1e-4
    assert np.max(np.abs(n1 - want_nodes)) <= 1e-6 * np.max(np.abs(want_nodes))
    mask = d0 > 0
    e0 = np.sqrt(np.mean((d0[mask] - gt[mask]) ** 2))
    e1 = np.sqrt(np.mean((d1[mask] - gt[mask]) ** 2))
    assert e1 < 0.7 * e0, (e0, e1)


def test_full_size_sgm_bit_exact(hip, oracle):
    """configs[2]: 960x540 (sgm_scale 1 of 1920x1080), 128 planes, 8 paths."""
    main, nbr, M, t = _sgm_pair(960, 540, seed=11)
    out = hip.sgm_run(main, nbr, M, t, 1.0, 12.0, 128, 6, 96, want_volumes=True)
    depths = oracle.sgm_depths(1.0, 12.0, 128)
    cost = oracle.sgm_cost_volume(main, nbr, M, t, depths)
    assert np.array_equal(out["cost"], cost)
    sgm = oracle.sgm_aggregate(cost, 6, 96)
    assert np.array_equal(out["sgm"], sgm)
    depth, argmin = oracle.sgm_depth_from_volume(sgm, main, depths)
    assert np.array_equal(out["argmin"], argmin)
    assert np.array_equal(out["depth"], depth)


# ---------------------------------------------------------------------------
# topology tests between Newton batches
# ---------------------------------------------------------------------------
def _topology_setup(hip, oracle, width, height, n_subs, scale, noise):
    from smvs_amd import synth
    prob = synth.make_problem(width, height, n_subs, scale, noise=noise)
    surf = dict(prob["surf"])
    # every patch of the grid takes part: silhouette-straddling and
    # background patches exercise the occlusion / border / NCC branches
    surf["patch_valid"] = np.ones_like(surf["patch_valid"])
    surf["node_valid"] = np.ones_like(surf["node_valid"])
    surf["patch_vis"] = np.zeros_like(surf["patch_vis"])
    ctx = hip.ViewContext(width, height, n_subs)
    ctx.set_views(prob["views"])                  # cameras (planes are replaced below)
    for v, img in enumerate(prob["images"]):
        ctx.upload_image(v - 1, img)
    ctx.set_scale(scale)                          # device scale space
    grads = [ctx.download_planes(v - 1)[0] for v in range(n_subs + 1)]
    images = [img.astype(np.float32) / np.float32(255.0) for img in prob["images"]]
    tp = oracle.TopologyProblem(surf, images, grads, prob["views"]["M"],
                                prob["views"]["t"], prob["main"].flen)
    return prob, surf, ctx, tp


def _inverse_calibration(flen, w, h):
    f = np.float32
    ax = f(flen) * f(max(w, h))
    return np.array([f(1) / ax, 0, -f(w) * f(0.5) / ax, 0, f(1) / ax,
                     -f(h) * f(0.5) / ax, 0, 0, 1], dtype=np.float32)


@pytest.mark.parametrize("size,n_subs,scale", [((320, 256), 4, 2), ((384, 256), 3, 3),
                                               ((512, 384), 2, 4), ((160, 128), 2, 1),
                                               ((640, 512), 3, 6), ((576, 416), 3, 5)])
def test_topology_subviews_mse_and_cuts_match_oracle(hip, oracle, size, n_subs, scale):
    """create_subview_surfaces, mse_for_patch and the cut_boundaries loop
    (depth_optimizer.cc:360-604, 747-912) on the device against the oracle:
    identical visibility masks and validity, MSE to 1e-10."""
    prob, surf, ctx, tp = _topology_setup(hip, oracle, size[0], size[1], n_subs, scale, 0.01)
    ctx.set_surface(surf)
    vis_gpu = ctx.topology_subviews(None, use_ncc=True)
    vis_ref = tp.subviews()
    assert np.array_equal(vis_gpu, vis_ref)
    # the test is not vacuous: visible, partly visible and invisible patches
    assert (vis_ref == 0).any() and (vis_ref == (1 << n_subs) - 1).any()
    assert ((vis_ref != 0) & (vis_ref != (1 << n_subs) - 1)).any()

    # continue from the oracle's surface (invisible patches deleted)
    surf2 = dict(surf)
    surf2["patch_valid"] = tp.patch_valid.copy()
    surf2["node_valid"] = tp.node_valid.copy()
    surf2["patch_vis"] = tp.patch_vis.copy()
    ctx.set_surface(surf2)
    mse_gpu = ctx.topology_patch_mse()
    mse_ref = tp.patch_mse()
    valid = surf2["patch_valid"] != 0
    assert np.all(mse_gpu[~valid] == -1.0)
    assert np.max(np.abs(mse_gpu[valid] - mse_ref[valid]) / np.maximum(mse_ref[valid], 1e-30)) < 1e-10

    pv, nv, deleted = ctx.topology_cut_boundaries(
        _inverse_calibration(prob["main"].flen, size[0], size[1]))
    deleted_ref = tp.cut_boundaries()
    assert deleted == deleted_ref
    if scale <= 3:
        assert deleted_ref > 0   # the passes actually delete something
    assert np.array_equal(pv, tp.patch_valid)
    assert np.array_equal(nv, tp.node_valid)
    ctx.close()


@pytest.mark.parametrize("size,n_subs,scale", [((320, 256), 4, 2), ((384, 256), 3, 3)])
def test_cut_boundaries_in_three_launches_and_in_five(hip, oracle, monkeypatch, size, n_subs, scale):
    """Round 6 fused the two ends of a cut_boundaries pass by recomputation
    (topo_border_candidates_kernel, topo_cut_fused_kernel; the pass's counters
    alternate between two word pairs): same deletions, patch and node validity
    as the five launches (SMVS_CUT_FUSED=0) and as the oracle, also when calls
    of both forms follow each other on one context."""
    prob, surf, ctx, tp = _topology_setup(hip, oracle, size[0], size[1], n_subs, scale, 0.01)
    tp.subviews()
    surf2 = dict(surf)
    surf2["patch_valid"] = tp.patch_valid.copy()
    surf2["node_valid"] = tp.node_valid.copy()
    surf2["patch_vis"] = tp.patch_vis.copy()
    inv = _inverse_calibration(prob["main"].flen, size[0], size[1])
    deleted_ref = tp.cut_boundaries()
    assert deleted_ref > 0
    for form in ("1", "0", "1", "1", "0"):
        monkeypatch.setenv("SMVS_CUT_FUSED", form)
        ctx.set_surface(surf2)
        pv, nv, deleted = ctx.topology_cut_boundaries(inv)
        assert deleted == deleted_ref, form
        assert np.array_equal(pv, tp.patch_valid), form
        assert np.array_equal(nv, tp.node_valid), form
    monkeypatch.delenv("SMVS_CUT_FUSED", raising=False)
    ctx.close()


@pytest.mark.parametrize("size,n_subs,scale", [((320, 256), 6, 2), ((576, 416), 3, 5)])
def test_topology_shared_reciprocals_give_the_bits_of_the_divisions(hip, oracle, monkeypatch,
                                                                   size, n_subs, scale):
    """The visibility and MSE kernels take the ten quotients of a warp from one
    refined reciprocal of d and one of d * d (csrc/topology.hip, SharedDivisor):
    the instruction sequence of the division without its scaling steps.  With
    SMVS_TOPO_DIVIDE=exact every quotient is the division itself -- masks and
    errors must be the same bits (and the oracle's, the tests above).  So must
    they be with the NCC samples of a lane taken one at a time instead of in
    pairs and the neighbours of the MSE kernel one by one instead of four at
    once: those only change when loads are issued.  And with the z-buffer as the
    reference keeps it (3 x 3 splats, nine lookups per pixel) instead of its
    5 x 5 minimum filter and one lookup (round 6, topo_dilate5_kernel)."""
    prob, surf, ctx, tp = _topology_setup(hip, oracle, size[0], size[1], n_subs, scale, 0.01)
    tp.subviews()
    surf2 = dict(surf)
    surf2["patch_valid"] = tp.patch_valid.copy()
    surf2["node_valid"] = tp.node_valid.copy()
    surf2["patch_vis"] = tp.patch_vis.copy()
    got = {}
    for mode in ("shared", "exact", "one_sample_at_a_time", "one_neighbour_at_a_time",
                 "z_buffer_window_3"):
        for name in ("SMVS_TOPO_DIVIDE", "SMVS_NCC_PAIRS", "SMVS_MSE_SUBS", "SMVS_ZBUF_WINDOW"):
            monkeypatch.delenv(name, raising=False)
        if mode == "z_buffer_window_3":     # the 3 x 3 z-buffer and nine lookups per pixel
            monkeypatch.setenv("SMVS_ZBUF_WINDOW", "3")
        elif mode == "exact":
            monkeypatch.setenv("SMVS_TOPO_DIVIDE", "exact")
        elif mode == "one_sample_at_a_time":       # the NCC samples of a lane, not in pairs
            monkeypatch.setenv("SMVS_NCC_PAIRS", "0")
        elif mode == "one_neighbour_at_a_time":    # the MSE kernel's neighbours, not four at once
            monkeypatch.setenv("SMVS_MSE_SUBS", "1")
        ctx.set_surface(surf)
        vis = ctx.topology_subviews(None, use_ncc=True).copy()
        ctx.set_surface(surf2)
        got[mode] = (vis, ctx.topology_patch_mse().copy())
    for name in ("SMVS_TOPO_DIVIDE", "SMVS_NCC_PAIRS", "SMVS_MSE_SUBS", "SMVS_ZBUF_WINDOW"):
        monkeypatch.delenv(name, raising=False)
    for mode in ("exact", "one_sample_at_a_time", "one_neighbour_at_a_time", "z_buffer_window_3"):
        assert np.array_equal(got["shared"][0], got[mode][0]), mode
        assert np.array_equal(got["shared"][1], got[mode][1]), mode
    assert np.array_equal(got["shared"][0], tp.patch_vis)
    assert (got["shared"][1] > 0).any()
    ctx.close()


@pytest.mark.parametrize("size,n_subs,scale", [((320, 256), 4, 2), ((512, 384), 3, 4),
                                               ((576, 416), 3, 5)])
def test_visibility_masks_do_not_depend_on_the_lanes_per_pair(hip, oracle, monkeypatch, size,
                                                              n_subs, scale):
    """Round 6 chose the lanes per (patch, neighbour) of the visibility kernel by
    patch size (csrc/topology.hip, SMVS_VIS_GROUP_<ps>): one lane, a row of
    sixteen, a whole wave and the whole workgroup must give the oracle's masks --
    the group size changes who sums what (DPP rows, permlane swaps, LDS across
    waves), which samples go to the LDS stash and how many groups share a
    workgroup's staged depths, never a decision."""
    prob, surf, ctx, tp = _topology_setup(hip, oracle, size[0], size[1], n_subs, scale, 0.01)
    want = tp.subviews()
    name = "SMVS_VIS_GROUP_%d" % (1 << scale)
    for lanes in (None, 1, 2, 16, 32, 64, 256):
        if lanes is None:
            monkeypatch.delenv(name, raising=False)
        else:
            monkeypatch.setenv(name, str(lanes))
        ctx.set_surface(surf)
        got = ctx.topology_subviews(None, use_ncc=True)
        assert np.array_equal(got, want), lanes
    monkeypatch.delenv(name, raising=False)
    ctx.close()


def test_topology_subviews_with_sgm_depth(hip, oracle):
    """use_sgm variant: the SGM depth is splatted into the z-buffers too and
    the NCC test is skipped (depth_optimizer.cc:463-466, 577-580)."""
    from smvs_amd import synth
    prob, surf, ctx, tp = _topology_setup(hip, oracle, 320, 256, 3, 2, 0.0)
    xs, ys = np.meshgrid(np.arange(320, dtype=float), np.arange(256, dtype=float))
    sgm = synth.depth_at(prob["scene"], prob["main"], xs, ys).astype(np.float32)
    sgm[::7, ::5] = 0.0            # holes, as SGM leaves them
    sgm[100:140, 150:200] *= 0.8   # a wrong foreground blob occludes real surface
    ctx.set_surface(surf)
    vis_gpu = ctx.topology_subviews(sgm, use_ncc=False)
    vis_ref = tp.subviews(sgm)
    assert np.array_equal(vis_gpu, vis_ref)
    ctx.close()


def test_sgm_init_depth_from_the_stored_map_converts_with_the_hosts_bits(hip, oracle):
    """smvs_ctx_sgm_init_depth_mve takes the "smvs-sgm" embedding as the view
    stores it (MVE's ray-length convention, stereo_view.h:100-135) and turns it
    into z-depth in the kernel that fetches it: the filtered map must be the one
    smvs_ctx_sgm_init_depth gives for the host's conversion
    (depthmap_convert_conventions: float products, float square root, the
    quotient 1 / len and the product in double) -- bit for bit."""
    from smvs_amd import synth
    W, H = 320, 256
    prob, surf, ctx, tp = _topology_setup(hip, oracle, W, H, 3, 2, 0.0)
    lw, lh = W // 2, H // 2
    xs, ys = np.meshgrid(np.arange(0, W, 2, dtype=float) + 0.5, np.arange(0, H, 2, dtype=float) + 0.5)
    z = synth.depth_at(prob["scene"], prob["main"], xs, ys).astype(np.float32)
    # (the scene is a plane at depth 8: make the values ones that do not survive
    # the round trip through the embedding unchanged)
    z *= (np.float32(1.0) + np.float32(0.1) * np.random.default_rng(5).random(z.shape, dtype=np.float32))
    z[::7, ::5] = 0.0
    inv = _inverse_calibration(prob["main"].flen, lw, lh)
    f = np.float32
    px = (np.arange(lw, dtype=np.float32) + f(0.5))[None, :]
    py = (np.arange(lh, dtype=np.float32) + f(0.5))[:, None]
    v = [inv[3 * r] * px + inv[3 * r + 1] * py + inv[3 * r + 2] for r in range(3)]
    length = np.sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]).astype(np.float32)
    assert all(a.dtype == np.float32 for a in v)
    stored = (z.astype(np.float64) * length.astype(np.float64)).astype(np.float32)      # to MVE
    back = (stored.astype(np.float64) * (1.0 / length.astype(np.float64))).astype(np.float32)
    want = ctx.sgm_init_depth(back)
    got = ctx.sgm_init_depth_mve(stored, inv)
    assert (want > 0).any() and np.array_equal(got, want)
    # ... and from the z-depth the SGM front end produced, both conversions on the device
    assert np.array_equal(ctx.sgm_init_depth_mve(z, inv, dm_is_z_depth=True), want)
    assert not np.array_equal(back, z)       # (the round trip is not the identity)
    ctx.close()


@pytest.mark.parametrize("channels", [3, 1])
def test_bilateral_filter_from_the_triangle_of_all_pairs_at_full_size(hip, monkeypatch, channels):
    """Round 6: the colour weights of smvs_ctx_sgm_init_depth as ONE lookup in the
    triangle of all byte pairs held in LDS (bilateral_triangle_kernel) against
    the compressed table with its selector (SMVS_BILATERAL=compressed; that one
    is the oracle's to the bit, test below): 1920 x 1080, RGB and grey guidance,
    a 960 x 540 map with holes."""
    rng = np.random.default_rng(5 + channels)
    W, H = 1920, 1080
    shape = (H, W, 3) if channels == 3 else (H, W)
    # smooth + noise: neighbouring bytes differ by small and by large amounts
    base = rng.integers(0, 256, size=(H // 8 + 1, W // 8 + 1) + shape[2:]).astype(np.float32)
    img = np.kron(base, np.ones((8, 8) + (1,) * (len(shape) - 2), np.float32))[:H, :W]
    img = np.clip(img + rng.normal(0, 6, size=shape), 0, 255).astype(np.uint8)
    low = (2.0 + rng.random((H // 2, W // 2))).astype(np.float32)
    low[rng.random(low.shape) < 0.2] = 0.0
    low[100:140, 300:420] = 0.0
    got = {}
    for form in ("triangle", "compressed"):
        monkeypatch.delenv("SMVS_BILATERAL", raising=False)
        if form == "compressed":
            monkeypatch.setenv("SMVS_BILATERAL", "compressed")
        ctx = hip.ViewContext(W, H, 1)
        ctx.upload_image(-1, img)
        got[form] = ctx.sgm_init_depth(low)
        ctx.close()
    monkeypatch.delenv("SMVS_BILATERAL", raising=False)
    assert np.count_nonzero(got["triangle"]) > 0.5 * W * H
    assert np.array_equal(got["triangle"], got["compressed"])


def test_context_sgm_init_depth_is_the_bilateral_filter_and_stays_resident(hip, oracle):
    """smvs_ctx_sgm_init_depth: depthmap_bilateral_filter guided by the main
    image the context already holds (depth_optimizer.cc:35-51) -- bit-identical
    to the stand-alone entry point with the float image, equal to the oracle --
    and the filtered map stays on the device for create_subview_surfaces:
    smvs_topology_subviews(NULL) then gives what it gives for the explicit map."""
    from smvs_amd import synth
    W, H = 320, 256
    prob, surf, ctx, tp = _topology_setup(hip, oracle, W, H, 3, 2, 0.0)
    xs, ys = np.meshgrid(np.arange(0, W, 2, dtype=float) + 0.5, np.arange(0, H, 2, dtype=float) + 0.5)
    low = synth.depth_at(prob["scene"], prob["main"], xs, ys).astype(np.float32)
    low[::9, ::4] = 0.0
    low[40:60, 70:100] *= 0.8
    ci = prob["images"][0].astype(np.float32) / np.float32(255.0)
    ctx.set_surface(surf)
    # before anything is resident: NULL means no SGM splat
    vis_none = ctx.topology_subviews(None, use_ncc=False)
    full = ctx.sgm_init_depth(low)
    # the colour and spatial weights come from tables the host fills with expf:
    # the CPU path's weights exactly
    assert np.array_equal(full, oracle.bilateral_upsample(low, ci))
    # the stand-alone entry point (any float guidance image) takes the
    # exponentials on the device, in double, rounded once: an ulp here and there
    alone = hip.bilateral_upsample(low, ci)
    assert np.array_equal(full == 0, alone == 0)
    assert np.max(np.abs(full - alone)) <= 4e-7 * np.max(np.abs(full))
    vis_resident = ctx.topology_subviews(None, use_ncc=False)
    assert np.array_equal(vis_resident, tp.subviews(full))
    assert not np.array_equal(vis_resident, vis_none)   # the wrong blob occludes
    # an explicit map replaces the resident one ...
    other = full.copy(); other[:] = 0.0
    assert np.array_equal(ctx.topology_subviews(other, use_ncc=False), vis_none)
    # ... and NULL does not silently fall back to stale data afterwards
    assert np.array_equal(ctx.topology_subviews(None, use_ncc=False), vis_none)
    ctx.sgm_init_depth(low)
    ctx.sgm_init_depth(None)
    assert np.array_equal(ctx.topology_subviews(None, use_ncc=False), vis_none)
    # no main image: a state error, not a crash
    from smvs_amd._capi import SmvsError
    ctx2 = hip.ViewContext(W, H, 3)
    with pytest.raises(SmvsError):
        ctx2.sgm_init_depth(low)
    ctx2.close()
    ctx.close()


def test_full_size_shading_step_matches_oracle(hip, oracle):
    """configs[3]: the shading-aware step at 1920x1080 with 8 neighbours --
    SH lighting fit (light_optimizer.cc:22-55) and the construct with the
    shading residual (gauss_newton_step.cc:420-517), sampled patches."""
    import bench
    from smvs_amd import synth
    prob = synth.make_problem(bench.W, bench.H, bench.NSUBS, bench.SCALE, shading=True,
                              noise=bench.NOISE)
    surf = prob["surf"]
    ctx = hip.ViewContext(bench.W, bench.H, bench.NSUBS)
    ctx.set_views(prob["views"])
    ctx.set_surface(surf)
    orc = oracle.OracleProblem(surf, prob["views"])
    # lighting fit on the current surface
    A_gpu, b_gpu = ctx.light_accumulate()
    normals = orc.normal_map()
    A_ref, b_ref = oracle.light_accumulate(normals, prob["views"]["shading"])
    assert _rel(A_gpu, A_ref) < 1e-9 and _rel(b_gpu, b_ref) < 1e-9
    lighting = oracle.light_solve(A_ref, b_ref)
    # construct with the shading term, geometric regulariser off (Q8) and on
    valid = np.flatnonzero(surf["patch_valid"])
    pick = np.random.default_rng(9).choice(valid, size=48, replace=False)
    for light_reg in (0.0, 0.5):
        n = ctx.gn_construct(bench.REG, light_reg, lighting)
        assert n == int(surf["patch_valid"].sum())
        Hp, gp = ctx.gn_patch_systems()
        for p in pick:
            g_ref, H_ref = orc.gn_patch(int(p), bench.REG, light_reg, lighting)
            assert _rel(np.triu(Hp[p]), np.triu(H_ref)) < 1e-9
            assert _rel(gp[p], g_ref) < 1e-9
    ctx.close()


# ---------------------------------------------------------------------------
# edge cases of the construct / loop
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("scale,size,n_subs", [(6, (512, 384), 3),    # 256 samples: 4 chunks per patch
                                                (2, (160, 128), 1),    # single neighbour: no pair terms
                                                (2, (160, 128), 16),   # SMVS_MAX_SUBS
                                                (3, (333, 217), 5),    # ragged image size
                                                (1, (96, 64), 2)])     # 2x2 patches: 4 samples per patch
def test_patch_systems_edge_cases(hip, oracle, scale, size, n_subs):
    prob, ctx, orc = _setup(hip, oracle, size[0], size[1], n_subs, scale)
    reg = 0.01
    n = ctx.gn_construct(reg)
    assert n == int(prob["surf"]["patch_valid"].sum()) and n > 0
    Hp, gp = ctx.gn_patch_systems()
    valid = np.flatnonzero(prob["surf"]["patch_valid"])
    pick = np.random.default_rng(3).choice(valid, size=min(30, valid.size), replace=False)
    for p in pick:
        g_ref, H_ref = orc.gn_patch(int(p), reg)
        assert _rel(np.triu(Hp[p]), np.triu(H_ref)) < 1e-9
        assert _rel(gp[p], g_ref) < 1e-9
    ctx.close()


def test_full_optimization_loop_matches_oracle(hip, oracle):
    """full_optimization: every node stays active, the loop ends when the
    mean reprojection shift drops under 0.01 px (depth_optimizer.cc:277-290)."""
    prob, ctx, orc = _setup(hip, oracle, 256, 192, 4, 3, noise=0.01)
    reg = 0.01
    stats = ctx.run_loop(reg, full_optimization=True, max_newton_steps=8)
    active = prob["surf"]["node_valid"].copy()
    steps = 0
    while steps < 8:
        steps += 1
        ref = orc.gn_construct(active, reg)
        x, _, _ = orc.cg_solve(ref["H9"], ref["present"], ref["P"], -ref["g"], 200,
                               0.01 * np.linalg.norm(ref["g"]), 1e-3)
        _, _, mean = orc.update_and_reactivate(x, active, full_optimization=True)
        if mean < 0.01:
            break
    assert stats["newton_steps"] == steps
    assert _rel(ctx.depth_map(), orc.depth_map()) <= 1e-4
    ctx.close()


def test_host_optimize_matches_oracle_960x540(hip, oracle):
    """Whole optimize() (scales 5..2, device Newton loop, device topology
    tests, host grid surgery) on a 960x540 sphere scene with 4 neighbours
    against the oracle: identical batch log, same valid pixels, depth
    relative L2 <= 1e-4 (north_star tolerance)."""
    from smvs_amd import synth, host
    inputs = synth.pipeline_inputs("sphere", 960, 540, 4, flen=1.2)
    got = host.optimize(inputs, regularization=0.01, num_iterations=5, min_scale=2)
    want = oracle.optimize(inputs, regularization=0.01, num_iterations=5, min_scale=2)
    assert_same_units(got["log"], want["log"], 960, 540, "sphere_960x540")
    assert len(got["log"]) >= 8
    assert np.array_equal(got["depth"] > 0, want["depth"] > 0)
    assert _rel(got["depth"], want["depth"]) <= 1e-4


_LOOP_PROBE = r"""
import json, sys
import numpy as np
import torch  # noqa: F401  (one HIP runtime, see conftest.py)
sys.path.insert(0, sys.argv[1])
import smvs_amd
from smvs_amd import synth
prob = synth.make_problem(256, 192, 4, 2, noise=0.01)
ctx = smvs_amd.ViewContext(256, 192, 4)
ctx.set_views(prob["views"])
ctx.set_surface(prob["surf"])
ctx.profile(True)
stats = ctx.run_loop(0.01, max_newton_steps=6, active_threshold=0.002)
launches = {k: int(v[1]) for k, v in ctx.profile_get().items()}
np.save(sys.argv[2], ctx.get_nodes())
print(json.dumps(dict(stats={k: int(v) for k, v in stats.items()}, launches=launches)))
"""


@pytest.mark.parametrize("mode", ["undersize", "solver"])
def test_gn_loop_abandoned_steps_are_redone(hip, mode, tmp_path):
    """The launch-ahead Newton loop abandons a step whose launches were sized
    for a shorter live list (and re-enqueues it) or whose resident solver gave
    up (and goes on with the streaming solver).  SMVS_LOOP_TEST forces either
    case in a child process; the result has to be the one of the undisturbed
    loop: bit for bit when only launches were redone, to 1e-9 when the other
    solver (another summation order) finished the loop."""
    import json, os, subprocess, sys
    from smvs_amd import synth
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "nodes.npy")
    env = dict(os.environ, SMVS_LOOP_TEST=mode)
    res = subprocess.run([sys.executable, "-c", _LOOP_PROBE, root, out], env=env,
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    reply = json.loads(res.stdout.strip().splitlines()[-1])
    child, launches = reply["stats"], reply["launches"]
    if launches["cg_resident"] == 0:
        pytest.skip("the resident solver does not apply on this device (fewer CUs "
                    "than tiles, or SMVS_CG_RESIDENT=0): the launch-ahead loop is not used")
    # the hook fired: launches were repeated / the streaming solver ran
    if mode == "undersize":
        assert launches["patch"] > child["newton_steps"] + 1
    else:
        assert launches["cg_spmv"] > 0 and 2 <= launches["cg_resident"] <= 3
    prob = synth.make_problem(256, 192, 4, 2, noise=0.01)
    ctx = hip.ViewContext(256, 192, 4)
    ctx.set_views(prob["views"])
    ctx.set_surface(prob["surf"])
    stats = ctx.run_loop(0.01, max_newton_steps=6, active_threshold=0.002)
    assert stats["newton_steps"] >= 4   # (long enough for the hooks to fire)
    nodes = ctx.get_nodes()
    ctx.close()
    child_nodes = np.load(out)
    for key in ("newton_steps", "active_patch_steps", "final_active_nodes", "nan_break"):
        assert child[key] == int(stats[key]), key
    if mode == "undersize":
        assert child["linear_iterations"] == int(stats["linear_iterations"])
        assert np.array_equal(child_nodes, nodes)
    else:
        assert _rel(child_nodes, nodes) <= 1e-9


def test_save_and_restore_nodes(hip, oracle):
    """smvs_ctx_save_nodes / smvs_ctx_restore_nodes: the saved start surface
    comes back bit for bit after a Newton loop, a second loop from it repeats
    the first one exactly, and restoring without a save is an error."""
    prob, ctx, _ = _setup(hip, oracle, 192, 128, 3, 2, noise=0.01)
    with pytest.raises(RuntimeError):
        ctx.restore_nodes()
    start = ctx.get_nodes()
    ctx.save_nodes()
    s1 = ctx.run_loop(0.01, max_newton_steps=4)
    n1 = ctx.get_nodes()
    assert not np.array_equal(n1, start)
    ctx.restore_nodes()
    assert np.array_equal(ctx.get_nodes(), start)
    s2 = ctx.run_loop(0.01, max_newton_steps=4)
    assert s1 == s2
    assert np.array_equal(ctx.get_nodes(), n1)
    ctx.close()


@pytest.mark.parametrize("solver", SOLVERS)
@pytest.mark.parametrize("full_optimization", [False, True])
def test_gn_loop_from_a_partial_active_set(hip, oracle, full_optimization, solver):
    """smvs_gn_run_loop with reset_active = 0: the loop starts from the active
    set the caller uploaded (the loop-begin kernel counts it and builds the
    first live list from it) -- against the oracle's loop from the same set."""
    prob, ctx, orc = _setup(hip, oracle, 224, 160, 3, 2, noise=0.02, solver=solver)
    rng = np.random.default_rng(5)
    valid = prob["surf"]["node_valid"]
    active = (valid & (rng.random(valid.size) < 0.4)).astype(np.uint8)
    ctx.set_active(active)
    stats = ctx.run_loop(0.01, max_newton_steps=5, reset_active=False,
                         full_optimization=full_optimization)
    n_init = int(active.sum()); n_act = n_init; steps = 0; patch_steps = 0; its = 0
    act = active.copy()
    while steps < 5 and n_act > n_init // 20:
        steps += 1
        ref = orc.gn_construct(act, 0.01)
        patch_steps += ref["active_patches"]
        x, it, _ = orc.cg_solve(ref["H9"], ref["present"], ref["P"], -ref["g"], 200,
                                0.01 * np.linalg.norm(ref["g"]), 1e-3)
        its += it
        new_act, n_new, mean = orc.update_and_reactivate(x, act, full_optimization)
        if full_optimization:
            if mean < 0.01:
                break
        else:
            act, n_act = new_act, n_new
    assert stats["newton_steps"] == steps
    assert stats["active_patch_steps"] == patch_steps
    assert stats["linear_iterations"] == its
    assert stats["final_active_nodes"] == n_act
    assert _rel(ctx.depth_map(), orc.depth_map()) <= 1e-5
    ctx.close()


# ---------------------------------------------------------------------------
# grids beyond the resident solver (> 131 k nodes): the streaming path is what
# the product runs there, chosen automatically
# ---------------------------------------------------------------------------
def test_large_grid_takes_the_streaming_path_and_matches_oracle(hip, oracle):
    """2304x1296 at scale 2: 575 x 323 patches, 186,624 nodes -- more than the
    256 x 512 nodes the chip-resident solver holds, so smvs_cg_solve and
    smvs_gn_run_loop fall through to the assembly kernel + streaming PCG
    without being told to.  One solve on the oracle's system (iteration count,
    info, x) and a Newton loop against the oracle's loop."""
    from smvs_amd import synth
    W, H, S = 2304, 1296, 4
    prob = synth.make_problem(W, H, S, 2, noise=0.004)
    surf = prob["surf"]
    assert (surf["npx"] + 1) * (surf["npy"] + 1) > 256 * 512
    ctx = hip.ViewContext(W, H, S)
    ctx.set_views(prob["views"])
    ctx.set_surface(surf)
    orc = oracle.OracleProblem(surf, prob["views"])
    oracle.lib().orc_set_threads(max(1, min(os.cpu_count() or 1, 64)))
    try:
        active = surf["node_valid"].copy()
        ref = orc.gn_construct(active, 0.01)
        ctx.profile(True); ctx.profile_reset()
        n = ctx.gn_construct(0.01)
        assert n == ref["active_patches"]
        H9, g, P = ctx.gn_download()
        # (Frobenius norm over 186 k nodes; measured 1.2e-9, dominated by the
        # few patches whose IRLS weights sit at 1 / 1e-4)
        assert _rel(H9, ref["H9"]) < 1e-8 and _rel(g, ref["g"]) < 1e-8
        ctx.gn_upload(ref["H9"], ref["g"], ref["P"])
        it, info = ctx.cg_solve(200, -1.0, 1e-3)
        xr, itr, infor = orc.cg_solve(ref["H9"], ref["present"], ref["P"], -ref["g"], 200,
                                      0.01 * np.linalg.norm(ref["g"]), 1e-3)
        assert (it, info) == (itr, infor)
        assert _rel(ctx.cg_x(), xr) < 1e-9
        # the whole loop, two steps
        ctx.set_nodes(surf["nodes"])
        stats = ctx.run_loop(0.01, max_newton_steps=2)
        launches = {k: int(v[1]) for k, v in ctx.profile_get().items()}
        assert launches["cg_resident"] == 0 and launches["cg_spmv"] > 0
        n_init = int(active.sum()); n_act = n_init; steps = 0; patch_steps = 0; its = 0
        while steps < 2 and n_act > n_init // 20:
            steps += 1
            ref = orc.gn_construct(active, 0.01)
            patch_steps += ref["active_patches"]
            x, k, _ = orc.cg_solve(ref["H9"], ref["present"], ref["P"], -ref["g"], 200,
                                   0.01 * np.linalg.norm(ref["g"]), 1e-3)
            its += k
            active, n_act, _ = orc.update_and_reactivate(x, active)
    finally:
        oracle.lib().orc_set_threads(1)
    assert stats["newton_steps"] == steps
    assert stats["active_patch_steps"] == patch_steps
    assert stats["linear_iterations"] == its
    assert stats["final_active_nodes"] == n_act
    assert _rel(ctx.depth_map(), orc.depth_map()) <= 1e-5
    ctx.close()


@pytest.mark.parametrize("solver", ["resident_ref", "streaming"])
def test_host_optimize_same_result_with_every_solver(hip, oracle, solver):
    """The whole optimize() with the reference-order resident solver and with
    the streaming solver against the default (one exchange per iteration):
    identical batch log, depth within the north-star tolerance of each other
    (each is within it of the oracle: test_host_optimize_matches_oracle_960x540)."""
    from smvs_amd import synth, host
    # (a scene on which the oracle's SSE and scalar branches agree to 5e-9)
    inputs = synth.pipeline_inputs("plane", 480, 320, 3)
    base = host.optimize(inputs, regularization=0.01, num_iterations=4, min_scale=2)
    got = host.optimize(inputs, regularization=0.01, num_iterations=4, min_scale=2,
                        solver=solver)
    assert _same_control_flow(got["log"], base["log"]), (got["log"], base["log"])
    assert np.array_equal(got["depth"] > 0, base["depth"] > 0)
    assert _rel(got["depth"], base["depth"]) <= 1e-4


def test_restore_nodes_is_bound_to_the_saved_surface(hip, oracle):
    """smvs_ctx_restore_nodes after a NEW smvs_ctx_set_surface of the same grid
    size is an error (the saved nodes belong to another surface), not a silent
    restore."""
    prob, ctx, _ = _setup(hip, oracle, 192, 128, 3, 2, noise=0.01)
    ctx.save_nodes()
    ctx.restore_nodes()
    other = dict(prob["surf"])
    other["nodes"] = prob["surf"]["nodes"] * 1.01
    ctx.set_surface(other)
    with pytest.raises(RuntimeError):
        ctx.restore_nodes()
    ctx.save_nodes()
    ctx.restore_nodes()
    ctx.close()


_WIDE_PROBE = _LOOP_PROBE.replace("active_threshold=0.002", "active_threshold=0.15")


def test_gn_loop_wide_counters(hip, tmp_path):
    """Surfaces of 2^24 nodes and more end a step with two atomics per
    workgroup instead of one packed word (finish_step_kernel, FinishArgs::wide);
    SMVS_LOOP_TEST=wide uses that path on a small surface: bit-identical to the
    packed path."""
    import json, subprocess, sys
    from smvs_amd import synth
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "nodes.npy")
    env = dict(os.environ, SMVS_LOOP_TEST="wide")
    res = subprocess.run([sys.executable, "-c", _WIDE_PROBE, root, out], env=env,
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    child = json.loads(res.stdout.strip().splitlines()[-1])["stats"]
    prob = synth.make_problem(256, 192, 4, 2, noise=0.01)
    ctx = hip.ViewContext(256, 192, 4)
    ctx.set_views(prob["views"])
    ctx.set_surface(prob["surf"])
    stats = ctx.run_loop(0.01, max_newton_steps=6)
    assert stats["newton_steps"] >= 2
    for key in ("newton_steps", "active_patch_steps", "final_active_nodes",
                "linear_iterations", "nan_break"):
        assert child[key] == int(stats[key]), key
    assert np.array_equal(np.load(out), ctx.get_nodes())
    ctx.close()


# The problems of tools/fuzz_parity.py where the device's CG iteration total
# differed from the oracle's in a sweep: tiny, ill-conditioned scale-1 /
# scale-2 surfaces with the shading term, solves of 50-70 iterations.
#   0-2: profiles/r2_fuzz_parity.txt (150 cases, seed 11) cases 6, 57, 81
#   3-6: profiles/r3_fuzz_parity.txt (150 cases, seed 0) cases 15, 59, 61, 130
#        -- found with the one-exchange recurrence, which AUTO no longer runs
#        on single-tile grids (cg_resident.hip resident_plan)
FUZZ_OUTLIERS = [
    dict(w=38, h=49, scale=1, n_subs=4, noise=0.03, seed=8446, max_steps=6),
    dict(w=66, h=29, scale=2, n_subs=4, noise=0.03, seed=4778, max_steps=5),
    dict(w=33, h=31, scale=1, n_subs=7, noise=0.01, seed=2627, max_steps=4),
    dict(w=45, h=31, scale=1, n_subs=8, noise=0.01, seed=1440, max_steps=1),
    dict(w=57, h=48, scale=1, n_subs=6, noise=0.03, seed=2853, max_steps=4),
    dict(w=43, h=26, scale=1, n_subs=8, noise=0.01, seed=7132, max_steps=4, light_reg=0.0),
    dict(w=53, h=33, scale=1, n_subs=4, noise=0.03, seed=4110, max_steps=1),
]


@pytest.mark.parametrize("solver", SOLVERS)
@pytest.mark.parametrize("case", range(len(FUZZ_OUTLIERS)))
def test_fuzz_outliers_keep_the_control_flow(hip, oracle, case, solver):
    """On these systems the iteration count of a long solve is not a function
    of the system alone: the oracle's OWN C solve ends outlier 6 after 68 or
    after 61 iterations depending on a 1e-12 relative perturbation of g, outlier
    3 after 50, 51 or 54 (tests/test_oracle_solver_math.py,
    test_oracle_iteration_count_flips_under_input_noise;
    profiles/r5_cg_association.txt) -- the convergence test zeta < 1e-3 comes
    within rounding of triggering seven iterations before it finally does.  The
    device's g and H agree with the oracle's to 1e-12 / 1e-10, not to the bit, so
    no solver can be held to the oracle's count here, whatever the association
    of its sums (round 5 gave the streaming solver exact sums of the rounded
    products, csrc/cg.hip: outlier 6 still ends after 61, the other mode).
    What holds, and is asserted for every solver: the number of Newton steps
    is the oracle's and the CG iteration total is within the spread the oracle
    shows against itself: 2 per solve or 12 %, whichever is larger (measured:
    resident solvers 0-2, streaming 0-7 of 68).  When the totals agree, the
    whole control flow (active patches per step, final active set) is
    identical and the depth is within the north-star 1e-4.  When a solve ended
    apart, x differs at the solver's own 1e-3 tolerance: the depth bound is then
    5e-4 (measured: 2.6e-4 at most) and a re-activation decision at the 0.15 px
    threshold may flip (2 of 270 active patch-steps on outlier 0), so the
    active-patch total is then asserted to 2 % only."""
    from smvs_amd import synth
    c = FUZZ_OUTLIERS[case]
    prob = synth.make_problem(c["w"], c["h"], c["n_subs"], c["scale"], shading=True,
                              noise=c["noise"], seed=c["seed"])
    surf, lighting, light_reg = prob["surf"], prob["lighting"], c.get("light_reg", 0.5)
    ctx = hip.ViewContext(c["w"], c["h"], c["n_subs"])
    ctx.set_solver(solver)
    ctx.set_views(prob["views"]); ctx.set_surface(surf)
    orc = oracle.OracleProblem(surf, prob["views"])
    stats = ctx.run_loop(0.01, light_reg, lighting, max_newton_steps=c["max_steps"])
    act = surf["node_valid"].copy()
    n_init = int(act.sum()); n_act = n_init; steps = its = psteps = 0
    while steps < c["max_steps"] and n_act > n_init // 20:
        steps += 1
        ref = orc.gn_construct(act, 0.01, light_reg, lighting)
        psteps += ref["active_patches"]
        xr, itr, _ = orc.cg_solve(ref["H9"], ref["present"], ref["P"], -ref["g"], 200,
                                  0.01 * np.linalg.norm(ref["g"]), 1e-3)
        its += itr
        act, n_act, _ = orc.update_and_reactivate(xr, act)
    ed = _rel(ctx.depth_map(), orc.depth_map())
    print("fuzz outlier %d [%s]: steps %d, CG iterations oracle %d device %d, depth %.2e"
          % (case, solver, steps, its, stats["linear_iterations"], ed))
    assert stats["newton_steps"] == steps
    slack = max(2 * steps, int(np.ceil(0.12 * its)))
    assert abs(stats["linear_iterations"] - its) <= slack
    if stats["linear_iterations"] == its:
        assert (stats["active_patch_steps"], stats["final_active_nodes"]) == (psteps, n_act)
        assert ed <= 1e-4
    else:
        assert abs(stats["active_patch_steps"] - psteps) <= 0.02 * psteps
        assert abs(stats["final_active_nodes"] - n_act) <= 0.05 * n_init
        assert ed <= 5e-4
    ctx.close()


_SHARED_GPU_PROBE = r"""
import json, sys
import numpy as np
import torch  # noqa: F401  (one HIP runtime, see conftest.py)
sys.path.insert(0, sys.argv[1])
import smvs_amd
from smvs_amd import synth
prob = synth.make_problem(512, 384, 4, 2, noise=0.01)
ctx = smvs_amd.ViewContext(512, 384, 4)
ctx.set_views(prob["views"])
ctx.set_surface(prob["surf"])
ctx.save_nodes()
ctx.profile(True)
stats = []
for rep in range(30):
    ctx.restore_nodes()
    stats.append(ctx.run_loop(0.01, max_newton_steps=4))
launches = {k: int(v[1]) for k, v in ctx.profile_get().items()}
np.save(sys.argv[2], ctx.get_nodes())
print(json.dumps(dict(steps=[int(s["newton_steps"]) for s in stats], launches=launches)))
"""


def test_two_processes_share_one_gpu_without_losing_the_resident_solver(hip, tmp_path):
    """The resident PCG needs every workgroup of its launch co-resident; two
    such kernels started together by two PROCESSES on one GPU could each hold
    half of the CUs.  The per-device lock is therefore also a file lock
    (DeviceBarrierLock): both processes keep the resident solver (no give-up
    into the streaming kernels) and compute the same nodes as a process alone."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SMVS_LOCK_DIR=str(tmp_path))
    procs = [subprocess.Popen([sys.executable, "-c", _SHARED_GPU_PROBE, root,
                               str(tmp_path / ("nodes%d.npy" % i))], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for i in range(2)]
    outs = [p.communicate(timeout=900) for p in procs]
    replies = []
    for p, (out, err) in zip(procs, outs):
        assert p.returncode == 0, err[-2000:]
        replies.append(json.loads(out.strip().splitlines()[-1]))
    if replies[0]["launches"]["cg_resident"] == 0:
        pytest.skip("the resident solver does not apply on this device")
    for r in replies:
        assert r["launches"]["cg_spmv"] == 0, r["launches"]      # never fell back
        assert r["launches"]["cg_resident"] >= sum(r["steps"])
        assert r["steps"] == replies[0]["steps"]
    assert np.array_equal(np.load(str(tmp_path / "nodes0.npy")),
                          np.load(str(tmp_path / "nodes1.npy")))
    assert any(f.startswith("smvs_hip_barrier_") for f in os.listdir(str(tmp_path)))


_COMPACT_PROBE = r"""
import json, sys
import numpy as np
import torch  # noqa: F401  (one HIP runtime, see conftest.py)
sys.path.insert(0, sys.argv[1])
import smvs_amd
from smvs_amd import synth
prob = synth.make_problem(640, 480, 3, 1, noise=0.004)
active = np.load(sys.argv[3])
ctx = smvs_amd.ViewContext(640, 480, 3)
ctx.set_views(prob["views"])
ctx.set_surface(prob["surf"])
ctx.set_active(active)
ctx.profile(True)
stats = ctx.run_loop(0.01, max_newton_steps=3, reset_active=False)
launches = {k: int(v[1]) for k, v in ctx.profile_get().items()}
np.save(sys.argv[2], ctx.get_nodes())
print(json.dumps(dict(stats={k: int(v) for k, v in stats.items()}, launches=launches)))
"""


def _partial_active_set(surf):
    """A third of the grid, a block far away and one lonely node: most tiles of
    the resident solver's grid have no active node, some groups of sixteen lose
    their first tile, one tile lives on a single node."""
    stride, rows = surf["npx"] + 1, surf["npy"] + 1
    act = np.zeros((rows, stride), np.uint8)
    act[:, : stride // 3] = 1
    act[rows - 40: rows - 12, stride - 60: stride - 25] = 1
    act[rows // 2, stride - 7] = 1
    return (act.reshape(-1) & surf["node_valid"]).astype(np.uint8)


def test_compacted_solve_is_bit_identical_and_matches_oracle(hip, oracle, tmp_path):
    """The reference's system holds the active nodes only
    (gauss_newton_step.cc:73-79, 91-105); the resident solver leaves out what has
    no active node -- tiles without one leave the launch, inactive rim nodes are
    neither published nor polled (cg_resident.hip, "the compacted solve").  On a
    76 k-node grid (150 tiles, 10 groups) with two thirds of the tiles dead:
    the Newton loop of the default build, of SMVS_CG_COMPACT=0 (every tile takes
    part, rounds 2-4) and of SMVS_HP_LAYOUT=aos (the per-patch systems patch-major
    instead of in planes) give the SAME bits -- nodes, steps, CG iterations,
    active patch-steps -- and all of them the oracle's control flow with the
    depth within 1e-5."""
    import json, subprocess, sys
    from smvs_amd import synth
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prob = synth.make_problem(640, 480, 3, 1, noise=0.004)
    surf = prob["surf"]
    active = _partial_active_set(surf)
    assert 0.2 < active.sum() / surf["node_valid"].sum() < 0.6
    np.save(str(tmp_path / "active.npy"), active)
    replies, nodes = {}, {}
    for tag, extra in (("default", {}), ("full_grid", {"SMVS_CG_COMPACT": "0"}),
                       ("patch_major", {"SMVS_HP_LAYOUT": "aos"})):
        env = dict(os.environ, SMVS_LOCK_DIR=str(tmp_path), **extra)
        out = subprocess.run([sys.executable, "-c", _COMPACT_PROBE, root,
                              str(tmp_path / (tag + ".npy")), str(tmp_path / "active.npy")],
                             env=env, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-2000:]
        replies[tag] = json.loads(out.stdout.strip().splitlines()[-1])
        nodes[tag] = np.load(str(tmp_path / (tag + ".npy")))
    if replies["default"]["launches"]["cg_resident"] == 0:
        pytest.skip("the resident solver does not apply on this device")
    for tag in replies:
        assert replies[tag]["launches"]["cg_spmv"] == 0, (tag, replies[tag]["launches"])
        assert replies[tag]["stats"] == replies["default"]["stats"], tag
        assert np.array_equal(nodes[tag], nodes["default"]), tag
    # ... and the oracle's loop from the same active set
    oracle.lib().orc_set_threads(max(1, min(os.cpu_count() or 1, 64)))
    try:
        orc = oracle.OracleProblem(surf, prob["views"])
        act = active.copy()
        n_init = int(act.sum()); n_act = n_init; steps = its = psteps = 0
        while steps < 3 and n_act > n_init // 20:
            steps += 1
            ref = orc.gn_construct(act, 0.01)
            psteps += ref["active_patches"]
            xr, itr, _ = orc.cg_solve(ref["H9"], ref["present"], ref["P"], -ref["g"], 200,
                                      0.01 * np.linalg.norm(ref["g"]), 1e-3)
            its += itr
            act, n_act, _ = orc.update_and_reactivate(xr, act)
    finally:
        oracle.lib().orc_set_threads(1)
    st = replies["default"]["stats"]
    assert (st["newton_steps"], st["active_patch_steps"], st["linear_iterations"],
            st["final_active_nodes"]) == (steps, psteps, its, n_act)
    ctx = hip.ViewContext(640, 480, 3)
    ctx.set_views(prob["views"]); ctx.set_surface(surf)
    ctx.set_nodes(nodes["default"])
    assert _rel(ctx.depth_map(), orc.depth_map()) <= 1e-5
    ctx.close()


def test_patch_chunks_on_four_waves_equal_one_wave(hip, tmp_path):
    """Scale 6 (16 x 16 samples per patch): the four chunks of a patch on four
    waves of a workgroup (gn_patch_kernel<1, 4>, the default: phase 1 side by
    side, the matrix-core accumulation in turns in chunk order) against one wave
    walking them one after the other (SMVS_PATCH_SPLIT=0, rounds 1-4): the SAME
    bits in every patch system and the same Newton loop, statistic for
    statistic.  (A version that added the four partial systems afterwards was
    1e-16 away -- and moved three patches of a 1920 x 1080 optimize() across a
    validity decision five scales later: test_full_size_optimize_basic.)"""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    probe = r"""
import json, sys
import numpy as np
import torch  # noqa: F401
sys.path.insert(0, sys.argv[1])
import smvs_amd
from smvs_amd import synth
prob = synth.make_problem(704, 512, 5, 6, noise=0.01)
ctx = smvs_amd.ViewContext(704, 512, 5)
ctx.set_views(prob["views"]); ctx.set_surface(prob["surf"])
ctx.gn_construct(0.01)
Hp, gp = ctx.gn_patch_systems()
np.save(sys.argv[2], np.concatenate([Hp.reshape(len(Hp), -1), gp.reshape(len(gp), -1)], 1))
stats = ctx.run_loop(0.01, max_newton_steps=4)
print(json.dumps({k: int(v) for k, v in stats.items()}))
"""
    out, stats = {}, {}
    for tag, extra in (("split", {}), ("serial", {"SMVS_PATCH_SPLIT": "0"})):
        env = dict(os.environ, **extra)
        res = subprocess.run([sys.executable, "-c", probe, root, str(tmp_path / (tag + ".npy"))],
                             env=env, capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stderr[-2000:]
        stats[tag] = json.loads(res.stdout.strip().splitlines()[-1])
        out[tag] = np.load(str(tmp_path / (tag + ".npy")))
    assert out["split"].shape == out["serial"].shape and np.abs(out["serial"]).max() > 0
    assert np.array_equal(out["split"], out["serial"])
    assert stats["split"] == stats["serial"] and stats["split"]["newton_steps"] >= 1
