"""More mathematical pins of oracle pieces for which the reference holds no
test or vector (no GPU): each is compared with an independently written
numpy implementation of the published algorithm.

 * the 8-path semi-global aggregation (lib/sgm_stereo.cc:408-667) against
   Hirschmueller's recurrence written per path with numpy;
 * the reprojection-based re-activation of nodes after a Newton step
   (lib/depth_optimizer.cc:271-303, 647-677) against a direct evaluation of
   the two projections per pixel.
"""
import numpy as np
import pytest


# --------------------------------------------------------------------- SGM
def _textbook_paths(cost, p1, p2):
    """L_r(p, d) = C(p, d) + min(L_r(p - r, d), L_r(p - r, d -+ 1) + P1,
    min_k L_r(p - r, k) + P2) - min_k L_r(p - r, k), L_r = C where p - r is
    outside the image; returns the eight L_r."""
    h, w, D = cost.shape
    C = cost.astype(np.int64)
    out = []
    for dx, dy in ((1, 0), (-1, 0), (0, 1), (1, 1), (-1, 1), (0, -1), (1, -1), (-1, -1)):
        L = np.zeros_like(C)
        ys = range(h) if dy >= 0 else range(h - 1, -1, -1)
        xs = range(w) if dx >= 0 else range(w - 1, -1, -1)
        for y in ys:
            for x in xs:
                px, py = x - dx, y - dy
                if px < 0 or px >= w or py < 0 or py >= h:
                    L[y, x] = C[y, x]
                    continue
                prev = L[py, px]
                m = prev.min()
                big = np.iinfo(np.int64).max // 4
                left = np.concatenate(([big], prev[:-1])) + p1
                right = np.concatenate((prev[1:], [big])) + p1
                L[y, x] = C[y, x] + np.minimum(np.minimum(prev, left), np.minimum(right, m + p2)) - m
        out.append(L)
    return out


@pytest.mark.parametrize("p1,p2", [(6, 96), (3, 20)])
def test_sgm_aggregation_is_the_eight_path_recurrence(oracle, p1, p2):
    rng = np.random.default_rng(12)
    h, w, D = 9, 11, 8
    cost = rng.integers(0, 64, size=(h, w, D)).astype(np.uint16)
    cost[rng.random((h, w)) < 0.1] = 255          # unwarped pixels, as the cost volume marks them
    S = oracle.sgm_aggregate(cost, p1, p2).astype(np.int64)
    paths = _textbook_paths(cost, p1, p2)
    T = sum(paths)
    # interior pixels: exactly the sum of the eight path costs
    assert np.array_equal(S[1:-1, 1:-1], T[1:-1, 1:-1])
    # On the border the reference seeds its sweeps per entry row AND per entry
    # column (sgm_stereo.cc:457-464, 511-534, 589-612, Q19): a border pixel
    # collects C once more per extra seed, never anything else.
    extra = S - T
    C = cost.astype(np.int64)
    border = np.ones((h, w), bool); border[1:-1, 1:-1] = False
    for y, x in zip(*np.nonzero(border)):
        d = int(np.argmax(C[y, x]))
        k = int(round(extra[y, x, d] / max(C[y, x, d], 1)))
        assert 0 <= k <= 4 and np.array_equal(extra[y, x], k * C[y, x]), (y, x, k, extra[y, x])
    # ... and the four corners are where the doubled diagonal seeds sit
    assert extra[0, 0].sum() > 0 and extra[h - 1, w - 1].sum() > 0


# ------------------------------------------------- re-activation of nodes
def _patch_depth_at_pixels(nodes16, ps):
    """bicubic Hermite patch (lib/bicubic_patch.cc) at the pixel centres
    (i + 0.5) / ps, written with numpy from the textbook basis"""
    def basis(t):
        return np.array([1 - 3 * t**2 + 2 * t**3, 3 * t**2 - 2 * t**3,
                         t - 2 * t**2 + t**3, t**3 - t**2])
    u = (np.arange(ps) + 0.5) / ps
    B = np.stack([basis(t) for t in u])            # [ps][4]: v0, v1, s0, s1
    n = nodes16.reshape(2, 2, 4)                   # [b][a][f, dx, dy, dxy]
    out = np.zeros((ps, ps))
    for b in range(2):
        for a in range(2):
            f, dx, dy, dxy = n[b, a]
            out += (f * np.outer(B[:, b], B[:, a]) + dx * np.outer(B[:, b], B[:, 2 + a])
                    + dy * np.outer(B[:, 2 + b], B[:, a]) + dxy * np.outer(B[:, 2 + b], B[:, 2 + a]))
    return out                                      # [j][i]


def test_reactivation_is_the_reprojection_shift_test(oracle):
    """After x is added to the nodes, a node stays active iff one of its
    patches has a pixel whose projection into a visible neighbour moved by more
    than 0.15 px (depth_optimizer.cc:277-303); projections as in
    correspondence.cc:36-51 without the half-pixel offsets (:669-672)."""
    from smvs_amd import synth
    prob = synth.make_problem(192, 128, 3, scale=3, noise=0.02, seed=4)
    surf, views = prob["surf"], prob["views"]
    orc = oracle.OracleProblem(surf, views)
    active = surf["node_valid"].copy()
    sysm = orc.gn_construct(active, 0.01)
    x, _, _ = orc.cg_solve(sysm["H9"], sysm["present"], sysm["P"], -sysm["g"], 200,
                           0.01 * np.linalg.norm(sysm["g"]), 1e-3)
    nodes0 = surf["nodes"].copy()
    new_active, count, _ = orc.update_and_reactivate(x, active)
    nodes1 = nodes0 + x.reshape(-1, 4) * surf["node_valid"][:, None]
    assert np.allclose(orc.nodes, nodes1, rtol=0, atol=0)      # Surface::update_nodes

    ps, npx, npy = 1 << surf["scale"], surf["npx"], surf["npy"]
    stride = npx + 1
    M = np.asarray(views["M"]).reshape(-1, 3, 3); t = np.asarray(views["t"]).reshape(-1, 3)
    want = np.zeros_like(active)
    margin = []
    for p in np.flatnonzero(surf["patch_valid"]):
        ix, iy = p % npx, p // npx
        ids = [iy * stride + ix, iy * stride + ix + 1, (iy + 1) * stride + ix, (iy + 1) * stride + ix + 1]
        if not active[ids].any():
            continue
        w0 = _patch_depth_at_pixels(nodes0[ids].reshape(16), ps)
        w1 = _patch_depth_at_pixels(nodes1[ids].reshape(16), ps)
        jj, ii = np.mgrid[0:ps, 0:ps]
        px = np.stack([surf["start_x"] + ix * ps + ii, surf["start_y"] + iy * ps + jj,
                       np.ones_like(ii)], -1).astype(float)
        moved = 0.0
        for s in range(len(views["subs"])):
            if not (surf["patch_vis"][p] >> s) & 1:
                continue
            ray = px @ M[s].T
            q0 = ray * w0[..., None] + t[s]
            q1 = ray * w1[..., None] + t[s]
            d = q0[..., :2] / q0[..., 2:] - q1[..., :2] / q1[..., 2:]
            moved = max(moved, np.sqrt((d ** 2).sum(-1)).max())
        margin.append(abs(moved - 0.15))
        if moved > 0.15:
            want[ids] = 1
    # decisions closer than 1e-9 px to the threshold could go either way under
    # a different summation order: none in this scene
    assert min(margin) > 1e-9
    assert np.array_equal(new_active, want)
    assert count == int(want.sum()) and 0 < count < int(active.sum())


# ------------------------------------------------ spherical harmonics basis
def test_sh_basis_spans_the_real_spherical_harmonics_of_bands_0_to_3(oracle):
    """The 16 functions of spherical_harmonics.h are (unnormalised) real
    spherical harmonics: on the unit sphere each one is a constant multiple of
    exactly one real Y_lm of scipy, l = 0..3, every (l, m) hit once."""
    sp = pytest.importorskip("scipy.special")
    rng = np.random.default_rng(2)
    v = rng.standard_normal((400, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    ours = np.stack([oracle.sh_evaluate_4_band(n) for n in v])           # [400][16]
    # the reference's axes are a permutation of the textbook's (its "z" term
    # 2 z^2 - x^2 - y^2 sits at index 6 with z = n[2]); the span is what matters
    theta = np.arccos(np.clip(v[:, 2], -1, 1)); phi = np.arctan2(v[:, 1], v[:, 0])
    real = []
    for l in range(4):
        for m in range(-l, l + 1):
            if hasattr(sp, "sph_harm_y"):
                y = sp.sph_harm_y(l, abs(m), theta, phi)
            else:
                y = sp.sph_harm(abs(m), l, phi, theta)
            real.append(np.sqrt(2) * y.imag if m < 0 else (y.real if m == 0 else np.sqrt(2) * y.real))
    real = np.stack(real, 1)                                              # [400][16]
    hit = set()
    for k in range(16):
        # correlation with every real harmonic: one of them is +-1, the rest 0
        c = np.array([abs(np.dot(ours[:, k], real[:, j])) / (np.linalg.norm(ours[:, k])
                      * np.linalg.norm(real[:, j])) for j in range(16)])
        j = int(np.argmax(c))
        assert c[j] > 1 - 1e-9, (k, c)
        hit.add(j)
    assert len(hit) == 16


# ------------------------------------------- image gradients of a StereoView
def test_image_gradients_are_the_least_squares_quadratic_fit(oracle):
    """stereo_view.cc:97-188 fits a quadratic to every 3 x 3 window; gradient
    and Hessian at the centre are the coefficients of numpy's least-squares
    fit of the same model (to float rounding of the stored planes)."""
    rng = np.random.default_rng(5)
    img = rng.random((12, 14)).astype(np.float32)
    g, hs = oracle.gradients_and_hessian(img)
    a, b = np.meshgrid([-1.0, 0.0, 1.0], [-1.0, 0.0, 1.0], indexing="ij")   # a = x offset, b = y offset
    A = np.stack([a.ravel() ** 2, b.ravel() ** 2, (a * b).ravel(), a.ravel(), b.ravel(),
                  np.ones(9)], 1)
    for (y, x) in [(1, 1), (5, 7), (10, 12), (3, 9)]:
        win = np.array([[img[y + bb, x + aa] for bb in (-1, 0, 1)] for aa in (-1, 0, 1)], dtype=float)
        cxx, cyy, cxy, cx, cy, _ = np.linalg.lstsq(A, win.ravel(), rcond=None)[0]
        assert np.allclose(g[y, x], [cx, cy], rtol=0, atol=2e-6)
        assert np.allclose(hs[y, x], [2 * cxx, cxy, 2 * cyy], rtol=0, atol=4e-6)
    # the one-pixel border carries no fit
    assert not g[0].any() and not g[:, 0].any() and not g[-1].any() and not g[:, -1].any()


# --------------------------------------------------------- normals of a surface
def test_normal_map_is_the_cross_product_of_the_surface_tangents(oracle):
    """Surface::get_normal_map (surface.cc:170-183, surface_patch.cc:30-55): with
    P(x, y) = w(x, y) (x / f, y / f, 1) the stored normal is P_x x P_y
    normalised, its first component negated (the reference's axis convention)
    -- checked with central differences of the depth map (1e-3: the
    differences straddle patch borders)."""
    from smvs_amd import synth
    prob = synth.make_problem(192, 128, 2, scale=4, noise=0.0, seed=3)
    orc = oracle.OracleProblem(prob["surf"], prob["views"])
    d = orc.depth_map().astype(np.float64)
    n = orc.normal_map().astype(np.float64)
    H, W = d.shape
    f = prob["views"]["flen"]
    ys, xs = np.mgrid[0:H, 0:W]
    P = np.stack([d * (xs + 0.5 - W / 2.0) / f, d * (ys + 0.5 - H / 2.0) / f, d], -1)
    Px = (P[1:-1, 2:] - P[1:-1, :-2]) / 2
    Py = (P[2:, 1:-1] - P[:-2, 1:-1]) / 2
    c = np.cross(Px, Py)
    ok = ((d[1:-1, 1:-1] > 0) & (d[1:-1, 2:] > 0) & (d[1:-1, :-2] > 0) & (d[2:, 1:-1] > 0)
          & (d[:-2, 1:-1] > 0))
    c = c[ok] / np.linalg.norm(c[ok], axis=-1, keepdims=True)
    err = np.abs(n[1:-1, 1:-1][ok] - c * np.array([-1.0, 1.0, 1.0]))
    assert ok.sum() > 4000
    assert np.median(err) < 2e-4 and np.percentile(err, 99) < 3e-3, (np.median(err), np.percentile(err, 99))
    assert np.abs(np.linalg.norm(n[1:-1, 1:-1][ok], axis=-1) - 1).max() < 1e-5


# ---------------------------- invariants the device kernels' data layouts rest on
def test_path_cost_minus_cost_fits_a_byte():
    """csrc/sgm.hip stores L - C as ONE BYTE per cell and direction when
    P2 <= 255: along every path 0 <= L - C <= P2 (the minimum in
    Hirschmueller's recurrence is over terms >= min L', one of them min L' + P2)."""
    rng = np.random.default_rng(3)
    cost = rng.integers(0, 256, size=(7, 9, 10)).astype(np.uint16)
    for p1, p2 in ((6, 96), (10, 255), (1, 1)):
        for L in _textbook_paths(cost, p1, p2):
            extra = L - cost.astype(np.int64)
            assert extra.min() >= 0 and extra.max() <= p2


def test_bilateral_colour_weight_depends_on_the_pair_almost_only_through_the_difference():
    """csrc/sgm.hip compresses the 256 x 256 colour weights
    expf(-(b/255 - a/255)^2 / (2 * 0.1^2)) (float arithmetic, the host's expf)
    to 511 differences x <= 4 candidates + a 2-bit selector; with another libm
    the device falls back to exponentials, this test says which one applies here."""
    import ctypes, ctypes.util
    libm = ctypes.CDLL(ctypes.util.find_library("m"))
    libm.expf.restype = ctypes.c_float
    libm.expf.argtypes = [ctypes.c_float]
    f = np.float32
    q = (np.arange(256, dtype=np.float32) / f(255.0)).astype(np.float32)
    diff = (q[None, :] - q[:, None]).astype(np.float32)           # [a][b] = b/255 - a/255
    arg = (-(diff * diff) / (f(2.0) * f(0.1) * f(0.1))).astype(np.float32)
    uniq = np.unique(arg)
    table = {float(v): libm.expf(float(v)) for v in uniq}
    wgt = np.vectorize(table.get, otypes=[np.float32])(arg)
    worst = 0
    for d in range(-255, 256):
        vals = np.unique(np.diagonal(wgt, offset=d))
        worst = max(worst, vals.size)
    assert worst <= 4, worst
    assert wgt[0, 0] == 1.0 and wgt[0, 255] > 0.0        # no weight underflows to zero


# ------------------------------------------------------ joint bilateral upsample
def test_bilateral_upsample_is_a_normalised_joint_bilateral_filter(oracle):
    """depth_optimizer.cc:957-1004: out(p) = sum_q d(q) g_s(q - p) prod_c g_c(I_c(q) - I_c(p))
    / sum_q (the same weights) over the (2k + 1)^2 window, d looked up in the
    low-resolution map (nearest, towards zero), zero depths skipped, borders
    clamped -- evaluated here in float64 with numpy (1e-5: the filter runs in float)."""
    rng = np.random.default_rng(8)
    h, w, ks, sigma = 18, 22, 3, 2.0
    dm = (2.0 + rng.random((h // 2, w // 2))).astype(np.float32)
    dm[rng.random(dm.shape) < 0.25] = 0.0
    ci = rng.integers(0, 256, size=(h, w, 3)).astype(np.float32) / np.float32(255.0)
    got = oracle.bilateral_upsample(dm, ci, sigma=sigma, kernel_size=ks)
    sx, sy = dm.shape[1] / w, dm.shape[0] / h
    want = np.zeros((h, w))
    for y in range(h):
        for x in range(w):
            qy = np.clip(y + np.arange(-ks, ks + 1), 0, h - 1)
            qx = np.clip(x + np.arange(-ks, ks + 1), 0, w - 1)
            QY, QX = np.meshgrid(qy, qx, indexing="ij")
            KY, KX = np.meshgrid(np.arange(-ks, ks + 1), np.arange(-ks, ks + 1), indexing="ij")
            d = dm[np.minimum((sy * QY).astype(int), dm.shape[0] - 1),
                   np.minimum((sx * QX).astype(int), dm.shape[1] - 1)].astype(float)
            ws = np.exp(-(KX ** 2 + KY ** 2) / (2 * sigma ** 2))
            wc = np.exp(-((ci[QY, QX].astype(float) - ci[y, x].astype(float)) ** 2).sum(-1) / (2 * 0.1 ** 2))
            wgt = ws * wc * (d != 0)
            want[y, x] = (wgt * d).sum() / wgt.sum() if wgt.sum() > 0 else 0.0
    assert np.array_equal(got == 0, want == 0)
    assert np.max(np.abs(got - want)) <= 1e-5 * np.max(np.abs(want))
    assert (got == 0).sum() < got.size // 2


# ---------------------------------------------------------------- lighting fit
def test_lighting_fit_is_the_least_squares_fit_of_the_image_to_the_sh_basis(oracle):
    """light_optimizer.cc:22-55: the 16 lighting parameters minimise
    sum_p (l . sh(n_p) - I_p)^2 over the pixels with a unit normal and an
    intensity of at least 0.05; the pseudo inverse of the normal equations
    gives what numpy's lstsq gives on the stacked rows."""
    rng = np.random.default_rng(6)
    n = rng.standard_normal((3000, 3)).astype(np.float32)
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    n[::17] = 0.0                                     # pixels without a surface
    truth = rng.standard_normal(16) * 0.2; truth[0] = 0.7
    sh = np.stack([oracle.sh_evaluate_4_band(v.astype(np.float64)) for v in n])
    img = (sh @ truth + 0.01 * rng.standard_normal(len(n))).astype(np.float32)
    img[5::23] = 0.01                                 # too dark: skipped
    A, b = oracle.light_accumulate(n, img)
    got = oracle.light_solve(A, b)
    use = (np.abs(np.linalg.norm(n.astype(np.float64), axis=1) - 1.0) <= 1e-6) & (img >= np.float32(0.05))
    assert 0.8 * len(n) < use.sum() < len(n)
    R = sh[use]
    assert np.allclose(A, R.T @ R, rtol=1e-12, atol=1e-9)
    assert np.allclose(b, R.T @ img[use].astype(np.float64), rtol=1e-12, atol=1e-9)
    want = np.linalg.lstsq(R, img[use].astype(np.float64), rcond=None)[0]
    assert np.allclose(got, want, rtol=1e-7, atol=1e-9)
    assert np.allclose(got, truth, atol=5e-3)         # and it recovers the lighting
