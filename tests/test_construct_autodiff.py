"""Mathematical pin of GaussNewtonStep::construct as a whole
(lib/gauss_newton_step.cc:145-518): on image planes that are exactly linear
(a quadratic image) the per-patch g and H of the oracle -- and of the device
-- must equal J^T W r and J^T W J with J obtained by AUTOMATIC
DIFFERENTIATION of an independently written residual model
(tests/analytic_model.py).  The reference tests its Jacobian pieces against
finite differences one by one (tests/gtest_correspondence.cc:286-493,
gtest_surface_deriv.cc:377-666); this is the same check on the assembled
result, photometric + pair terms + regulariser + shading.

Tolerance: the oracle / device sample the float32 planes with float
coordinates (linear_at), the model evaluates the linear function in double:
1e-7 relative on a tap (measured: 5e-8 on g, 1e-7 on H), 2e-6 is the bar written below (a wrong
sign, factor, index or missing term shows up at 1e-2 .. 1).
"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from analytic_model import PatchModel, analytic_problem  # noqa: E402

TOL = 2e-6

CASES = [  # scale, n_subs, shading, light_reg
    (2, 3, False, 0.0),      # photometric + pairs + regulariser, every pixel
    (3, 2, False, 0.0),      # sampling 2 (every other row / column)
    (2, 1, False, 0.0),      # single neighbour: no pair rows
    (2, 3, True, 0.0),       # shading term, geometric regulariser off (Q8)
    (2, 3, True, 0.5),       # shading + geometric regulariser scaled by light_reg / 100
]


def _rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300)


def _problem(scale, n_subs, shading):
    size = {2: (96, 72), 3: (128, 96)}[scale]
    return analytic_problem(size[0], size[1], n_subs, scale, shading, seed=100 * scale + n_subs)


def _pick(prob, count, seed=1):
    surf = prob["surf"]
    full = (1 << len(prob["views"]["subs"])) - 1
    # patches seen by every neighbour first (all rows present), then the others
    order = np.argsort(-(surf["patch_vis"] == full).astype(int), kind="stable")
    valid = [int(p) for p in order if surf["patch_valid"][p]]
    rng = np.random.default_rng(seed)
    head = valid[:max(len(valid) // 2, 1)]
    return [int(p) for p in rng.choice(head, size=min(count, len(head)), replace=False)]


def test_model_is_self_consistent():
    """J^T W r of the model is the gradient of sum c phi(r), phi(r) =
    |r| - eps ln(1 + |r| / eps): the IRLS weights 1 / (eps + |r|) are phi'(r) / r."""
    prob = _problem(2, 3, True)
    for p in _pick(prob, 2):
        m = PatchModel(prob["surf"], prob["views"], p, 0.01, 0.5, prob["lighting"],
                       prob["analytic"])
        g, _ = m.normal_equations()
        assert _rel(g, m.energy_gradient()) < 1e-12


@pytest.mark.parametrize("scale,n_subs,shading,light_reg", CASES)
def test_oracle_construct_matches_autodiff(oracle, scale, n_subs, shading, light_reg):
    prob = _problem(scale, n_subs, shading)
    lighting = prob["lighting"] if shading else None
    orc = oracle.OracleProblem(prob["surf"], prob["views"])
    worst_g = worst_H = 0.0
    for p in _pick(prob, 3):
        m = PatchModel(prob["surf"], prob["views"], p, 0.01, light_reg, lighting,
                       prob["analytic"])
        g_ad, H_ad = m.normal_equations()
        g_ref, H_ref = orc.gn_patch(p, 0.01, light_reg, lighting)
        assert np.linalg.norm(g_ad) > 0 and np.linalg.norm(H_ad) > 0
        worst_g = max(worst_g, _rel(g_ref, g_ad))
        worst_H = max(worst_H, _rel(np.triu(H_ref), np.triu(H_ad)))
    print("oracle vs autodiff: g %.2e  H %.2e" % (worst_g, worst_H))
    assert worst_g < TOL and worst_H < TOL, (worst_g, worst_H)


def test_autodiff_pin_is_not_vacuous(oracle):
    """The bar separates: dropping the pair rows, the regulariser or the
    depth-slope part of the warp Jacobian moves g / H by far more than TOL."""
    prob = _problem(2, 3, False)
    p = _pick(prob, 1)[0]
    orc = oracle.OracleProblem(prob["surf"], prob["views"])
    g_ref, H_ref = orc.gn_patch(p, 0.01)
    m = PatchModel(prob["surf"], prob["views"], p, 0.0, 0.0, None, prob["analytic"])
    g_noreg, H_noreg = m.normal_equations()
    assert _rel(np.triu(H_ref), np.triu(H_noreg)) > 100 * TOL
    views1 = dict(prob["views"])
    surf1 = dict(prob["surf"])
    surf1["patch_vis"] = prob["surf"]["patch_vis"] & np.uint32(1)   # one neighbour: no pairs
    m1 = PatchModel(surf1, views1, p, 0.01, 0.0, None, prob["analytic"])
    g1, H1 = m1.normal_equations()
    assert _rel(g_ref, g1) > 100 * TOL
    # the shading rows carry weight: the model without them misses the oracle
    probs = _problem(2, 3, True)
    ps = _pick(probs, 1)[0]
    orcs = oracle.OracleProblem(probs["surf"], probs["views"])
    g_s, H_s = orcs.gn_patch(ps, 0.01, 0.5, probs["lighting"])
    m2 = PatchModel(probs["surf"], probs["views"], ps, 0.01, 0.0, None, probs["analytic"])
    g2, H2 = m2.normal_equations()
    assert _rel(g_s, g2) > 100 * TOL and _rel(np.triu(H_s), np.triu(H2)) > 100 * TOL


@pytest.mark.gpu
@pytest.mark.parametrize("scale,n_subs,shading,light_reg", CASES)
def test_device_construct_matches_autodiff(scale, n_subs, shading, light_reg):
    """The same pin on the HIP path: gn_patch_kernel's per-patch systems
    against the autodiff model (no oracle involved)."""
    import smvs_amd
    if smvs_amd.device_count() < 1:
        pytest.fail("no HIP device visible: the GPU tests must run on a GPU")
    prob = _problem(scale, n_subs, shading)
    lighting = prob["lighting"] if shading else None
    surf = prob["surf"]
    ctx = smvs_amd.ViewContext(surf["width"], surf["height"], n_subs)
    ctx.set_views(prob["views"])
    ctx.set_surface(surf)
    ctx.gn_construct(0.01, light_reg, lighting)
    Hp, gp = ctx.gn_patch_systems()
    worst_g = worst_H = 0.0
    for p in _pick(prob, 4):
        m = PatchModel(surf, prob["views"], p, 0.01, light_reg, lighting, prob["analytic"])
        g_ad, H_ad = m.normal_equations()
        worst_g = max(worst_g, _rel(gp[p], g_ad))
        worst_H = max(worst_H, _rel(np.triu(Hp[p]), np.triu(H_ad)))
    ctx.close()
    print("device vs autodiff: g %.2e  H %.2e" % (worst_g, worst_H))
    assert worst_g < TOL and worst_H < TOL, (worst_g, worst_H)
