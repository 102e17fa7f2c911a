"""Independent statement of the energy GaussNewtonStep::construct linearises,
differentiated by automatic differentiation (torch, float64, CPU) -- test
infrastructure, like oracle/.

The reference builds, per surface patch, g = J^T W r and H = J^T W J from
hand-derived Jacobians (lib/gauss_newton_step.cc:145-518 with
lib/correspondence.cc:88-187, lib/surface_derivative.cc:31-190,
lib/bicubic_patch.cc:258-316).  This module writes down only the RESIDUALS --
from their definitions, with every derivative w.r.t. pixel position taken by
autograd instead of the reference's closed forms -- and lets autograd produce
J.  With image planes that are exactly linear (a quadratic image: bilinear
taps are exact, the Hessian plane is the gradient plane's derivative) the
reference's g and H must agree with J^T W r and J^T W J of this model up to
float32 tap rounding.  It is the reference's own test style
(tests/gtest_correspondence.cc:286-493, tests/gtest_surface_deriv.cc:377-666:
analytic derivative vs finite differences) applied to construct() as a whole,
and it does not share a line with oracle/smvs_oracle.c.

Definitions used (reference file:line of the quantity, not of a formula):
  surface    w(X, Y): bicubic Hermite patch, node order n00, n10, n01, n11,
             parameters (f, dx, dy, dxy) in patch units, local coordinate
             ((X - px0) + 0.5) / ps               surface_patch.cc:57-120
  warp       (a, b, d) = w * M (u, v, 1) + t, proj = (a / d, b / d) with
             u = X + 0.5, v = Y + 0.5; taps at proj - 0.5
                                                  gauss_newton_step.cc:183-198
  photometric residual  rho_j = d/d(X,Y) [ I_j(proj_j(X, Y)) ] - grad I_main
             = (dproj/d(X,Y))^T G_j(proj_j - 0.5) - G_main[X, Y]; rows: rho_j
             for every neighbour, rho_j - rho_j2 for every pair j < j2
                                                  :266-333
  normal     n = normalize(w_X, -w_Y, (x w_X + y w_Y + w) / f), x, y centred
             pixel coordinates                    surface_derivative.cc:17-28
  regulariser rows: the six pixel derivatives of n in the order
             (n0_X, -(-n1)_X ... ) = (xx, -yx, zx, xy, -yy, zy) i.e.
             d n / dX then d n / dY               surface_derivative.cc:69-107
  shading    s = sum_l p_l SH_l(n); residual = grad_XY(s) / s - grad I_lin / I_lin
             with the SH Jacobian frozen when grad_XY(s) is differentiated
             w.r.t. the nodes                     gauss_newton_step.cc:420-517
  weights    IRLS 1 / (1e-4 + |r|) times: 1 (photometric);
             reg * 0.005 / max(0.03, |G_main|_1) * S(S+1)/2 * geom
             (regulariser, geom = 1 or light_reg / 100 with lighting);
             0.001 * S(S+1)/2 / (1e-4 + |grad I_lin|_1) (shading)
                                                  :212-214, 385-443
"""
import numpy as np
import torch

R_FACTOR = 1e-4


def _hermite(t):
    t2, t3 = t * t, t * t * t
    return (1.0 - 3.0 * t2 + 2.0 * t3,      # value at node 0
            3.0 * t2 - 2.0 * t3,            # value at node 1
            t - 2.0 * t2 + t3,              # slope at node 0
            -t2 + t3)                       # slope at node 1


def surface_w(theta, X, Y, px0, py0, ps):
    """theta: (4 nodes, 4 params); X, Y pixel coordinates (continuous)."""
    x = (X - px0 + 0.5) / ps
    y = (Y - py0 + 0.5) / ps
    hx, hy = _hermite(x), _hermite(y)
    w = 0.0
    for node in range(4):
        a, b = node & 1, node >> 1
        f, dx, dy, dxy = theta[node]
        w = w + f * hx[a] * hy[b] + dx * hx[2 + a] * hy[b] \
            + dy * hx[a] * hy[2 + b] + dxy * hx[2 + a] * hy[2 + b]
    return w


def _dX(f, X):
    return torch.autograd.grad(f.sum(), X, create_graph=True)[0]


def sh_basis(n):
    """scaled real SH, 16 values (spherical_harmonics.h:60-151)."""
    x, y, z = n[..., 0], n[..., 1], n[..., 2]
    x2, y2, z2 = x * x, y * y, z * z
    one = torch.ones_like(x)
    return torch.stack([
        one, y, z, x,
        x * y, y * z, -x2 - y2 + 2.0 * z2, x * z, x2 - y2,
        (3.0 * x2 - y2) * y, x * y * z, (4.0 * z2 - x2 - y2) * y,
        (2.0 * z2 - 3.0 * x2 - 3.0 * y2) * z, (4.0 * z2 - x2 - y2) * x,
        (x2 - y2) * z, (x2 - 3.0 * y2) * x], dim=-1)


class PatchModel:
    """Residuals of one patch as a function of its 16 node parameters."""

    def __init__(self, surf, views, patch, reg, light_reg=0.0, lighting=None,
                 linear_planes=None):
        """linear_planes[j] = (g0 (2,), A (2, 2)): G_j(x, y) = g0 + A (x, y)
        -- the analytic form of views['subs'][j] (the model never samples the
        float planes of the neighbours)."""
        self.ps = 1 << surf["scale"]
        scale = surf["scale"]
        self.sampling = 4 if scale >= 5 else (2 if scale >= 3 else 1)
        npx = surf["npx"]
        self.ix, self.iy = patch % npx, patch // npx
        self.px0 = surf["start_x"] + self.ix * self.ps
        self.py0 = surf["start_y"] + self.iy * self.ps
        n00 = self.iy * (npx + 1) + self.ix
        ids = [n00, n00 + 1, n00 + npx + 1, n00 + npx + 2]
        self.theta0 = torch.tensor(np.asarray(surf["nodes"], float).reshape(-1, 4)[ids])
        vis = int(surf["patch_vis"][patch])
        self.subs = [j for j in range(len(views["subs"])) if (vis >> j) & 1]
        self.M = [torch.tensor(np.asarray(views["M"], float).reshape(-1, 3, 3)[j])
                  for j in self.subs]
        self.t = [torch.tensor(np.asarray(views["t"], float).reshape(-1, 3)[j])
                  for j in self.subs]
        self.planes = [(torch.tensor(np.asarray(linear_planes[j][0], float)),
                        torch.tensor(np.asarray(linear_planes[j][1], float)))
                       for j in self.subs]
        # sampled pixels: every sampling-th row and column (surface_patch.cc:111-119)
        loc = np.arange(0, self.ps, self.sampling)
        jj, ii = np.meshgrid(loc, loc, indexing="ij")
        self.Xi = (self.px0 + ii.reshape(-1)).astype(int)
        self.Yi = (self.py0 + jj.reshape(-1)).astype(int)
        self.W, self.H = surf["width"], surf["height"]
        self.flen = float(np.float32(views["flen"]))
        self.inv_flen = float(np.float32(views["inv_flen"]))
        self.G_main = torch.tensor(np.asarray(views["grad"], float)[self.Yi, self.Xi])
        self.reg, self.light_reg = reg, light_reg
        self.lighting = None if lighting is None else torch.tensor(np.asarray(lighting, float))
        if lighting is not None:
            self.I_lin = torch.tensor(np.asarray(views["shading"], float)[self.Yi, self.Xi])
            self.G_lin = torch.tensor(np.asarray(views["shading_grad"], float)[self.Yi, self.Xi])

    # ------------------------------------------------------------------
    def residuals(self, theta):
        """-> (r, c): residual rows and their constant weight factors."""
        X = torch.tensor(self.Xi, dtype=torch.float64, requires_grad=True)
        Y = torch.tensor(self.Yi, dtype=torch.float64, requires_grad=True)
        w = surface_w(theta, X, Y, self.px0, self.py0, self.ps)
        rows, consts = [], []
        S = len(self.subs)
        num_diffs = S * (S + 1) // 2
        rho = []
        for M, t, (g0, A) in zip(self.M, self.t, self.planes):
            u, v = X + 0.5, Y + 0.5
            p = M[0, 0] * u + M[0, 1] * v + M[0, 2]
            q = M[1, 0] * u + M[1, 1] * v + M[1, 2]
            r = M[2, 0] * u + M[2, 1] * v + M[2, 2]
            a, b, d = w * p + t[0], w * q + t[1], w * r + t[2]
            prx, pry = a / d, b / d
            # gradient plane of the neighbour at the tap position
            sx, sy = prx - 0.5, pry - 0.5
            Gx = g0[0] + A[0, 0] * sx + A[0, 1] * sy
            Gy = g0[1] + A[1, 0] * sx + A[1, 1] * sy
            # chain rule through the warp (total derivative: w depends on X, Y)
            rho_x = _dX(prx, X) * Gx + _dX(pry, X) * Gy
            rho_y = _dX(prx, Y) * Gx + _dX(pry, Y) * Gy
            rho.append(torch.stack([rho_x, rho_y], dim=-1))
        for j in range(S):
            rows.append(rho[j] - self.G_main)
            consts.append(torch.ones_like(rows[-1]))
            for j2 in range(j + 1, S):
                rows.append(rho[j] - rho[j2])
                consts.append(torch.ones_like(rows[-1]))
        if self.reg > 0.0:
            x = X + 0.5 - self.W / 2.0
            y = Y + 0.5 - self.H / 2.0
            wX, wY = _dX(w, X), _dX(w, Y)

            def normal(scale_z):
                nv = torch.stack([wX, -wY, (x * wX + y * wY + w) * scale_z], dim=-1)
                return nv / nv.norm(dim=-1, keepdim=True)
            n = normal(1.0 / self.flen)
            div = torch.stack([_dX(n[:, 0], X), _dX(n[:, 1], X), _dX(n[:, 2], X),
                               _dX(n[:, 0], Y), _dX(n[:, 1], Y), _dX(n[:, 2], Y)], dim=-1)
            bw = self.reg * 0.005 / torch.clamp(self.G_main.abs().sum(-1), min=0.03) * num_diffs
            if self.lighting is None or self.light_reg > 0.0:
                geom = 1.0 if self.lighting is None else self.light_reg / 100.0
                rows.append(div)
                consts.append((bw * geom)[:, None].expand(-1, 6))
            if self.lighting is not None:
                p = self.lighting
                n_s = normal(self.inv_flen)
                sh = sh_basis(n_s)
                shading = (sh * p).sum(-1)
                # SH Jacobian at the current normal, frozen (:449-492)
                nd = n_s.detach().requires_grad_(True)
                D = torch.stack([torch.autograd.grad(sh_basis(nd)[:, l].sum(), nd)[0]
                                 for l in range(16)], dim=1)          # (P, 16, 3)
                pD = (D[:, 1:, :] * p[1:, None]).sum(1)               # (P, 3), l >= 1
                sgrad = torch.stack([(pD * div[:, 0:3]).sum(-1),
                                     (pD * div[:, 3:6]).sum(-1)], dim=-1)
                err = sgrad / shading[:, None] - self.G_lin / self.I_lin[:, None]
                sw = 0.001 * num_diffs / (R_FACTOR + self.G_lin.abs().sum(-1))
                skip = (self.G_lin.norm(dim=-1) < 1e-10) | (shading.detach() ** 2 < 1e-10) \
                    | (self.I_lin ** 2 < 1e-10)
                sw = torch.where(skip, torch.zeros_like(sw), sw)
                rows.append(err)
                consts.append(sw[:, None].expand(-1, 2))
        r = torch.cat([x.reshape(-1) for x in rows])
        c = torch.cat([x.reshape(-1) for x in consts])
        return r, c

    def normal_equations(self):
        """g (16) and H (16 x 16) of the Gauss-Newton / IRLS linearisation."""
        theta = self.theta0.clone().requires_grad_(True)
        r, c = self.residuals(theta)
        J = torch.stack([torch.autograd.grad(r[k], theta, retain_graph=True)[0].reshape(16)
                         for k in range(r.numel())])
        r, c = r.detach(), c.detach()
        Wt = c / (R_FACTOR + r.abs())
        g = J.T @ (Wt * r)
        H = J.T @ (Wt[:, None] * J)
        return g.numpy(), H.numpy()

    def energy_gradient(self):
        """grad of sum c * phi(r), phi(r) = |r| - eps ln(1 + |r| / eps): equals
        g wherever the residual Jacobian is exact (everything but the frozen
        SH Jacobian of the shading term)."""
        theta = self.theta0.clone().requires_grad_(True)
        r, c = self.residuals(theta)
        E = (c * (r.abs() - R_FACTOR * torch.log1p(r.abs() / R_FACTOR))).sum()
        return torch.autograd.grad(E, theta)[0].reshape(16).numpy()


def linear_planes(rng, n_subs, width, height):
    """Gradient / Hessian planes of quadratic images, exactly representable in
    float32: G(x, y) = g0 + A (x, y), A symmetric, all values multiples of
    2^-20 below 2^4."""
    planes, analytic = [], []
    xs, ys = np.meshgrid(np.arange(width, dtype=np.float64),
                         np.arange(height, dtype=np.float64))
    for _ in range(n_subs):
        A = rng.integers(-40, 41, size=(2, 2)).astype(np.float64) / 2 ** 14
        A[1, 0] = A[0, 1]
        g0 = rng.integers(-3000, 3001, size=2).astype(np.float64) / 2 ** 16
        grad = np.stack([g0[0] + A[0, 0] * xs + A[0, 1] * ys,
                         g0[1] + A[1, 0] * xs + A[1, 1] * ys], axis=-1)
        g32 = grad.astype(np.float32)
        assert np.array_equal(g32.astype(np.float64), grad)   # exact in float32
        hess = np.empty((height, width, 3), np.float32)
        hess[..., 0], hess[..., 1], hess[..., 2] = A[0, 0], A[0, 1], A[1, 1]
        planes.append((g32, hess))
        analytic.append((g0, A))
    return planes, analytic


def analytic_problem(width, height, n_subs, scale, shading, seed):
    """A synth.make_problem geometry (cameras, noisy surface, visibility) with
    its image planes replaced by analytic ones."""
    from smvs_amd import synth
    prob = synth.make_problem(width, height, n_subs, scale, shading=shading, noise=0.01)
    rng = np.random.default_rng(seed)
    views = dict(prob["views"])
    sizes = [(g.shape[1], g.shape[0]) for g, _ in views["subs"]]
    assert all(s == sizes[0] for s in sizes)
    planes, analytic = linear_planes(rng, n_subs, sizes[0][0], sizes[0][1])
    views["subs"] = planes
    views["grad"] = (rng.integers(-4000, 4001, size=(height, width, 2)) / 2 ** 16).astype(np.float32)
    if shading:
        views["shading"] = (0.3 + 0.6 * rng.random((height, width))).astype(np.float32)
        views["shading_grad"] = (rng.integers(-3000, 3001, size=(height, width, 2))
                                 / 2 ** 16).astype(np.float32)
    surf = dict(prob["surf"])
    nodes = np.array(surf["nodes"], float).reshape(-1, 4).copy()
    # curvature and mixed terms so that every basis function matters
    nodes[:, 1:3] += 0.02 * rng.standard_normal((nodes.shape[0], 2))
    nodes[:, 3] = 0.01 * rng.standard_normal(nodes.shape[0])
    surf["nodes"] = nodes
    return dict(surf=surf, views=views, lighting=prob["lighting"], analytic=analytic)
