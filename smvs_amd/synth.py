"""Deterministic synthetic multi-view scenes (test / bench tooling).

Produces exactly what the hot path consumes: per-view float planes in MVE
layout (gradient 2 ch, Hessian 3 ch, optional shading image + gradient), the
float reprojections (M_i, t_i) widened to double (reference:
lib/depth_optimizer.cc:679-699), and a bicubic node grid initialised from the
analytic depth.  The steps *before* the path (Gaussian scale blur and the
3x3 quadratic-fit gradient/Hessian of lib/stereo_view.cc:24-46, 97-188) are
done here in numpy with the semantics SURVEY.md Appendix A assumes for MVE.
"""
import numpy as np


# ------------------------------------------------------------------ cameras
class Camera:
    """MVE-style pinhole: x_cam = R X + t, flen normalised by max(w, h)."""

    def __init__(self, R, t, flen, width, height):
        self.R = np.asarray(R, dtype=np.float64)
        self.t = np.asarray(t, dtype=np.float64)
        self.flen = float(flen)
        self.width, self.height = int(width), int(height)

    def K(self, dtype=np.float32):
        w, h = self.width, self.height
        a = dtype(self.flen) * dtype(max(w, h))
        return np.array([[a, 0, dtype(w) * dtype(0.5)],
                         [0, a, dtype(h) * dtype(0.5)],
                         [0, 0, 1]], dtype=dtype)

    def Kinv(self, dtype=np.float32):
        w, h = self.width, self.height
        a = dtype(self.flen) * dtype(max(w, h))
        return np.array([[1 / a, 0, -dtype(w) * dtype(0.5) / a],
                         [0, 1 / a, -dtype(h) * dtype(0.5) / a],
                         [0, 0, 1]], dtype=dtype)

    @property
    def center(self):
        return -self.R.T @ self.t


def look_at(center, target, up=(0.0, -1.0, 0.0)):
    """World-to-camera rotation / translation with +z towards target,
    +x right, +y down (image convention)."""
    center = np.asarray(center, float); target = np.asarray(target, float)
    z = target - center; z /= np.linalg.norm(z)
    x = np.cross(-np.asarray(up, float), z); x /= np.linalg.norm(x)
    y = np.cross(z, x)
    R = np.stack([x, y, z])
    return R, -R @ center


def reprojection(main, sub):
    """CameraInfo::fill_reprojection in float32 [MVE-unverified], widened to
    double as depth_optimizer.cc:694-697 does."""
    f = np.float32
    Ks = main.Kinv(f); Kd = sub.K(f)
    Rs = main.R.astype(f); Rd = sub.R.astype(f)
    ts = main.t.astype(f); td = sub.t.astype(f)
    Rrel = (Rd @ Rs.T).astype(f)
    M = (Kd @ Rrel @ Ks).astype(f)
    t = (Kd @ (td - Rrel @ ts)).astype(f)
    return M.astype(np.float64).reshape(9), t.astype(np.float64)


# -------------------------------------------------------------------- scene
class SphereScene:
    """Unit sphere at distance `dist` in front of a background plane."""

    def __init__(self, dist=4.0, radius=1.0, plane_z=8.0, seed=2000,
                 px_size=0.004):
        """px_size: world size of one pixel on the sphere front; the albedo
        is a band-limited sum of sinusoids with wavelengths of 8..160 pixels
        so every view resolves the same texture."""
        self.c = np.array([0.0, 0.0, dist])
        self.r = radius
        self.plane_z = plane_z
        rng = np.random.default_rng(seed)
        n = 24
        lam = px_size * np.exp(rng.uniform(np.log(8.0), np.log(160.0), size=n))
        dirs = rng.standard_normal((n, 3))
        dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
        self.freq = dirs * (2 * np.pi / lam)[:, None]
        self.phase = rng.uniform(0, 2 * np.pi, size=n)
        self.amp = rng.uniform(0.4, 1.0, size=n)

    def intersect(self, origin, dirs):
        """Ray origin (3,), dirs (...,3) -> world points (...,3), hit mask of
        the sphere."""
        oc = origin - self.c
        a = np.sum(dirs * dirs, axis=-1)
        b = 2.0 * np.sum(dirs * oc, axis=-1)
        c = np.dot(oc, oc) - self.r ** 2
        disc = b * b - 4 * a * c
        hit = disc > 0
        s = np.where(hit, (-b - np.sqrt(np.where(hit, disc, 0.0))) / (2 * a), np.inf)
        hit &= s > 0
        sp = (self.plane_z - origin[2]) / dirs[..., 2]
        s = np.where(hit, s, sp)
        return origin + s[..., None] * dirs, hit

    def albedo(self, X):
        v = np.zeros(X.shape[:-1])
        for k in range(len(self.amp)):
            v += self.amp[k] * np.sin(X @ self.freq[k] + self.phase[k])
        v = v / np.sqrt(np.sum(self.amp ** 2))
        return np.clip(0.5 + 0.22 * v, 0.02, 0.98)

    def normal(self, X, hit):
        n = np.where(hit[..., None], (X - self.c) / self.r, np.array([0.0, 0.0, -1.0]))
        return n


def pixel_rays(cam, xs, ys):
    """World-space ray directions through continuous pixel coords (xs, ys)."""
    Kinv = cam.Kinv(np.float64)
    d_cam = np.stack([Kinv[0, 0] * xs + Kinv[0, 2],
                      Kinv[1, 1] * ys + Kinv[1, 2], np.ones_like(xs)], axis=-1)
    return d_cam @ cam.R  # R^T applied to rows


def depth_at(scene, cam, xs, ys):
    """z-depth in camera `cam` at continuous pixel coordinates."""
    dirs = pixel_rays(cam, xs, ys)
    X, _ = scene.intersect(cam.center, dirs)
    return (X @ cam.R.T + cam.t)[..., 2]


def depth_and_normal_maps(scene, cams):
    """Per camera: the depth map in MVE's convention (ray length, what
    StereoView::write_depth_to_view stores) and the normal map in the
    optimiser's camera-space convention (surface_derivative.cc:17-28: a
    fronto-parallel surface has normal (0, 0, 1); mesh_generator.cc:199-207
    turns (n0, -n1, -n2) into the world-space normal).  float32."""
    depths, normals = [], []
    for cam in cams:
        ys, xs = np.mgrid[0:cam.height, 0:cam.width].astype(np.float64)
        dirs = pixel_rays(cam, xs + 0.5, ys + 0.5)
        X, hit = scene.intersect(cam.center, dirs)
        depths.append(np.linalg.norm(X - cam.center, axis=-1).astype(np.float32))
        c = scene.normal(X, hit) @ cam.R.T            # outward normal, camera space
        normals.append(np.stack([c[..., 0], -c[..., 1], -c[..., 2]], axis=-1)
                       .astype(np.float32))
    return depths, normals


def render(scene, cam, lighting=None):
    """u8 grey image (H, W); with `lighting` (16 SH coeffs) Lambert-like
    shading through the reference's scaled SH basis."""
    h, w = cam.height, cam.width
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)
    dirs = pixel_rays(cam, xs + 0.5, ys + 0.5)
    X, hit = scene.intersect(cam.center, dirs)
    img = scene.albedo(X)
    if lighting is not None:
        n_world = scene.normal(X, hit)
        n = n_world @ cam.R.T
        n = n * np.array([1.0, -1.0, -1.0])  # towards the camera, y up
        img = img * np.clip(sh_shading(n, lighting), 0.05, None)
        img = np.clip(img, 0.0, 1.0)
    return np.clip(np.floor(img * 255.0 + 0.5), 0, 255).astype(np.uint8)


def sh_shading(n, lp):
    x, y, z = n[..., 0], n[..., 1], n[..., 2]
    x2, y2, z2 = x * x, y * y, z * z
    sh = [np.ones_like(x), y, z, x, x * y, y * z, -x2 - y2 + 2 * z2, x * z,
          x2 - y2, (3 * x2 - y2) * y, x * y * z, (4 * z2 - x2 - y2) * y,
          (2 * z2 - 3 * x2 - 3 * y2) * z, (4 * z2 - x2 - y2) * x,
          (x2 - y2) * z, (x2 - 3 * y2) * x]
    out = np.zeros_like(x)
    for l in range(16):
        out += lp[l] * sh[l]
    return out


def ring_cameras(width, height, n_subs, flen=1.2, ring=0.25, target=(0, 0, 4.0)):
    main = Camera(*look_at((0, 0, 0), target), flen, width, height)
    subs = []
    for k in range(n_subs):
        ang = 2 * np.pi * k / n_subs + 0.3
        c = (ring * np.cos(ang), ring * np.sin(ang), 0.0)
        subs.append(Camera(*look_at(c, target), flen, width, height))
    return main, subs


# ----------------------------------------------------- pre-path image planes
def blur_gaussian(img, sigma):
    """mve::image::blur_gaussian semantics per SURVEY Appendix A
    [MVE-unverified]: separable, half width ceil(2.884 sigma), clamped."""
    img = img.astype(np.float32)
    if sigma < 0.1:
        return img.copy()
    ks = int(np.ceil(2.884 * sigma))
    k = np.exp(-(np.arange(-ks, ks + 1, dtype=np.float32) ** 2)
               / np.float32(2 * sigma * sigma)).astype(np.float32)
    k /= k.sum()

    def conv(a, axis):
        out = np.zeros_like(a)
        n = a.shape[axis]
        idx = np.arange(n)
        for i, wgt in enumerate(k):
            src = np.clip(idx + i - ks, 0, n - 1)
            out += wgt * np.take(a, src, axis=axis)
        return out
    return conv(conv(img, 1), 0)


def _quadfit_matrix():
    rows = []
    for a in (-1, 0, 1):
        for b in (-1, 0, 1):
            rows.append([a * a, b * b, a * b, a, b, 1.0])
    return np.linalg.pinv(np.array(rows))  # 6 x 9, window order a outer b inner


_QF = _quadfit_matrix()


def gradients_and_hessian(img):
    """StereoView::compute_gradients_and_hessian (stereo_view.cc:97-188):
    3x3 quadratic fit; grad = (I_x, I_y), hess = (I_xx, I_xy, I_yy); the
    1-pixel border stays 0."""
    img = img.astype(np.float64)
    h, w = img.shape
    win = []
    for a in (-1, 0, 1):
        for b in (-1, 0, 1):
            win.append(img[1 + b:h - 1 + b, 1 + a:w - 1 + a])
    win = np.stack(win, axis=-1)
    r = win @ _QF.T
    grad = np.zeros((h, w, 2), np.float32)
    hess = np.zeros((h, w, 3), np.float32)
    grad[1:-1, 1:-1, 0] = r[..., 3]
    grad[1:-1, 1:-1, 1] = r[..., 4]
    hess[1:-1, 1:-1, 0] = 2.0 * r[..., 0]
    hess[1:-1, 1:-1, 1] = r[..., 2]
    hess[1:-1, 1:-1, 2] = 2.0 * r[..., 1]
    return grad, hess


def scale_planes(byte_img, scale):
    """What StereoView::set_scale leaves behind (stereo_view.cc:24-46)."""
    img = byte_img.astype(np.float32) / np.float32(255.0)
    sigma = 0.12 * 2.0 ** scale + 0.2
    return gradients_and_hessian(blur_gaussian(img, sigma))


def shading_planes(byte_img):
    """StereoView::initialize_linear without gamma (stereo_view.cc:64-84)."""
    shading = byte_img.astype(np.float32) / np.float32(255.0)
    grad, _ = gradients_and_hessian(shading)
    return shading, grad


# ------------------------------------------------------------------ surface
def grid_for_scale(width, height, scale):
    """Patch grid as Surface::Surface builds it (surface.cc:28-37)."""
    ps = 1 << scale
    npx = (width - 2) // ps - 1
    npy = (height - 2) // ps - 1
    sx = (width - npx * ps) // 2
    sy = (height - npy * ps) // 2
    return dict(scale=scale, npx=npx, npy=npy, start_x=sx, start_y=sy)


def surface_from_depth(scene, main, subs, scale, noise=0.0, seed=7,
                       border=0.03):
    """Node grid initialised from the analytic depth (+ optional noise on f),
    with every patch marked visible in the sub views whose reprojection of
    the patch corners stays inside the `border` margin."""
    W, H = main.width, main.height
    g = grid_for_scale(W, H, scale)
    ps = 1 << scale
    npx, npy = g["npx"], g["npy"]
    ix = np.arange(npx + 1); iy = np.arange(npy + 1)
    X, Y = np.meshgrid(g["start_x"] + ix * ps, g["start_y"] + iy * ps)
    X = X.astype(float); Y = Y.astype(float)
    e = 0.5
    f = depth_at(scene, main, X, Y)
    fx = (depth_at(scene, main, X + e, Y) - depth_at(scene, main, X - e, Y)) / (2 * e)
    fy = (depth_at(scene, main, X, Y + e) - depth_at(scene, main, X, Y - e)) / (2 * e)
    nodes = np.zeros((npy + 1, npx + 1, 4))
    nodes[..., 0] = f
    nodes[..., 1] = np.clip(fx * ps, -0.5, 0.5)
    nodes[..., 2] = np.clip(fy * ps, -0.5, 0.5)
    if noise > 0:
        rng = np.random.default_rng(seed)
        nodes[..., 0] *= 1.0 + noise * rng.standard_normal(f.shape)
    nodes = nodes.reshape(-1, 4)

    Ms, ts = zip(*[reprojection(main, s) for s in subs])
    px = g["start_x"] + np.arange(npx) * ps
    py = g["start_y"] + np.arange(npy) * ps
    PX, PY = np.meshgrid(px, py)
    vis = np.zeros((npy, npx), dtype=np.uint32)
    corners = [(0, 0), (ps, 0), (0, ps), (ps, ps)]
    fgrid = f
    for j, (M, t) in enumerate(zip(Ms, ts)):
        M = M.reshape(3, 3)
        ok = np.ones((npy, npx), bool)
        for k, (cx, cy) in enumerate(corners):
            d = fgrid[(cy // ps):(cy // ps) + npy, (cx // ps):(cx // ps) + npx]
            u = PX + cx + 0.0; v = PY + cy + 0.0
            a = d * (M[0, 0] * u + M[0, 1] * v + M[0, 2]) + t[0]
            b = d * (M[1, 0] * u + M[1, 1] * v + M[1, 2]) + t[1]
            c = d * (M[2, 0] * u + M[2, 1] * v + M[2, 2]) + t[2]
            qx, qy = a / c, b / c
            cut = border * max(subs[j].width, subs[j].height)
            ok &= (qx >= cut) & (qx < subs[j].width - cut)
            ok &= (qy >= cut) & (qy < subs[j].height - cut)
        vis |= (ok.astype(np.uint32) << np.uint32(j))
    # emulate DepthOptimizer::cut_boundaries (depth_optimizer.cc:360-400):
    # drop patches that straddle (or touch) the sphere silhouette
    dirs = pixel_rays(main, X, Y)
    _, hit = scene.intersect(main.center, dirs)
    mixed = np.zeros((npy, npx), bool)
    h00, h10, h01, h11 = hit[:-1, :-1], hit[:-1, 1:], hit[1:, :-1], hit[1:, 1:]
    mixed = ~((h00 == h10) & (h00 == h01) & (h00 == h11))
    grown = mixed.copy()
    grown[1:, :] |= mixed[:-1, :]; grown[:-1, :] |= mixed[1:, :]
    grown[:, 1:] |= mixed[:, :-1]; grown[:, :-1] |= mixed[:, 1:]
    vis[grown] = 0
    vis = vis.reshape(-1)
    patch_valid = (vis != 0).astype(np.uint8)
    node_valid = np.zeros((npy + 1, npx + 1), np.uint8)
    pv = patch_valid.reshape(npy, npx)
    node_valid[:-1, :-1] |= pv; node_valid[:-1, 1:] |= pv
    node_valid[1:, :-1] |= pv; node_valid[1:, 1:] |= pv
    surf = dict(g)
    surf.update(width=W, height=H, nodes=nodes,
                node_valid=node_valid.reshape(-1), patch_valid=patch_valid,
                patch_vis=vis)
    return surf


def make_problem(width, height, n_subs, scale, shading=False, noise=0.002,
                 seed=2000, lighting=None):
    """Everything one GN step needs, as plain numpy (both the HIP path and
    the oracle are fed from this dict)."""
    main, subs = ring_cameras(width, height, n_subs)
    scene = SphereScene(seed=seed, px_size=3.0 / (main.flen * max(width, height)))
    if shading and lighting is None:
        rng = np.random.default_rng(3000)
        lighting = np.zeros(16)
        lighting[0] = 0.8
        lighting[1:4] = rng.uniform(-0.25, 0.25, 3)
        lighting[4:] = rng.uniform(-0.05, 0.05, 12)
    imgs = [render(scene, main, lighting if shading else None)]
    imgs += [render(scene, s, lighting if shading else None) for s in subs]
    planes = [scale_planes(im, scale) for im in imgs]
    Ms, ts = zip(*[reprojection(main, s) for s in subs])
    K = main.K(np.float32)
    views = dict(flen=float(K[0, 0]), inv_flen=float(main.Kinv(np.float32)[0, 0]),
                 grad=planes[0][0], subs=[(g, h) for g, h in planes[1:]],
                 M=np.array(Ms), t=np.array(ts), shading=None, shading_grad=None)
    if shading:
        sh_img, sh_grad = shading_planes(imgs[0])
        views["shading"] = sh_img
        views["shading_grad"] = sh_grad
    surf = surface_from_depth(scene, main, subs, scale, noise=noise)
    return dict(surf=surf, views=views, lighting=lighting, images=imgs,
                scene=scene, main=main, sub_cams=subs)


# ------------------------------------------------- whole-pipeline scene inputs
class PlaneScene:
    """Slanted textured plane z = z0 + ax x + ay y (world = reference camera
    frame); same interface as SphereScene."""

    def __init__(self, z0=5.0, ax=0.12, ay=0.08, seed=1234, px_size=0.008):
        self.z0, self.ax, self.ay = z0, ax, ay
        rng = np.random.default_rng(seed)
        n = 24
        lam = px_size * np.exp(rng.uniform(np.log(8.0), np.log(160.0), size=n))
        dirs = rng.standard_normal((n, 3))
        dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
        self.freq = dirs * (2 * np.pi / lam)[:, None]
        self.phase = rng.uniform(0, 2 * np.pi, size=n)
        self.amp = rng.uniform(0.4, 1.0, size=n)

    def intersect(self, origin, dirs):
        # z = z0 + ax x + ay y  with  X = o + s d
        nrm = np.array([-self.ax, -self.ay, 1.0])
        s = (self.z0 - origin @ nrm) / (dirs @ nrm)
        return origin + s[..., None] * dirs, np.ones(dirs.shape[:-1], bool)

    def albedo(self, X):
        v = np.zeros(X.shape[:-1])
        for k in range(len(self.amp)):
            v += self.amp[k] * np.sin(X @ self.freq[k] + self.phase[k])
        v = v / np.sqrt(np.sum(self.amp ** 2))
        return np.clip(0.5 + 0.22 * v, 0.02, 0.98)

    def normal(self, X, hit):
        n = np.array([self.ax, self.ay, -1.0])
        n = n / np.linalg.norm(n)
        return np.broadcast_to(n, X.shape).copy()


def render_rgb(scene, cam, lighting=None):
    """u8 RGB image (H, W, 3): the grey rendering with a fixed tint so the
    three channels differ (ncc_for_patch reads channels 0..2)."""
    g = render(scene, cam, lighting).astype(np.float32)
    rgb = np.stack([g * 0.95, g, g * 0.9], axis=-1)
    return np.clip(np.floor(rgb + 0.5), 0, 255).astype(np.uint8)


def pipeline_inputs(kind, width, height, n_subs, flen=1.0, n_features=2000,
                    seed=1234, lighting=None):
    """Images + cameras + sparse features for a whole DepthOptimizer::optimize
    run (BASELINE.json configs[0] for kind='plane')."""
    rng = np.random.default_rng(seed)
    if kind == "plane":
        scene = PlaneScene(seed=seed, px_size=5.0 / (flen * max(width, height)))
        main = Camera(np.eye(3), np.zeros(3), flen, width, height)
        subs = []
        for k in range(n_subs):
            sign = 1.0 if k % 2 == 0 else -1.0
            ang = 0.02 * sign * (1 + k // 2)
            R = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0],
                          [-np.sin(ang), 0, np.cos(ang)]])
            subs.append(Camera(R, np.array([0.3 * sign * (1 + k // 2), 0.05 * k, 0.0]),
                               flen, width, height))
    else:
        main, subs = ring_cameras(width, height, n_subs, flen=flen)
        scene = SphereScene(seed=seed, px_size=3.0 / (flen * max(width, height)))
    cams = [main] + subs
    images = [render_rgb(scene, c, lighting) for c in cams]
    # sparse SfM features: random main-view pixels lifted to the surface
    xs = rng.uniform(0.05 * width, 0.95 * width, n_features)
    ys = rng.uniform(0.05 * height, 0.95 * height, n_features)
    X, _ = scene.intersect(main.center, pixel_rays(main, xs, ys))
    ys_g, xs_g = np.mgrid[0:height, 0:width].astype(np.float64)
    truth = depth_at(scene, main, xs_g + 0.5, ys_g + 0.5).astype(np.float32)
    return dict(scene=scene, cams=cams, images=images,
                features=X.astype(np.float32), truth=truth,
                view_ids=list(range(len(cams))))
