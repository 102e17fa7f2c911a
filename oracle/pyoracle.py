"""ctypes binding of the CPU oracle (oracle/libsmvs_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product package (smvs_amd/).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None

c_double_p = C.POINTER(C.c_double)
c_float_p = C.POINTER(C.c_float)
c_u8_p = C.POINTER(C.c_uint8)
c_u16_p = C.POINTER(C.c_uint16)
c_u32_p = C.POINTER(C.c_uint32)
c_u64_p = C.POINTER(C.c_uint64)
c_i32_p = C.POINTER(C.c_int32)


def build():
    """(Re)build the oracle shared library (and oracle/_ref when the
    reference tree is present)."""
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libsmvs_oracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.orc_bicubic_eval.restype = C.c_double
        _LIB.orc_vec_dot.restype = C.c_double
        _LIB.orc_linear_at_f32.restype = C.c_float
        _LIB.orc_linear_at_u8.restype = C.c_uint8
    return _LIB


def ref_ldl():
    """The reference's own ldl_inverse compiled from /root/reference (None if
    oracle/_ref was never built)."""
    global _REF
    if _REF is None:
        path = os.path.join(_HERE, "_ref", "libref_ldl.so")
        if not os.path.exists(path):
            return None
        _REF = C.CDLL(path)
    return _REF


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# ------------------------------------------------------------------ bicubic
def bicubic_coeffs(nodes16):
    nodes16 = f64(nodes16).reshape(16)
    out = np.zeros(16)
    lib().orc_bicubic_coeffs(_p(nodes16, c_double_p), _p(out, c_double_p))
    return out


def bicubic_eval(coeffs, kind, x, y):
    coeffs = f64(coeffs)
    return lib().orc_bicubic_eval(_p(coeffs, c_double_p), C.c_int(kind),
                                  C.c_double(x), C.c_double(y))


def node_derivatives(x, y):
    out = np.zeros(96)
    lib().orc_node_derivatives(C.c_double(x), C.c_double(y), _p(out, c_double_p))
    return out


def node_derivatives_for_patchsize(x, y, patchsize):
    out = np.zeros(96)
    lib().orc_node_derivatives_for_patchsize(C.c_double(x), C.c_double(y),
        C.c_double(patchsize), _p(out, c_double_p))
    return out


def node_derivatives_for_pixel(pid, patchsize):
    out = np.zeros(96)
    lib().orc_node_derivatives_for_pixel(C.c_int(pid), C.c_int(patchsize),
                                         _p(out, c_double_p))
    return out


def patch_values_at_pixels(nodes16, px, py, size, subsample=1):
    nodes16 = f64(nodes16).reshape(16)
    n = size * size
    pixels = np.zeros((n, 2)); depths = np.zeros(n)
    first = np.zeros((n, 2)); second = np.zeros((n, 3))
    pids = np.zeros(n, dtype=np.int32)
    cnt = lib().orc_patch_values_at_pixels(_p(nodes16, c_double_p), px, py,
        size, subsample, _p(pixels, c_double_p), _p(depths, c_double_p),
        _p(first, c_double_p), _p(second, c_double_p), _p(pids, c_i32_p))
    return pixels[:cnt], depths[:cnt], first[:cnt], second[:cnt], pids[:cnt]


# ----------------------------------------------------------- correspondence
class Corr(C.Structure):
    _fields_ = [("p", C.c_double), ("q", C.c_double), ("r", C.c_double),
                ("t", C.c_double * 3), ("w", C.c_double),
                ("w_prime", C.c_double * 2), ("a", C.c_double),
                ("b", C.c_double), ("d", C.c_double), ("d2", C.c_double),
                ("p_prime", C.c_double * 2), ("q_prime", C.c_double * 2),
                ("r_prime", C.c_double * 2)]


class Correspondence:
    def __init__(self, M, t, u, v, w, w_dx=0.0, w_dy=0.0):
        self.c = Corr()
        M = f64(M).reshape(9); t = f64(t).reshape(3)
        lib().orc_corr_update(C.byref(self.c), _p(M, c_double_p),
            _p(t, c_double_p), C.c_double(u), C.c_double(v), C.c_double(w),
            C.c_double(w_dx), C.c_double(w_dy))

    def fill(self):
        out = np.zeros(2)
        lib().orc_corr_fill(C.byref(self.c), _p(out, c_double_p))
        return out

    def jacobian(self):
        out = np.zeros(4)
        lib().orc_corr_fill_jacobian(C.byref(self.c), _p(out, c_double_p))
        return out

    def derivative(self, dn):
        dn = f64(dn); out = np.zeros((16, 2))
        lib().orc_corr_fill_derivative(C.byref(self.c), _p(dn, c_double_p),
                                       _p(out, c_double_p))
        return out

    def jacobian_derivative_grad(self, grad, dn):
        dn = f64(dn); grad = f64(grad); out = np.zeros((16, 2))
        lib().orc_corr_fill_jacobian_derivative_grad(C.byref(self.c),
            _p(grad, c_double_p), _p(dn, c_double_p), _p(out, c_double_p))
        return out

    @property
    def depth(self):
        return self.c.d


# ------------------------------------------------------ surface derivative
def fill_normal(x, y, inv_flen, w, dx, dy):
    out = np.zeros(3)
    lib().orc_fill_normal(*(C.c_double(v) for v in (x, y, inv_flen, w, dx, dy)),
                          _p(out, c_double_p))
    return out


def normal_derivative(dn, x, y, f, w, dx, dy):
    dn = f64(dn); out = np.zeros(48)
    lib().orc_normal_derivative(_p(dn, c_double_p),
        *(C.c_double(v) for v in (x, y, f, w, dx, dy)), _p(out, c_double_p))
    return out


def normal_divergence(x, y, f, w, dx, dy, dxy, dxx, dyy):
    out = np.zeros(6)
    lib().orc_normal_divergence(
        *(C.c_double(v) for v in (x, y, f, w, dx, dy, dxy, dxx, dyy)),
        _p(out, c_double_p))
    return out


def normal_divergence_deriv(dn, x, y, f, w, dx, dy, dxy, dxx, dyy):
    dn = f64(dn); out = np.zeros(96)
    lib().orc_normal_divergence_deriv(_p(dn, c_double_p),
        *(C.c_double(v) for v in (x, y, f, w, dx, dy, dxy, dxx, dyy)),
        _p(out, c_double_p))
    return out


def sh_evaluate_4_band(n):
    n = f64(n); out = np.zeros(16)
    lib().orc_sh_evaluate_4_band(_p(n, c_double_p), _p(out, c_double_p))
    return out


def sh_derivative_4_band(n):
    n = f64(n); out = np.zeros(48)
    lib().orc_sh_derivative_4_band(_p(n, c_double_p), _p(out, c_double_p))
    return out


# -------------------------------------------------------------------- algebra
def ldl_inverse(A):
    A = f64(A).copy(); n = A.shape[0]
    lib().orc_ldl_inverse(_p(A, c_double_p), C.c_int(n))
    return A


def ref_ldl_inverse(A):
    A = f64(A).copy(); n = A.shape[0]
    ref_ldl().ref_ldl_inverse(_p(A, c_double_p), C.c_int(n))
    return A


def vec_dot(a, b):
    a = f64(a); b = f64(b)
    return lib().orc_vec_dot(_p(a, c_double_p), _p(b, c_double_p),
                             C.c_size_t(a.size))


def linear_at_f32(img, x, y, ch):
    img = f32(img); h, w, c = img.shape
    return lib().orc_linear_at_f32(_p(img, c_float_p), w, h, c, C.c_float(x),
                                   C.c_float(y), ch)


# ------------------------------------------------------------ GN structures
class Surface(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("scale", C.c_int),
                ("patchsize", C.c_int), ("npx", C.c_int), ("npy", C.c_int),
                ("start_x", C.c_int), ("start_y", C.c_int),
                ("nodes", c_double_p), ("node_valid", c_u8_p),
                ("patch_valid", c_u8_p), ("patch_vis", c_u32_p)]


class SubView(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int),
                ("grad", c_float_p), ("hess", c_float_p)]


class Views(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int),
                ("flen", C.c_float), ("inv_flen", C.c_float),
                ("grad", c_float_p), ("shading", c_float_p),
                ("shading_grad", c_float_p), ("n_subs", C.c_int),
                ("subs", C.POINTER(SubView)), ("M", c_double_p),
                ("t", c_double_p)]


class GNOptions(C.Structure):
    _fields_ = [("regularization", C.c_double),
                ("light_surf_regularization", C.c_double)]


class OracleProblem:
    """Owns the numpy buffers behind orc_views / orc_surface.

    `surf` is a dict with keys width,height,scale,npx,npy,start_x,start_y,
    nodes [(N,4) f64], node_valid [N u8], patch_valid [P u8], patch_vis [P u32];
    `views` a dict with flen, inv_flen, grad [(H,W,2) f32], optional shading
    [(H,W)], shading_grad [(H,W,2)], subs = list of (grad (h,w,2), hess (h,w,3)),
    M [(S,9)], t [(S,3)].
    """

    def __init__(self, surf, views):
        self.nodes = f64(surf["nodes"]).reshape(-1, 4).copy()
        self.node_valid = np.ascontiguousarray(surf["node_valid"], dtype=np.uint8).copy()
        self.patch_valid = np.ascontiguousarray(surf["patch_valid"], dtype=np.uint8).copy()
        self.patch_vis = np.ascontiguousarray(surf["patch_vis"], dtype=np.uint32).copy()
        s = Surface()
        s.width, s.height = surf["width"], surf["height"]
        s.scale = surf["scale"]; s.patchsize = 1 << surf["scale"]
        s.npx, s.npy = surf["npx"], surf["npy"]
        s.start_x, s.start_y = surf["start_x"], surf["start_y"]
        s.nodes = _p(self.nodes, c_double_p)
        s.node_valid = _p(self.node_valid, c_u8_p)
        s.patch_valid = _p(self.patch_valid, c_u8_p)
        s.patch_vis = _p(self.patch_vis, c_u32_p)
        self.surf = s
        self.num_nodes = (s.npx + 1) * (s.npy + 1)
        self.node_stride = s.npx + 1

        self.grad = f32(views["grad"])
        self.shading = f32(views["shading"]) if views.get("shading") is not None else None
        self.shading_grad = f32(views["shading_grad"]) if views.get("shading_grad") is not None else None
        self.sub_bufs = [(f32(g), f32(h)) for g, h in views["subs"]]
        self.M = f64(views["M"]).reshape(-1, 9)
        self.t = f64(views["t"]).reshape(-1, 3)
        n = len(self.sub_bufs)
        self.subs = (SubView * n)()
        for i, (g, h) in enumerate(self.sub_bufs):
            self.subs[i].height, self.subs[i].width = g.shape[0], g.shape[1]
            self.subs[i].grad = _p(g, c_float_p)
            self.subs[i].hess = _p(h, c_float_p)
        v = Views()
        v.height, v.width = self.grad.shape[0], self.grad.shape[1]
        v.flen = views["flen"]; v.inv_flen = views["inv_flen"]
        v.grad = _p(self.grad, c_float_p)
        v.shading = _p(self.shading, c_float_p)
        v.shading_grad = _p(self.shading_grad, c_float_p)
        v.n_subs = n
        v.subs = self.subs
        v.M = _p(self.M, c_double_p); v.t = _p(self.t, c_double_p)
        self.views = v

    def gn_patch(self, patch_id, regularization, light_reg=0.0, lighting=None):
        ps = self.surf.patchsize
        nd = np.concatenate([node_derivatives_for_pixel(i, ps)
                             for i in range(ps * ps)])
        opts = GNOptions(regularization, light_reg)
        g = np.zeros(16); H = np.zeros(256)
        lt = f64(lighting) if lighting is not None else None
        lib().orc_gn_patch(C.byref(self.views), C.byref(self.surf),
            C.byref(opts), _p(lt, c_double_p), C.c_int(patch_id),
            _p(nd, c_double_p), _p(g, c_double_p), _p(H, c_double_p))
        return g, H.reshape(16, 16)

    def gn_construct(self, active, regularization, light_reg=0.0, lighting=None):
        N = self.num_nodes
        active = np.ascontiguousarray(active, dtype=np.uint8)
        opts = GNOptions(regularization, light_reg)
        H9 = np.zeros((N, 9, 16)); present = np.zeros((N, 9), dtype=np.uint8)
        g = np.zeros(4 * N); P = np.zeros((N, 16))
        lt = f64(lighting) if lighting is not None else None
        cnt = lib().orc_gn_construct(C.byref(self.views), C.byref(self.surf),
            C.byref(opts), _p(lt, c_double_p), _p(active, c_u8_p),
            _p(H9, c_double_p), _p(present, c_u8_p), _p(g, c_double_p),
            _p(P, c_double_p))
        return dict(H9=H9, present=present, g=g, P=P, active_patches=cnt)

    def spmv(self, H9, present, x):
        x = f64(x); y = np.zeros_like(x)
        lib().orc_block_spmv(self.num_nodes, self.node_stride,
            _p(f64(H9), c_double_p), _p(present, c_u8_p), _p(x, c_double_p),
            _p(y, c_double_p))
        return y

    def cg_solve(self, H9, present, P, b, max_iterations=200,
                 error_tolerance=1e-20, q_tolerance=1e-3):
        b = f64(b); x = np.zeros_like(b); it = C.c_int(0)
        info = lib().orc_cg_solve(self.num_nodes, self.node_stride,
            _p(f64(H9), c_double_p), _p(present, c_u8_p), _p(f64(P), c_double_p),
            _p(b, c_double_p), _p(x, c_double_p), C.c_int(max_iterations),
            C.c_double(error_tolerance), C.c_double(q_tolerance), C.byref(it))
        return x, it.value, info

    def update_and_reactivate(self, delta, active, full_optimization=False):
        delta = f64(delta)
        active = np.ascontiguousarray(active, dtype=np.uint8).copy()
        mean = C.c_double(0.0)
        n = lib().orc_update_and_reactivate(C.byref(self.views),
            C.byref(self.surf), _p(delta, c_double_p), _p(active, c_u8_p),
            C.c_int(1 if full_optimization else 0), C.byref(mean))
        return active, n, mean.value

    def depth_map(self):
        out = np.zeros((self.surf.height, self.surf.width), dtype=np.float32)
        lib().orc_depth_map(C.byref(self.surf), _p(out, c_float_p))
        return out

    def normal_map(self, inv_flen=None):
        out = np.zeros((self.surf.height, self.surf.width, 3), dtype=np.float32)
        lib().orc_normal_map(C.byref(self.surf),
            C.c_float(self.views.inv_flen if inv_flen is None else inv_flen),
            _p(out, c_float_p))
        return out


def light_accumulate(normals, image):
    normals = f32(normals); image = f32(image)
    A = np.zeros((16, 16)); b = np.zeros(16)
    lib().orc_light_accumulate(_p(normals, c_float_p), _p(image, c_float_p),
        C.c_int(image.size), _p(A, c_double_p), _p(b, c_double_p))
    return A, b


def light_solve(A, b):
    A = f64(A); b = f64(b); out = np.zeros(16)
    lib().orc_light_solve(_p(A, c_double_p), _p(b, c_double_p), _p(out, c_double_p))
    return out


# ------------------------------------------------------------------------ SGM
def census_filter(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    if img.ndim == 2:
        img = img[:, :, None]
    h, w, c = img.shape
    out = np.zeros((h, w, c), dtype=np.uint64)
    lib().orc_census_filter(_p(img, c_u8_p), w, h, c, _p(out, c_u64_p))
    return out


def sgm_depths(min_depth, max_depth, num_steps):
    out = np.zeros(num_steps, dtype=np.float32)
    lib().orc_sgm_depths(C.c_float(min_depth), C.c_float(max_depth),
                         C.c_int(num_steps), _p(out, c_float_p))
    return out


def sgm_cost_volume(main_img, neighbor, M, t, depths):
    main_img = np.ascontiguousarray(main_img, dtype=np.uint8)
    neighbor = np.ascontiguousarray(neighbor, dtype=np.uint8)
    M = f32(M).reshape(9); t = f32(t).reshape(3); depths = f32(depths)
    h, w = main_img.shape; nh, nw = neighbor.shape
    cost = np.zeros((h, w, depths.size), dtype=np.uint16)
    lib().orc_sgm_cost_volume(_p(main_img, c_u8_p), w, h, _p(neighbor, c_u8_p),
        nw, nh, _p(M, c_float_p), _p(t, c_float_p), _p(depths, c_float_p),
        C.c_int(depths.size), _p(cost, c_u16_p))
    return cost


def sgm_aggregate(cost, p1=6, p2=96, literal=False):
    cost = np.ascontiguousarray(cost, dtype=np.uint16)
    h, w, d = cost.shape
    sgm = np.zeros_like(cost)
    lib().orc_sgm_set_literal(C.c_int(1 if literal else 0))
    lib().orc_sgm_aggregate(_p(cost, c_u16_p), w, h, d, C.c_uint16(p1),
                            C.c_uint16(p2), _p(sgm, c_u16_p))
    lib().orc_sgm_set_literal(C.c_int(0))
    return sgm


def sgm_depth_from_volume(sgm, main_img, depths):
    sgm = np.ascontiguousarray(sgm, dtype=np.uint16)
    main_img = np.ascontiguousarray(main_img, dtype=np.uint8)
    depths = f32(depths)
    h, w, d = sgm.shape
    depth = np.zeros((h, w), dtype=np.float32)
    argmin = np.zeros((h, w), dtype=np.int32)
    lib().orc_sgm_depth_from_volume(_p(sgm, c_u16_p), _p(main_img, c_u8_p), w, h,
        _p(depths, c_float_p), d, _p(depth, c_float_p), _p(argmin, c_i32_p))
    return depth, argmin


def sgm_lr_check(d_main, d_neig, M, t):
    d_main = f32(d_main).copy(); d_neig = f32(d_neig)
    M = f32(M).reshape(9); t = f32(t).reshape(3)
    h, w = d_main.shape; nh, nw = d_neig.shape
    lib().orc_sgm_lr_check(_p(d_main, c_float_p), w, h, _p(d_neig, c_float_p),
                           nw, nh, _p(M, c_float_p), _p(t, c_float_p))
    return d_main


def bilateral_upsample(dm, ci, sigma=5.0, kernel_size=5):
    dm = f32(dm); ci = f32(ci)
    if ci.ndim == 2:
        ci = ci[:, :, None]
    h, w, c = ci.shape; dh, dw = dm.shape
    out = np.zeros((h, w), dtype=np.float32)
    lib().orc_bilateral_upsample(_p(dm, c_float_p), dw, dh, _p(ci, c_float_p),
        w, h, c, C.c_float(sigma), C.c_int(kernel_size), _p(out, c_float_p))
    return out


# ----------------------------------------------------------- whole optimizer
class ViewInput(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("channels", C.c_int),
                ("bytes", c_u8_p), ("flen", C.c_float), ("rot", C.c_float * 9),
                ("trans", C.c_float * 3), ("view_id", C.c_int)]


class Bundle(C.Structure):
    _fields_ = [("num_features", C.c_int), ("positions", c_float_p),
                ("ref_offsets", c_i32_p), ("ref_views", c_i32_p)]


class OptOptions(C.Structure):
    _fields_ = [("regularization", C.c_double),
                ("light_surf_regularization", C.c_double),
                ("num_iterations", C.c_int), ("min_scale", C.c_int),
                ("use_shading", C.c_int), ("use_sgm", C.c_int),
                ("full_optimization", C.c_int), ("sgm_width", C.c_int),
                ("sgm_height", C.c_int), ("gamma_correction", C.c_int)]


class OptLog(C.Structure):
    _fields_ = [("count", C.c_int), ("scale", C.c_int * 256),
                ("iter", C.c_int * 256), ("newton_steps", C.c_int * 256),
                ("valid_patches", C.c_int * 256), ("cg_iterations", C.c_int * 256),
                ("final_scale", C.c_int), ("final_patches", C.c_int),
                ("has_lighting", C.c_int), ("lighting", C.c_double * 16),
                ("loop_seconds", C.c_double * 256),
                ("active_patch_steps", C.c_longlong * 256)]


def _view_input(img, cam, view_id, keep):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    if img.ndim == 2:
        img = img[:, :, None]
    keep.append(img)
    v = ViewInput()
    v.height, v.width, v.channels = img.shape
    v.bytes = _p(img, c_u8_p)
    v.flen = cam.flen
    for i, x in enumerate(np.asarray(cam.R, dtype=np.float32).reshape(9)):
        v.rot[i] = float(x)
    for i, x in enumerate(np.asarray(cam.t, dtype=np.float32).reshape(3)):
        v.trans[i] = float(x)
    v.view_id = view_id
    return v


def optimize(inputs, regularization=0.01, light_reg=0.0, num_iterations=5,
             min_scale=2, use_shading=False, sgm_depth=None,
             full_optimization=False, gamma_correction=False):
    """orc_optimize on the dict of smvs_amd.synth.pipeline_inputs()."""
    keep = []
    cams, images = inputs["cams"], inputs["images"]
    main = _view_input(images[0], cams[0], inputs["view_ids"][0], keep)
    n_subs = len(cams) - 1
    subs = (ViewInput * n_subs)()
    for j in range(n_subs):
        subs[j] = _view_input(images[j + 1], cams[j + 1], inputs["view_ids"][j + 1], keep)
    feats = f32(inputs["features"]).reshape(-1, 3)
    nf = feats.shape[0]
    nviews = len(cams)
    offsets = (np.arange(nf + 1) * nviews).astype(np.int32)
    refs = np.tile(np.asarray(inputs["view_ids"], dtype=np.int32), nf)
    b = Bundle(nf, _p(feats, c_float_p), _p(offsets, c_i32_p), _p(refs, c_i32_p))
    o = OptOptions()
    o.regularization = regularization; o.light_surf_regularization = light_reg
    o.num_iterations = num_iterations; o.min_scale = min_scale
    o.use_shading = 1 if use_shading else 0
    o.use_sgm = 1 if sgm_depth is not None else 0
    o.full_optimization = 1 if full_optimization else 0
    o.gamma_correction = 1 if gamma_correction else 0
    sd = None
    if sgm_depth is not None:
        sd = f32(sgm_depth)
        o.sgm_height, o.sgm_width = sd.shape
    h, w = main.height, main.width
    depth = np.zeros((h, w), dtype=np.float32)
    normals = np.zeros((h, w, 3), dtype=np.float32)
    log = OptLog()
    rc = lib().orc_optimize(C.byref(main), subs, n_subs, C.byref(b),
        _p(sd, c_float_p), C.byref(o), _p(depth, c_float_p),
        _p(normals, c_float_p), C.byref(log))
    if rc != 0:
        raise RuntimeError("orc_optimize failed: %d" % rc)
    steps = [dict(scale=log.scale[i], iter=log.iter[i],
                  newton_steps=log.newton_steps[i],
                  valid_patches=log.valid_patches[i],
                  cg_iterations=log.cg_iterations[i],
                  loop_seconds=log.loop_seconds[i],
                  active_patch_steps=log.active_patch_steps[i]) for i in range(log.count)]
    return dict(depth=depth, normals=normals, log=steps,
                final_patches=log.final_patches,
                lighting=np.array(log.lighting[:]) if log.has_lighting else None)


def _bundle(inputs, keep):
    feats = f32(inputs["features"]).reshape(-1, 3)
    nf = feats.shape[0]
    nviews = len(inputs["cams"])
    offsets = (np.arange(nf + 1) * nviews).astype(np.int32)
    refs = np.tile(np.asarray(inputs["view_ids"], dtype=np.int32), nf)
    keep += [feats, offsets, refs]
    return Bundle(nf, _p(feats, c_float_p), _p(offsets, c_i32_p), _p(refs, c_i32_p))


def sgm_depth_range(inputs, view_index=0):
    """SGMStereo::fill_depth_range_for_view for one view of pipeline_inputs()."""
    keep = []
    v = _view_input(inputs["images"][view_index], inputs["cams"][view_index],
                    inputs["view_ids"][view_index], keep)
    b = _bundle(inputs, keep)
    out = np.zeros(2, dtype=np.float32)
    lib().orc_sgm_depth_range(C.byref(b), C.byref(v), _p(out, c_float_p))
    return out


def sgm_depth_for_view(inputs, sgm_scale=1, min_depth=0.0, max_depth=0.0,
                       num_steps=128, penalty1=6, penalty2=96, roundtrip=False):
    """reconstruct_sgm_depth_for_view (app/smvsrecon.cc:346-384) on the dict
    of smvs_amd.synth.pipeline_inputs(): SGM against the first two neighbours,
    L/R check, merge; roundtrip adds write_depth_to_view + get_sgm_depth."""
    keep = []
    cams, images = inputs["cams"], inputs["images"]
    main = _view_input(images[0], cams[0], inputs["view_ids"][0], keep)
    n = len(cams) - 1
    subs = (ViewInput * n)()
    for j in range(n):
        subs[j] = _view_input(images[j + 1], cams[j + 1], inputs["view_ids"][j + 1], keep)
    b = _bundle(inputs, keep)
    w, h = main.width, main.height
    for _ in range(sgm_scale):
        w, h = (w + 1) // 2, (h + 1) // 2
    out = np.zeros((h, w), dtype=np.float32)
    ow = C.c_int(0); oh = C.c_int(0)
    rc = lib().orc_sgm_depth_for_view(C.byref(main), subs, n, C.byref(b), sgm_scale,
        C.c_float(min_depth), C.c_float(max_depth), num_steps, penalty1, penalty2,
        1 if roundtrip else 0, _p(out, c_float_p), C.byref(ow), C.byref(oh))
    if rc != 0:
        raise RuntimeError("orc_sgm_depth_for_view failed: %d" % rc)
    assert (ow.value, oh.value) == (w, h)
    return out


def cut_depth_maps(cams, depths, normals, view_ids=None):
    """generate_mesh's normal preparation + MeshGenerator::cut_depth_maps
    (mesh_generator.cc:189-208, 24-158).  depths[i]: (h, w) ray-length depth,
    normals[i]: (h, w, 3) camera-space.  Returns (cut depths, world normals)."""
    n = len(cams)
    keep = []
    views = (ViewInput * n)()
    dummy = np.zeros((1, 1, 1), np.uint8)
    for i in range(n):
        views[i] = _view_input(dummy, cams[i], i if view_ids is None else view_ids[i], keep)
    d = [f32(x).copy() for x in depths]
    nm = [f32(x).copy() for x in normals]
    w = (C.c_int * n)(*[x.shape[1] for x in d])
    h = (C.c_int * n)(*[x.shape[0] for x in d])
    dp = (c_float_p * n)(*[_p(x, c_float_p) for x in d])
    npp = (c_float_p * n)(*[_p(x, c_float_p) for x in nm])
    rc = lib().orc_cut_depth_maps(n, views, w, h, dp, npp)
    if rc != 0:
        raise RuntimeError("orc_cut_depth_maps failed: %d" % rc)
    return d, nm


def rescale_half_size_u8(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    out = np.zeros(((h + 1) // 2, (w + 1) // 2), dtype=np.uint8)
    lib().orc_rescale_half_size_u8(_p(img, c_u8_p), w, h, _p(out, c_u8_p))
    return out


def gradients_and_hessian(img):
    img = f32(img); h, w = img.shape
    g = np.zeros((h, w, 2), np.float32); hs = np.zeros((h, w, 3), np.float32)
    lib().orc_gradients_and_hessian(_p(img, c_float_p), w, h, _p(g, c_float_p),
                                    _p(hs, c_float_p))
    return g, hs


def scale_planes(img_u8, scale):
    img = np.ascontiguousarray(img_u8, dtype=np.uint8)
    if img.ndim == 2:
        img = img[:, :, None]
    h, w, c = img.shape
    g = np.zeros((h, w, 2), np.float32); hs = np.zeros((h, w, 3), np.float32)
    lib().orc_scale_planes(_p(img, c_u8_p), w, h, c, C.c_int(scale),
                           _p(g, c_float_p), _p(hs, c_float_p))
    return g, hs


# ---------------------------------------------------------------------------
# topology tests between Newton batches (depth_optimizer.cc:360-604, 747-912)
# ---------------------------------------------------------------------------
class TopoView(C.Structure):
    _fields_ = [("w", C.c_int), ("h", C.c_int), ("c", C.c_int),
                ("image", c_float_p), ("grad", c_float_p), ("flen", C.c_float)]


class TopologyProblem:
    """Surface + float images + gradient planes for the orc_topology_* entry
    points.  images[0] / grads[0] belong to the main view."""

    def __init__(self, surf, images, grads, M, t, flen):
        self.nodes = f64(surf["nodes"]).reshape(-1, 4).copy()
        self.node_valid = np.ascontiguousarray(surf["node_valid"], dtype=np.uint8).copy()
        self.patch_valid = np.ascontiguousarray(surf["patch_valid"], dtype=np.uint8).copy()
        self.patch_vis = np.ascontiguousarray(surf["patch_vis"], dtype=np.uint32).copy()
        s = Surface()
        s.width, s.height = surf["width"], surf["height"]
        s.scale = surf["scale"]; s.patchsize = 1 << surf["scale"]
        s.npx, s.npy = surf["npx"], surf["npy"]
        s.start_x, s.start_y = surf["start_x"], surf["start_y"]
        s.nodes = _p(self.nodes, c_double_p)
        s.node_valid = _p(self.node_valid, c_u8_p)
        s.patch_valid = _p(self.patch_valid, c_u8_p)
        s.patch_vis = _p(self.patch_vis, c_u32_p)
        self.surf = s
        self.images = [f32(im if im.ndim == 3 else im[:, :, None]) for im in images]
        self.grads = [f32(g) for g in grads]
        self.M = f64(M).reshape(-1, 9); self.t = f64(t).reshape(-1, 3)
        n = len(images) - 1
        self.n_subs = n
        self.main = TopoView()
        self.subs = (TopoView * n)()
        for k, tv in enumerate([self.main] + [self.subs[j] for j in range(n)]):
            im = self.images[k]
            tv.h, tv.w, tv.c = im.shape
            tv.image = _p(im, c_float_p)
            tv.grad = _p(self.grads[k], c_float_p)
            tv.flen = flen

    def subviews(self, sgm_depth=None):
        sd = f32(sgm_depth) if sgm_depth is not None else None
        lib().orc_topology_subviews(C.byref(self.surf), C.byref(self.main), self.subs,
                                    self.n_subs, _p(self.M, c_double_p), _p(self.t, c_double_p),
                                    _p(sd, c_float_p), 1 if sd is not None else 0)
        return self.patch_vis.copy()

    def patch_mse(self):
        out = np.zeros(self.patch_valid.size)
        lib().orc_topology_patch_mse(C.byref(self.surf), C.byref(self.main), self.subs,
                                     self.n_subs, _p(self.M, c_double_p), _p(self.t, c_double_p),
                                     _p(out, c_double_p))
        return out

    def cut_boundaries(self):
        lib().orc_topology_cut_boundaries.restype = C.c_int
        return lib().orc_topology_cut_boundaries(C.byref(self.surf), C.byref(self.main),
                                                 self.subs, self.n_subs, _p(self.M, c_double_p),
                                                 _p(self.t, c_double_p))


# ------------------------------------------------ scene level: view selection
class SceneView(C.Structure):
    _fields_ = [("present", C.c_int), ("id", C.c_int), ("flen", C.c_float),
                ("rot", C.c_float * 9), ("trans", C.c_float * 3),
                ("has_image", C.c_int), ("width", C.c_int), ("height", C.c_int)]


def select_neighbors(scene, view, num_neighbors=6, use_bundle=True):
    """ViewSelection::get_neighbors_for_view (lib/view_selection.cc:14-161).
    scene: dict(views=[dict(present, id, flen, rot, trans, has_image, width,
    height)], features=(F, 3) float32, refs=[list of view ids per feature])."""
    keep = []
    n = len(scene["views"])
    views = (SceneView * n)()
    for i, v in enumerate(scene["views"]):
        views[i].present = 1 if v.get("present", True) else 0
        views[i].id = int(v["id"])
        views[i].flen = float(v["flen"])
        for k, x in enumerate(np.asarray(v["rot"], dtype=np.float32).reshape(9)):
            views[i].rot[k] = float(x)
        for k, x in enumerate(np.asarray(v["trans"], dtype=np.float32).reshape(3)):
            views[i].trans[k] = float(x)
        views[i].has_image = 1 if v.get("has_image", True) else 0
        views[i].width, views[i].height = int(v["width"]), int(v["height"])
    bundle = None
    if use_bundle:
        feats = f32(scene["features"]).reshape(-1, 3)
        offsets = np.zeros(len(scene["refs"]) + 1, dtype=np.int32)
        offsets[1:] = np.cumsum([len(r) for r in scene["refs"]])
        flat = np.asarray([i for r in scene["refs"] for i in r], dtype=np.int32)
        if flat.size == 0:
            flat = np.zeros(1, dtype=np.int32)
        keep += [feats, offsets, flat]
        bundle = Bundle(feats.shape[0], _p(feats, c_float_p), _p(offsets, c_i32_p),
                        _p(flat, c_i32_p))
    out = np.zeros(max(n, 1), dtype=np.int32)
    f = lib().orc_select_neighbors
    f.restype = C.c_int
    count = f(C.c_int(n), views, C.byref(bundle) if bundle is not None else None,
              C.c_int(view), C.c_int(num_neighbors), _p(out, c_i32_p))
    return [int(x) for x in out[:count]]


# ------------------------------------------------ Surface operations on their own
def surface_script(inputs, init_scale, ops, init_depth=None, delete_every=3):
    """lib/surface.cc: Surface::create + a script of operations (1 expand,
    2 subdivide_patches, 3 fill_patches_from_depth, 4 remove_isolated_patches,
    5 delete every delete_every-th valid patch + remove_nodes_without_patch).
    inputs as for optimize(); returns dict(scale, npx, npy, start_x, start_y,
    nodes, node_valid, patch_valid)."""
    keep = []
    main = _view_input(inputs["images"][0], inputs["cams"][0], inputs["view_ids"][0], keep)
    bundle = _bundle(inputs, keep)
    h, w = np.asarray(inputs["images"][0]).shape[:2]
    cap_n, cap_p = (w + 2) * (h + 2), (w + 1) * (h + 1)
    nodes = np.zeros(cap_n * 4); nv = np.zeros(cap_n, np.uint8); pv = np.zeros(cap_p, np.uint8)
    info = np.zeros(5, np.int32)
    ops_a = np.asarray(list(ops) + [0], dtype=np.int32)
    depth = None if init_depth is None else f32(init_depth)
    f = lib().orc_surface_script
    f.restype = C.c_int
    rc = f(C.byref(main), C.byref(bundle), _p(depth, c_float_p) if depth is not None else None,
           C.c_int(init_scale), _p(ops_a, c_i32_p), C.c_int(len(ops)), C.c_int(delete_every),
           _p(info, c_i32_p), _p(nodes, c_double_p), _p(nv, c_u8_p), _p(pv, c_u8_p))
    if rc != 0:
        raise RuntimeError("orc_surface_script failed")
    scale, npx, npy, sx, sy = (int(x) for x in info)
    nn, npatch = (npx + 1) * (npy + 1), npx * npy
    return dict(scale=scale, npx=npx, npy=npy, start_x=sx, start_y=sy,
                nodes=nodes[:4 * nn].reshape(nn, 4).copy(), node_valid=nv[:nn].copy(),
                patch_valid=pv[:npatch].copy())


def sgm_image(inputs, view_index=0, halvings=1):
    keep = []
    v = _view_input(inputs["images"][view_index], inputs["cams"][view_index],
                    inputs["view_ids"][view_index], keep)
    out = np.zeros(v.width * v.height, np.uint8)
    w, h = C.c_int(0), C.c_int(0)
    lib().orc_sgm_image(C.byref(v), C.c_int(halvings), _p(out, c_u8_p), C.byref(w), C.byref(h))
    return out[:w.value * h.value].reshape(h.value, w.value).copy()


def view_reprojection(inputs, src, dst):
    keep = []
    a = _view_input(inputs["images"][src], inputs["cams"][src], inputs["view_ids"][src], keep)
    b = _view_input(inputs["images"][dst], inputs["cams"][dst], inputs["view_ids"][dst], keep)
    M = np.zeros(9, np.float32); t = np.zeros(3, np.float32)
    f = lib().orc_view_reprojection
    f.restype = None
    f(C.byref(a), C.byref(b), _p(M, c_float_p), _p(t, c_float_p))
    return M, t


def rescale_half_size_gaussian(img):
    """mve::image::rescale_half_size_gaussian<uint8_t> as restated in
    oracle/smvs_oracle_front.c ([MVE-unverified] M29)."""
    a = np.ascontiguousarray(img, dtype=np.uint8)
    squeeze = a.ndim == 2
    if squeeze:
        a = a[:, :, None]
    h, w, c = a.shape
    out = np.zeros(((h + 1) // 2, (w + 1) // 2, c), np.uint8)
    lib().orc_rescale_half_size_gaussian_u8(_p(a, c_u8_p), w, h, c, _p(out, c_u8_p))
    return out[:, :, 0] if squeeze else out
