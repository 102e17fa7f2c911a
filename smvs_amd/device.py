"""Thin Python handle on one smvs_ctx (one reference view on one GPU).

All arithmetic happens in libsmvs_hip.so; this module only marshals numpy
buffers across the C ABI of include/smvs_hip.h.
"""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import LoopParams, LoopStats, check

_dp = C.POINTER(C.c_double)
_fp = C.POINTER(C.c_float)
_u8p = C.POINTER(C.c_uint8)
_u16p = C.POINTER(C.c_uint16)
_u32p = C.POINTER(C.c_uint32)
_i32p = C.POINTER(C.c_int32)


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def device_count():
    return _capi.load().smvs_device_count()


class ViewContext:
    """One reference view + its neighbours on one GPU."""

    def __init__(self, width, height, n_subs, device=0):
        self.lib = _capi.load()
        self.width, self.height, self.n_subs = width, height, n_subs
        self.handle = C.c_void_p()
        check(self.lib.smvs_ctx_create(device, width, height, n_subs,
                                       C.byref(self.handle)))
        self.num_nodes = 0
        self.num_patches = 0

    def close(self):
        if self.handle:
            self.lib.smvs_ctx_destroy(self.handle)
            self.handle = C.c_void_p()
            for ptr in getattr(self, "_pinned", []):
                self.lib.smvs_pinned_free(ptr)
            self._pinned = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    SOLVERS = dict(auto=0, streaming=1, resident_ref=2)

    def set_solver(self, mode):
        """smvs_ctx_set_solver: 'auto' | 'streaming' | 'resident_ref'."""
        check(self.lib.smvs_ctx_set_solver(self.handle, self.SOLVERS[mode]))

    # ------------------------------------------------------------ uploads
    def set_views(self, views):
        """views: dict as produced by smvs_amd.synth.make_problem."""
        M = _f64(views["M"]).reshape(-1, 9); t = _f64(views["t"]).reshape(-1, 3)
        assert M.shape[0] == self.n_subs
        check(self.lib.smvs_ctx_set_cameras(self.handle, _p(M, _dp), _p(t, _dp),
              C.c_float(views["flen"]), C.c_float(views["inv_flen"])))
        grad = _f32(views["grad"])
        assert grad.shape == (self.height, self.width, 2)
        sh = _f32(views["shading"]) if views.get("shading") is not None else None
        shg = _f32(views["shading_grad"]) if views.get("shading_grad") is not None else None
        check(self.lib.smvs_ctx_upload_main(self.handle, _p(grad, _fp),
                                            _p(sh, _fp), _p(shg, _fp)))
        for j, (g, h) in enumerate(views["subs"]):
            g = _f32(g); h = _f32(h)
            check(self.lib.smvs_ctx_upload_sub(self.handle, j, g.shape[1],
                  g.shape[0], _p(g, _fp), _p(h, _fp)))

    def upload_image(self, view, img_u8):
        """view = -1: main view, otherwise neighbour index."""
        img = np.ascontiguousarray(img_u8, dtype=np.uint8)
        if img.ndim == 2:
            img = img[:, :, None]
        h, w, c = img.shape
        check(self.lib.smvs_ctx_upload_image(self.handle, view, w, h, c,
                                             _p(img, _u8p)))
        if not hasattr(self, "_image_shapes"):
            self._image_shapes = {}
        self._image_shapes[view] = (h, w)

    def upload_image_async(self, view, img_u8):
        """smvs_ctx_upload_image_async from page-locked memory: the image is
        copied into a smvs_pinned_alloc buffer that this object keeps until
        close(); the transfer and the conversion are only enqueued."""
        img = np.ascontiguousarray(img_u8, dtype=np.uint8)
        if img.ndim == 2:
            img = img[:, :, None]
        h, w, c = img.shape
        ptr = C.c_void_p()
        check(self.lib.smvs_pinned_alloc(C.c_size_t(img.size), C.byref(ptr)))
        if not hasattr(self, "_pinned"):
            self._pinned = []
        self._pinned.append(ptr)
        C.memmove(ptr, img.ctypes.data, img.size)
        check(self.lib.smvs_ctx_upload_image_async(self.handle, view, w, h, c,
                                                   C.cast(ptr, _u8p)))
        if not hasattr(self, "_image_shapes"):
            self._image_shapes = {}
        self._image_shapes[view] = (h, w)

    def sgm_init_depth(self, dm, sigma=5.0, kernel_size=5):
        """depthmap_bilateral_filter(dm, main image) guided by the uploaded main
        image; the result also stays in the context for topology_subviews()."""
        if dm is None:
            check(self.lib.smvs_ctx_sgm_init_depth(self.handle, None, 0, 0,
                  C.c_float(sigma), kernel_size, None))
            return None
        d = _f32(dm)
        out = np.zeros((self.height, self.width), dtype=np.float32)
        check(self.lib.smvs_ctx_sgm_init_depth(self.handle, _p(d, _fp), d.shape[1],
              d.shape[0], C.c_float(sigma), kernel_size, _p(out, _fp)))
        return out

    def sgm_init_depth_mve(self, dm_mve, inv_calibration9, sigma=5.0, kernel_size=5,
                           dm_is_z_depth=False):
        """sgm_init_depth() from the map as the view stores it (MVE's ray-length
        convention): converted to z-depth on the device (smvs_ctx_sgm_init_depth_mve)."""
        d = _f32(dm_mve)
        inv = _f32(np.asarray(inv_calibration9).reshape(9))
        out = np.zeros((self.height, self.width), dtype=np.float32)
        check(self.lib.smvs_ctx_sgm_init_depth_mve(self.handle, _p(d, _fp), d.shape[1],
              d.shape[0], _p(inv, _fp), 1 if dm_is_z_depth else 0, C.c_float(sigma),
              kernel_size, _p(out, _fp)))
        return out

    def set_scale(self, scale):
        check(self.lib.smvs_ctx_set_scale(self.handle, scale))

    def download_planes(self, view):
        h, w = self._image_shapes[view]
        grad = np.zeros((h, w, 2), np.float32)
        hess = np.zeros((h, w, 3), np.float32) if view >= 0 else None
        check(self.lib.smvs_ctx_download_planes(self.handle, view, _p(grad, _fp),
                                                _p(hess, _fp)))
        return grad, hess

    def set_surface(self, surf):
        nodes = _f64(surf["nodes"]).reshape(-1, 4)
        nv = np.ascontiguousarray(surf["node_valid"], dtype=np.uint8)
        pv = np.ascontiguousarray(surf["patch_valid"], dtype=np.uint8)
        vis = np.ascontiguousarray(surf["patch_vis"], dtype=np.uint32)
        self.num_nodes = (surf["npx"] + 1) * (surf["npy"] + 1)
        self.num_patches = surf["npx"] * surf["npy"]
        assert nodes.shape[0] == self.num_nodes and pv.size == self.num_patches
        check(self.lib.smvs_ctx_set_surface(self.handle, surf["scale"],
              surf["npx"], surf["npy"], surf["start_x"], surf["start_y"],
              _p(nodes, _dp), _p(nv, _u8p), _p(pv, _u8p), _p(vis, _u32p)))

    def set_active(self, active=None):
        a = None if active is None else np.ascontiguousarray(active, dtype=np.uint8)
        check(self.lib.smvs_ctx_set_active(self.handle, _p(a, _u8p)))

    def get_active(self):
        a = np.zeros(self.num_nodes, dtype=np.uint8); n = C.c_int(0)
        check(self.lib.smvs_get_active(self.handle, _p(a, _u8p), C.byref(n)))
        return a, n.value

    def get_nodes(self):
        nodes = np.zeros((self.num_nodes, 4))
        check(self.lib.smvs_get_nodes(self.handle, _p(nodes, _dp)))
        return nodes

    def set_nodes(self, nodes):
        nodes = _f64(nodes).reshape(-1, 4)
        assert nodes.shape[0] == self.num_nodes
        check(self.lib.smvs_set_nodes(self.handle, _p(nodes, _dp)))

    def save_nodes(self):
        """Keep a device-resident copy of the nodes (restore_nodes brings it
        back without a transfer)."""
        check(self.lib.smvs_ctx_save_nodes(self.handle))

    def restore_nodes(self):
        check(self.lib.smvs_ctx_restore_nodes(self.handle))

    def clone_loop_state(self):
        """smvs_ctx_clone_loop_state: a second context holding what this one's
        Newton loop reads now (planes, cameras, surface, masks), nodes saved."""
        other = ViewContext.__new__(ViewContext)
        other.lib = self.lib
        other.width, other.height, other.n_subs = self.width, self.height, self.n_subs
        other.handle = C.c_void_p()
        check(self.lib.smvs_ctx_clone_loop_state(self.handle, C.byref(other.handle)))
        other.num_nodes, other.num_patches = self.num_nodes, self.num_patches
        return other

    # ------------------------------------------------------------ GN step
    def gn_construct(self, regularization, light_reg=0.0, lighting=None):
        lt = _f64(lighting) if lighting is not None else None
        n = C.c_int(0)
        check(self.lib.smvs_gn_construct(self.handle, C.c_double(regularization),
              C.c_double(light_reg), _p(lt, _dp), C.byref(n)))
        return n.value

    def gn_download(self):
        N = self.num_nodes
        H9 = np.zeros((N, 9, 16)); g = np.zeros(4 * N); P = np.zeros((N, 16))
        check(self.lib.smvs_gn_download(self.handle, _p(H9, _dp), _p(g, _dp),
                                        _p(P, _dp)))
        return H9, g, P

    def gn_upload(self, H9, g, P):
        H9 = _f64(H9); g = _f64(g); P = _f64(P)
        check(self.lib.smvs_gn_upload(self.handle, _p(H9, _dp), _p(g, _dp),
                                      _p(P, _dp)))

    def gn_patch_systems(self):
        Hp = np.zeros((self.num_patches, 16, 16)); gp = np.zeros((self.num_patches, 16))
        check(self.lib.smvs_gn_download_patch_systems(self.handle, _p(Hp, _dp),
                                                      _p(gp, _dp)))
        return Hp, gp

    def cg_solve(self, max_iterations=200, error_tolerance=-1.0, q_tolerance=1e-3):
        it = C.c_int(0); info = C.c_int(0)
        check(self.lib.smvs_cg_solve(self.handle, max_iterations,
              C.c_double(error_tolerance), C.c_double(q_tolerance),
              C.byref(it), C.byref(info)))
        return it.value, info.value

    def cg_x(self):
        x = np.zeros(4 * self.num_nodes)
        check(self.lib.smvs_cg_download_x(self.handle, _p(x, _dp)))
        return x

    def cg_set_x(self, x):
        x = _f64(x)
        check(self.lib.smvs_cg_upload_x(self.handle, _p(x, _dp)))

    def update_and_reactivate(self, threshold=0.15, full_optimization=False):
        n = C.c_int(0); mean = C.c_double(0.0); nan = C.c_int(0)
        check(self.lib.smvs_update_and_reactivate(self.handle, C.c_double(threshold),
              1 if full_optimization else 0, C.byref(n), C.byref(mean), C.byref(nan)))
        return n.value, mean.value, nan.value

    def run_loop(self, regularization, light_reg=0.0, lighting=None,
                 full_optimization=False, max_newton_steps=200,
                 cg_max_iterations=200, reset_active=True,
                 active_threshold=0.15, full_opt_threshold=0.01):
        # (the struct of the previous call is reused when nothing changed: a
        # Newton batch is short enough for the marshalling to show)
        key = (regularization, light_reg, full_optimization, max_newton_steps,
               cg_max_iterations, reset_active, active_threshold, full_opt_threshold)
        cached = getattr(self, "_loop_params", None)
        if lighting is None and cached is not None and cached[0] == key:
            p = cached[1]
            s = LoopStats()
            check(self.lib.smvs_gn_run_loop(self.handle, C.byref(p), C.byref(s)))
            return dict(newton_steps=s.newton_steps,
                        linear_iterations=s.linear_iterations,
                        active_patch_steps=s.active_patch_steps,
                        final_active_nodes=s.final_active_nodes,
                        nan_break=s.nan_break)
        p = LoopParams()
        p.regularization = regularization
        p.light_surf_regularization = light_reg
        p.full_optimization = 1 if full_optimization else 0
        p.max_newton_steps = max_newton_steps
        p.cg_max_iterations = cg_max_iterations
        p.cg_q_tolerance = 1e-3
        p.active_threshold = active_threshold
        p.full_opt_threshold = full_opt_threshold
        p.use_lighting = 1 if lighting is not None else 0
        if lighting is not None:
            for i in range(16):
                p.lighting[i] = float(lighting[i])
        p.reset_active = 1 if reset_active else 0
        self._loop_params = (key, p) if lighting is None else None
        s = LoopStats()
        check(self.lib.smvs_gn_run_loop(self.handle, C.byref(p), C.byref(s)))
        return dict(newton_steps=s.newton_steps,
                    linear_iterations=s.linear_iterations,
                    active_patch_steps=s.active_patch_steps,
                    final_active_nodes=s.final_active_nodes,
                    nan_break=s.nan_break)

    # ------------------------------------------------------------ outputs
    def depth_map(self):
        out = np.zeros((self.height, self.width), dtype=np.float32)
        check(self.lib.smvs_get_depth_map(self.handle, _p(out, _fp)))
        return out

    def normal_map(self):
        out = np.zeros((self.height, self.width, 3), dtype=np.float32)
        check(self.lib.smvs_get_normal_map(self.handle, _p(out, _fp)))
        return out

    def maps(self, inv_calibration=None, pinned=False):
        """smvs_get_maps: depth and normal map in one pass; inv_calibration
        (9 floats) -> the depth in MVE's ray-length convention; pinned: the
        outputs live in page-locked memory from smvs_pinned_alloc (direct DMA)."""
        import ctypes as C
        inv = None if inv_calibration is None \
            else np.ascontiguousarray(inv_calibration, dtype=np.float32).reshape(9)
        n = self.height * self.width
        if not pinned:
            depth = np.zeros((self.height, self.width), dtype=np.float32)
            normals = np.zeros((self.height, self.width, 3), dtype=np.float32)
            check(self.lib.smvs_get_maps(self.handle, _p(inv, _fp) if inv is not None else None,
                                         _p(depth, _fp), _p(normals, _fp)))
            return depth, normals
        pd, pn = C.c_void_p(), C.c_void_p()
        check(self.lib.smvs_pinned_alloc(C.c_size_t(4 * n), C.byref(pd)))
        check(self.lib.smvs_pinned_alloc(C.c_size_t(12 * n), C.byref(pn)))
        try:
            check(self.lib.smvs_get_maps(self.handle, _p(inv, _fp) if inv is not None else None,
                                         C.cast(pd, _fp), C.cast(pn, _fp)))
            depth = np.ctypeslib.as_array(C.cast(pd, _fp), (self.height, self.width)).copy()
            normals = np.ctypeslib.as_array(C.cast(pn, _fp),
                                            (self.height, self.width, 3)).copy()
        finally:
            check(self.lib.smvs_pinned_free(pd))
            check(self.lib.smvs_pinned_free(pn))
        return depth, normals

    def light_accumulate(self):
        A = np.zeros((16, 16)); b = np.zeros(16)
        check(self.lib.smvs_light_accumulate(self.handle, _p(A, _dp), _p(b, _dp)))
        return A, b

    def light_accumulate_dev(self):
        """Leaves A (256) + b (16) doubles in the context's device buffer and
        returns its device address (for an in-place RCCL all-reduce)."""
        ptr = C.c_void_p()
        check(self.lib.smvs_light_accumulate_dev(self.handle, C.byref(ptr)))
        return ptr.value

    def light_upload(self, A, b):
        A = _f64(A).reshape(16, 16); b = _f64(b).reshape(16)
        check(self.lib.smvs_light_upload(self.handle, _p(A, _dp), _p(b, _dp)))

    def light_download(self):
        A = np.zeros((16, 16)); b = np.zeros(16)
        check(self.lib.smvs_light_download(self.handle, _p(A, _dp), _p(b, _dp)))
        return A, b

    # ------------------------------------------------------------ topology
    def topology_subviews(self, sgm_depth=None, use_ncc=True):
        """create_subview_surfaces' per-(patch, neighbour) tests -> bit masks.
        sgm_depth None: the map sgm_init_depth() left in the context, if any."""
        sd = _f32(sgm_depth) if sgm_depth is not None else None
        if sd is not None:
            assert sd.shape == (self.height, self.width)
        vis = np.zeros(self.num_patches, dtype=np.uint32)
        check(self.lib.smvs_topology_subviews(self.handle, _p(sd, _fp),
              1 if use_ncc else 0, _p(vis, _u32p)))
        return vis

    def topology_patch_mse(self):
        mse = np.zeros(self.num_patches)
        check(self.lib.smvs_topology_patch_mse(self.handle, _p(mse, _dp)))
        return mse

    def topology_cut_boundaries(self, inv_calibration):
        """`while (deleted > 10) cut_boundaries()` -> (patch_valid, node_valid, deleted)."""
        k = _f32(inv_calibration).reshape(9)
        pv = np.zeros(self.num_patches, dtype=np.uint8)
        nv = np.zeros(self.num_nodes, dtype=np.uint8)
        n = C.c_int(0)
        check(self.lib.smvs_topology_cut_boundaries(self.handle, _p(k, _fp),
              _p(pv, _u8p), _p(nv, _u8p), C.byref(n)))
        return pv, nv, n.value

    def synchronize(self):
        check(self.lib.smvs_ctx_synchronize(self.handle))

    # ---------------------------------------------------------- profiling
    def profile(self, on=True):
        check(self.lib.smvs_profile_enable(self.handle, 1 if on else 0))

    def profile_reset(self):
        check(self.lib.smvs_profile_reset(self.handle))

    def profile_get(self):
        ms = (C.c_double * 8)(); cnt = (C.c_longlong * 8)()
        check(self.lib.smvs_profile_get(self.handle, ms, cnt))
        return {k: (ms[i], cnt[i]) for i, k in enumerate(_capi.K_NAMES)}


def sgm_run(main_img, neighbor_img, M, t, min_depth, max_depth, num_steps=128,
            p1=6, p2=96, device=0, want_volumes=False):
    lib = _capi.load()
    main_img = np.ascontiguousarray(main_img, dtype=np.uint8)
    neighbor_img = np.ascontiguousarray(neighbor_img, dtype=np.uint8)
    M = _f32(M).reshape(9); t = _f32(t).reshape(3)
    h, w = main_img.shape; nh, nw = neighbor_img.shape
    depth = np.zeros((h, w), dtype=np.float32)
    argmin = np.zeros((h, w), dtype=np.int32)
    cost = np.zeros((h, w, num_steps), dtype=np.uint16) if want_volumes else None
    sgm = np.zeros((h, w, num_steps), dtype=np.uint16) if want_volumes else None
    check(lib.smvs_sgm_run(device, _p(main_img, _u8p), w, h,
          _p(neighbor_img, _u8p), nw, nh, _p(M, _fp), _p(t, _fp),
          C.c_float(min_depth), C.c_float(max_depth), num_steps,
          C.c_uint16(p1), C.c_uint16(p2), _p(depth, _fp), _p(argmin, _i32p),
          _p(cost, _u16p), _p(sgm, _u16p)))
    return dict(depth=depth, argmin=argmin, cost=cost, sgm=sgm)


def bilateral_upsample(dm, ci, sigma=5.0, kernel_size=5, device=0):
    lib = _capi.load()
    dm = _f32(dm); ci = _f32(ci)
    if ci.ndim == 2:
        ci = ci[:, :, None]
    h, w, c = ci.shape; dh, dw = dm.shape
    out = np.zeros((h, w), dtype=np.float32)
    check(lib.smvs_bilateral_upsample(device, _p(dm, _fp), dw, dh, _p(ci, _fp),
          w, h, c, C.c_float(sigma), kernel_size, _p(out, _fp)))
    return out


class SgmNeighbor(C.Structure):
    """smvs_sgm_neighbor of include/smvs_hip.h."""
    _fields_ = [("image", _u8p), ("width", C.c_int), ("height", C.c_int),
                ("M_fwd", C.c_float * 9), ("t_fwd", C.c_float * 3),
                ("M_bwd", C.c_float * 9), ("t_bwd", C.c_float * 3),
                ("range_main", C.c_float * 2), ("range_neighbor", C.c_float * 2)]


def sgm_depth_for_view(main_img, neighbors, num_steps=128, p1=6, p2=96, device=0):
    """reconstruct_sgm_depth_for_view on the device.  neighbors: list of dicts
    {image, M_fwd, t_fwd, M_bwd, t_bwd, range_main, range_neighbor} (SGM-scale
    u8 images, float reprojections)."""
    lib = _capi.load()
    main_img = np.ascontiguousarray(main_img, dtype=np.uint8)
    h, w = main_img.shape
    keep = []
    arr = (SgmNeighbor * len(neighbors))()
    for k, nb in enumerate(neighbors):
        img = np.ascontiguousarray(nb["image"], dtype=np.uint8)
        keep.append(img)
        arr[k].image = _p(img, _u8p)
        arr[k].height, arr[k].width = img.shape
        for name, n in (("M_fwd", 9), ("t_fwd", 3), ("M_bwd", 9), ("t_bwd", 3),
                        ("range_main", 2), ("range_neighbor", 2)):
            v = _f32(nb[name]).reshape(n)
            for i in range(n):
                getattr(arr[k], name)[i] = float(v[i])
    depth = np.zeros((h, w), dtype=np.float32)
    check(lib.smvs_sgm_depth_for_view(device, _p(main_img, _u8p), w, h, arr,
          len(neighbors), num_steps, C.c_uint16(p1), C.c_uint16(p2), _p(depth, _fp)))
    return depth


class MeshView(C.Structure):
    """smvs_mesh_view of include/smvs_hip.h."""
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("flen", C.c_float),
                ("rot", C.c_float * 9), ("trans", C.c_float * 3),
                ("depth", _fp), ("normals", _fp)]


def cut_depth_maps(cams, depths, normals, device=0):
    """MeshGenerator::cut_depth_maps over all views on the device.  cams:
    objects with .flen, .R, .t; depths[i] (h, w) ray-length depth; normals[i]
    (h, w, 3) camera space.  Returns (cut depth maps, world-space normals)."""
    lib = _capi.load()
    n = len(cams)
    d = [_f32(x).copy() for x in depths]
    nm = [_f32(x).copy() for x in normals]
    arr = (MeshView * n)()
    for i in range(n):
        arr[i].height, arr[i].width = d[i].shape
        arr[i].flen = float(cams[i].flen)
        for k, x in enumerate(np.asarray(cams[i].R, dtype=np.float32).reshape(9)):
            arr[i].rot[k] = float(x)
        for k, x in enumerate(np.asarray(cams[i].t, dtype=np.float32).reshape(3)):
            arr[i].trans[k] = float(x)
        arr[i].depth = _p(d[i], _fp)
        arr[i].normals = _p(nm[i], _fp)
    check(lib.smvs_cut_depth_maps(device, arr, n))
    return d, nm
