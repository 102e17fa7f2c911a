"""ctypes handle on the C++ host mirror (csrc/host/libsmvs_host.so):
smvs_amd::DepthOptimizer::optimize and the SGM initialisation, driven with
the inputs of smvs_amd.synth.pipeline_inputs()."""
import ctypes as C
import os

import numpy as np

from . import _capi

HOST_LIB = os.path.join(_capi.HERE, "csrc", "host", "libsmvs_host.so")
_lib = None

_fp = C.POINTER(C.c_float)
_u8p = C.POINTER(C.c_uint8)
_i32p = C.POINTER(C.c_int32)


class HostView(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("channels", C.c_int),
                ("bytes", _u8p), ("flen", C.c_float), ("rot", C.c_float * 9),
                ("trans", C.c_float * 3), ("view_id", C.c_int)]


class HostBundle(C.Structure):
    _fields_ = [("num_features", C.c_int), ("positions", _fp),
                ("ref_offsets", _i32p), ("ref_views", _i32p)]


class HostOptions(C.Structure):
    _fields_ = [("regularization", C.c_double),
                ("light_surf_regularization", C.c_double),
                ("num_iterations", C.c_int), ("min_scale", C.c_int),
                ("use_shading", C.c_int), ("use_sgm", C.c_int),
                ("full_optimization", C.c_int), ("device", C.c_int),
                ("solver", C.c_int), ("gamma_correction", C.c_int)]


class HostLog(C.Structure):
    _fields_ = [("count", C.c_int), ("scale", C.c_int * 256),
                ("iter", C.c_int * 256), ("newton_steps", C.c_int * 256),
                ("valid_patches", C.c_int * 256), ("cg_iterations", C.c_int * 256),
                ("active_patch_steps", C.c_longlong * 256),
                ("loop_seconds", C.c_double * 256),
                ("has_lighting", C.c_int), ("lighting", C.c_double * 16)]


def load():
    global _lib
    if _lib is None:
        _capi.load()  # libsmvs_hip.so first
        if not os.path.exists(HOST_LIB):
            raise _capi.SmvsError(-2, "host library %s is missing" % HOST_LIB)
        _lib = C.CDLL(HOST_LIB)
        _lib.smvs_host_last_error.restype = C.c_char_p
    return _lib


def _view(img, cam, view_id, keep):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    if img.ndim == 2:
        img = img[:, :, None]
    keep.append(img)
    v = HostView()
    v.height, v.width, v.channels = img.shape
    v.bytes = img.ctypes.data_as(_u8p)
    v.flen = cam.flen
    for i, x in enumerate(np.asarray(cam.R, dtype=np.float32).reshape(9)):
        v.rot[i] = float(x)
    for i, x in enumerate(np.asarray(cam.t, dtype=np.float32).reshape(3)):
        v.trans[i] = float(x)
    v.view_id = view_id
    return v


def _marshal(inputs, keep):
    cams, images = inputs["cams"], inputs["images"]
    main = _view(images[0], cams[0], inputs["view_ids"][0], keep)
    n_subs = len(cams) - 1
    subs = (HostView * n_subs)()
    for j in range(n_subs):
        subs[j] = _view(images[j + 1], cams[j + 1], inputs["view_ids"][j + 1], keep)
    feats = np.ascontiguousarray(inputs["features"], dtype=np.float32).reshape(-1, 3)
    nf = feats.shape[0]
    offsets = (np.arange(nf + 1) * len(cams)).astype(np.int32)
    refs = np.tile(np.asarray(inputs["view_ids"], dtype=np.int32), nf)
    keep += [feats, offsets, refs]
    b = HostBundle(nf, feats.ctypes.data_as(_fp), offsets.ctypes.data_as(_i32p),
                   refs.ctypes.data_as(_i32p))
    return main, subs, n_subs, b


def optimize(inputs, regularization=0.01, light_reg=0.0, num_iterations=5,
             min_scale=2, use_shading=False, sgm_depth=None,
             full_optimization=False, device=0, solver="auto", want_maps=True,
             gamma_correction=False):
    """want_maps=False: optimize() only -- the depth / normal maps the reference
    reads back with get_depth() / get_normals() afterwards are not fetched (the
    embeddings optimize() itself writes are; tools/optimize_timeline.py)."""
    lib = load()
    keep = []
    main, subs, n_subs, b = _marshal(inputs, keep)
    o = HostOptions(regularization, light_reg, num_iterations, min_scale,
                    1 if use_shading else 0, 1 if sgm_depth is not None else 0,
                    1 if full_optimization else 0, device,
                    dict(auto=0, streaming=1, resident_ref=2)[solver],
                    1 if gamma_correction else 0)
    h, w = main.height, main.width
    depth = np.zeros((h, w), dtype=np.float32)
    normals = np.zeros((h, w, 3), dtype=np.float32)
    log = HostLog()
    sd = rt = None
    sw = sh = 0
    if sgm_depth is not None:
        sd = np.ascontiguousarray(sgm_depth, dtype=np.float32)
        sh, sw = sd.shape
        rt = np.zeros_like(sd)
    rc = lib.smvs_host_optimize(C.byref(main), subs, n_subs, C.byref(b),
        sd.ctypes.data_as(_fp) if sd is not None else None, sw, sh,
        rt.ctypes.data_as(_fp) if rt is not None else None, C.byref(o),
        depth.ctypes.data_as(_fp) if want_maps else None,
        normals.ctypes.data_as(_fp) if want_maps else None, C.byref(log))
    if rc != 0:
        raise _capi.SmvsError(rc, lib.smvs_host_last_error().decode())
    steps = [dict(scale=log.scale[i], iter=log.iter[i],
                  newton_steps=log.newton_steps[i],
                  valid_patches=log.valid_patches[i],
                  cg_iterations=log.cg_iterations[i],
                  active_patch_steps=log.active_patch_steps[i],
                  loop_seconds=log.loop_seconds[i]) for i in range(log.count)]
    return dict(depth=depth if want_maps else None,
                normals=normals if want_maps else None, log=steps, sgm_roundtrip=rt,
                lighting=np.array(log.lighting[:]) if log.has_lighting else None)


def optimize_views(inputs, n_jobs, regularization=0.01, light_reg=0.0,
                   num_iterations=5, min_scale=2, use_shading=False, sgm_scale=None,
                   first_device=0, num_devices=1, views_in_flight=1, solver="auto",
                   keep_job=0):
    """n_jobs reference views (each job the same scene: main + neighbours of
    `inputs`) through smvs_amd::ViewQueue -- the per-view tasks of smvsrecon
    (app/smvsrecon.cc:658-733): StereoViews, SGM front end (sgm_scale is not
    None), DepthOptimizer::optimize, depth + normal maps.  Returns the maps of
    job `keep_job`, the per-job logs and wall times."""
    lib = load()
    keep = []
    main, subs, n_subs, b = _marshal(inputs, keep)
    mains = (HostView * n_jobs)(*([main] * n_jobs))
    all_subs = (HostView * (n_jobs * n_subs))()
    for j in range(n_jobs):
        for k in range(n_subs):
            all_subs[j * n_subs + k] = subs[k]
    o = HostOptions(regularization, light_reg, num_iterations, min_scale,
                    1 if use_shading else 0, 1 if sgm_scale is not None else 0,
                    0, first_device, dict(auto=0, streaming=1, resident_ref=2)[solver], 0)
    h, w = main.height, main.width
    depth = np.zeros((h, w), dtype=np.float32)
    normals = np.zeros((h, w, 3), dtype=np.float32)
    secs = np.zeros(n_jobs)
    total = C.c_double(0.0)
    logs = (HostLog * n_jobs)()
    rc = lib.smvs_host_optimize_views(mains, all_subs, C.c_int(n_jobs), C.c_int(n_subs),
        C.byref(b), C.byref(o), C.c_int(-1 if sgm_scale is None else sgm_scale),
        C.c_int(first_device), C.c_int(num_devices), C.c_int(views_in_flight),
        C.c_int(keep_job), depth.ctypes.data_as(_fp), normals.ctypes.data_as(_fp),
        secs.ctypes.data_as(C.POINTER(C.c_double)), C.byref(total), logs)
    if rc != 0:
        raise _capi.SmvsError(rc, lib.smvs_host_last_error().decode())
    out_logs = [[dict(scale=l.scale[i], iter=l.iter[i], newton_steps=l.newton_steps[i],
                      valid_patches=l.valid_patches[i], cg_iterations=l.cg_iterations[i],
                      active_patch_steps=l.active_patch_steps[i],
                      loop_seconds=l.loop_seconds[i])
                 for i in range(l.count)] for l in logs]
    return dict(depth=depth, normals=normals, logs=out_logs, job_seconds=secs,
                total_seconds=total.value, views_per_s=n_jobs / total.value)


def gn_solve_step(inputs, init_scale, regularization=0.01, device=0):
    """One construct + solve through the compatibility classes
    smvs_amd::GaussNewtonStep / ConjugateGradient (reference signatures,
    lib/gauss_newton_step.h:40-50, lib/conjugate_gradient.h:55-56) on the
    surface Surface::create builds at init_scale.  Returns the planes and
    reprojections the step used (to feed the oracle the same inputs), the
    surface, g, x, H9, P and the solver's status."""
    lib = load()
    keep = []
    main, subs, n_subs, b = _marshal(inputs, keep)
    surf = surface_script(inputs, init_scale, [])
    h, w = main.height, main.width
    N = (surf["npx"] + 1) * (surf["npy"] + 1)
    mg = np.zeros((h, w, 2), np.float32)
    sg = np.zeros((n_subs, h, w, 2), np.float32)
    sh = np.zeros((n_subs, h, w, 3), np.float32)
    M = np.zeros((n_subs, 9)); t = np.zeros((n_subs, 3)); fl = np.zeros(2, np.float32)
    g = np.zeros(4 * N); x = np.zeros(4 * N); H9 = np.zeros((N, 9, 16)); P = np.zeros((N, 16))
    cg = np.zeros(2, np.int32)
    dp = C.POINTER(C.c_double)
    rc = lib.smvs_host_gn_solve_step(C.byref(main), subs, C.c_int(n_subs), C.byref(b),
        C.c_int(init_scale), C.c_double(regularization), C.c_int(device),
        mg.ctypes.data_as(_fp), sg.ctypes.data_as(_fp), sh.ctypes.data_as(_fp),
        M.ctypes.data_as(dp), t.ctypes.data_as(dp), fl.ctypes.data_as(_fp),
        g.ctypes.data_as(dp), x.ctypes.data_as(dp), H9.ctypes.data_as(dp),
        P.ctypes.data_as(dp), cg.ctypes.data_as(_i32p))
    if rc != 0:
        raise _capi.SmvsError(rc, lib.smvs_host_last_error().decode())
    surf.update(width=w, height=h,
                patch_vis=np.where(surf["patch_valid"] != 0, (1 << n_subs) - 1, 0).astype(np.uint32))
    views = dict(flen=float(fl[0]), inv_flen=float(fl[1]), grad=mg,
                 subs=[(sg[j], sh[j]) for j in range(n_subs)], M=M, t=t,
                 shading=None, shading_grad=None)
    return dict(surf=surf, views=views, g=g, x=x, H9=H9, P=P,
                iterations=int(cg[0]), info=int(cg[1]))


class ReconSettings(C.Structure):
    _fields_ = [("image_embedding", C.c_char_p), ("regularization", C.c_float),
                ("output_scale", C.c_int), ("use_shading", C.c_int), ("use_sgm", C.c_int),
                ("force_recon", C.c_int), ("force_sgm", C.c_int),
                ("full_optimization", C.c_int), ("sgm_min", C.c_float),
                ("sgm_max", C.c_float), ("sgm_scale", C.c_int), ("num_neighbors", C.c_int),
                ("min_neighbors", C.c_int), ("first_device", C.c_int),
                ("num_devices", C.c_int), ("views_in_flight", C.c_int),
                ("input_scale", C.c_int), ("max_pixels", C.c_int)]


def reconstruct_scene(scene_dir, view_ids=None, image_embedding="undistorted",
                      regularization=1.0, output_scale=2, use_shading=False, use_sgm=True,
                      force_recon=False, force_sgm=False, sgm_range=(0.0, 0.0), sgm_scale=1,
                      num_neighbors=6, min_neighbors=3, first_device=0, num_devices=1,
                      views_in_flight=2, input_scale=-1, max_pixels=1700000, details=False):
    """smvsrecon's scene-level run (app/smvsrecon.cc:400-745) through
    smvs_amd::reconstruct_scene: returns (reconstructed ids, skipped, seconds)
    [, input scale used if `details`].  input_scale < 0 (the default, as
    app/smvsrecon.cc:44): smvsrecon's automatic choice from max_pixels
    (:477-500); 0: full resolution; > 0: the views are read from the
    embedding undist-L<input_scale>, created with rescale_half_size_gaussian
    where missing (:621-650), and the outputs are named smvs-B<input_scale>."""
    lib = load()
    st = ReconSettings(image_embedding.encode(), regularization, output_scale,
                       1 if use_shading else 0, 1 if use_sgm else 0,
                       1 if force_recon else 0, 1 if force_sgm else 0, 0,
                       sgm_range[0], sgm_range[1], sgm_scale, num_neighbors, min_neighbors,
                       first_device, num_devices, views_in_flight, input_scale, max_pixels)
    ids = None if view_ids is None else np.asarray(view_ids, dtype=np.int32)
    # (room for every view of the scene: smvs_host_scene_info)
    cap = C.c_int(0)
    lib.smvs_host_scene_info(scene_dir.encode(), image_embedding.encode(), C.c_int(0),
                             C.byref(cap), None, None, None, None, None, None, None)
    out = np.zeros(max(cap.value, 1), np.int32)
    n = C.c_int(0); sk = C.c_int(0); secs = C.c_double(0.0); used = C.c_int(0)
    rc = lib.smvs_host_reconstruct_scene(scene_dir.encode(), C.byref(st),
        ids.ctypes.data_as(_i32p) if ids is not None else None,
        C.c_int(0 if ids is None else ids.size), out.ctypes.data_as(_i32p),
        C.c_int(out.size), C.byref(n), C.byref(sk), C.byref(secs), C.byref(used))
    if rc != 0:
        raise _capi.SmvsError(rc, lib.smvs_host_last_error().decode())
    res = ([int(x) for x in out[:min(n.value, out.size)]], sk.value, secs.value)
    return res + (used.value,) if details else res


def load_byte_image(path):
    """A u8 image embedding (.png via csrc/host/png_io.cc, or .mvei)."""
    lib = load()
    whc = (C.c_int * 3)()
    rc = lib.smvs_host_load_byte_image(path.encode(), whc, None, C.c_size_t(0))
    if rc != 0:
        raise _capi.SmvsError(rc, lib.smvs_host_last_error().decode())
    w, h, c = whc[0], whc[1], whc[2]
    out = np.zeros((h, w, c), np.uint8)
    rc = lib.smvs_host_load_byte_image(path.encode(), whc, out.ctypes.data_as(_u8p),
                                       C.c_size_t(out.size))
    if rc != 0:
        raise _capi.SmvsError(rc, lib.smvs_host_last_error().decode())
    return out if c > 1 else out.reshape(h, w)


def save_png(path, array):
    lib = load()
    a = np.ascontiguousarray(array, dtype=np.uint8)
    if a.ndim == 2:
        a = a[:, :, None]
    rc = lib.smvs_host_save_png(path.encode(), a.ctypes.data_as(_u8p), C.c_int(a.shape[1]),
                                C.c_int(a.shape[0]), C.c_int(a.shape[2]))
    if rc != 0:
        raise _capi.SmvsError(rc, lib.smvs_host_last_error().decode())


def rescale_half_size_gaussian(array):
    """mve::image::rescale_half_size_gaussian<uint8_t> of the host mirror."""
    lib = load()
    a = np.ascontiguousarray(array, dtype=np.uint8)
    squeeze = a.ndim == 2
    if squeeze:
        a = a[:, :, None]
    h, w, c = a.shape
    out = np.zeros(((h + 1) // 2, (w + 1) // 2, c), np.uint8)
    rc = lib.smvs_host_rescale_half_size_gaussian(a.ctypes.data_as(_u8p), C.c_int(w),
                                                  C.c_int(h), C.c_int(c),
                                                  out.ctypes.data_as(_u8p))
    if rc != 0:
        raise _capi.SmvsError(rc, lib.smvs_host_last_error().decode())
    return out[:, :, 0] if squeeze else out


def scene_info(scene_dir, image_embedding="undistorted", max_views=4096):
    """Scene::create + bundle of csrc/host/scene_io.cc (no device)."""
    lib = load()
    n = C.c_int(0); nf = C.c_int(0)
    present = np.zeros(max_views, np.int32); flen = np.zeros(max_views, np.float32)
    rot = np.zeros((max_views, 9), np.float32); trans = np.zeros((max_views, 3), np.float32)
    w = np.zeros(max_views, np.int32); h = np.zeros(max_views, np.int32)
    rc = lib.smvs_host_scene_info(scene_dir.encode(), image_embedding.encode(),
        C.c_int(max_views), C.byref(n), present.ctypes.data_as(_i32p),
        flen.ctypes.data_as(_fp), rot.ctypes.data_as(_fp), trans.ctypes.data_as(_fp),
        w.ctypes.data_as(_i32p), h.ctypes.data_as(_i32p), C.byref(nf))
    if rc != 0:
        raise _capi.SmvsError(rc, lib.smvs_host_last_error().decode())
    k = n.value
    return dict(present=present[:k], flen=flen[:k], rot=rot[:k], trans=trans[:k],
                width=w[:k], height=h[:k], n_features=nf.value)


def mvei_roundtrip(in_path, out_path):
    lib = load()
    if lib.smvs_host_mvei_roundtrip(in_path.encode(), out_path.encode()) != 0:
        raise _capi.SmvsError(-1, lib.smvs_host_last_error().decode())


def view_queue_selftest(n_tasks, num_devices, views_in_flight, throwing_task=-1):
    """smvs_amd::ViewQueue without a device: -> (tasks per device, per worker)."""
    lib = load()
    dev = np.zeros(num_devices, np.int32)
    wrk = np.zeros(num_devices * views_in_flight, np.int32)
    rc = lib.smvs_host_view_queue_selftest(C.c_int(n_tasks), C.c_int(num_devices),
        C.c_int(views_in_flight), C.c_int(throwing_task), dev.ctypes.data_as(_i32p),
        wrk.ctypes.data_as(_i32p))
    if rc != 0:
        raise _capi.SmvsError(rc, lib.smvs_host_last_error().decode())
    return dev, wrk


def sgm_depth(inputs, sgm_scale=1, min_depth=0.0, max_depth=0.0, device=0):
    lib = load()
    keep = []
    main, subs, n_subs, b = _marshal(inputs, keep)
    w, h = main.width, main.height
    for _ in range(sgm_scale):
        w, h = (w + 1) // 2, (h + 1) // 2
    out = np.zeros((h, w), dtype=np.float32)
    ow = C.c_int(0); oh = C.c_int(0)
    rc = lib.smvs_host_sgm_depth(C.byref(main), subs, n_subs, C.byref(b),
        sgm_scale, C.c_float(min_depth), C.c_float(max_depth), device,
        out.ctypes.data_as(_fp), C.byref(ow), C.byref(oh))
    if rc != 0:
        raise _capi.SmvsError(rc, lib.smvs_host_last_error().decode())
    assert (ow.value, oh.value) == (w, h)
    return out


def select_neighbors(scene, view, num_neighbors=6, use_bundle=True):
    """smvs_amd::ViewSelection (csrc/host/view_selection.cc, mirror of
    lib/view_selection.cc:14-161) on a scene description: dict(views=[dict(
    present, id, flen, rot, trans, has_image, width, height)], features=(F, 3),
    refs=[view ids per feature]).  Returns indices into the view list."""
    lib = load()
    keep = []
    n = len(scene["views"])
    views = (HostView * n)()
    dummy = np.zeros(1, dtype=np.uint8)
    for i, v in enumerate(scene["views"]):
        present = v.get("present", True)
        views[i].width = int(v["width"]) if present else 0
        views[i].height = int(v["height"]) if present else 0
        views[i].channels = 1
        views[i].bytes = dummy.ctypes.data_as(_u8p) if v.get("has_image", True) else None
        views[i].flen = float(v["flen"])
        for k, x in enumerate(np.asarray(v["rot"], dtype=np.float32).reshape(9)):
            views[i].rot[k] = float(x)
        for k, x in enumerate(np.asarray(v["trans"], dtype=np.float32).reshape(3)):
            views[i].trans[k] = float(x)
        views[i].view_id = int(v["id"])
    bundle = None
    if use_bundle:
        feats = np.ascontiguousarray(scene["features"], dtype=np.float32).reshape(-1, 3)
        offsets = np.zeros(len(scene["refs"]) + 1, dtype=np.int32)
        offsets[1:] = np.cumsum([len(r) for r in scene["refs"]])
        flat = np.asarray([i for r in scene["refs"] for i in r], dtype=np.int32)
        if flat.size == 0:
            flat = np.zeros(1, dtype=np.int32)
        keep += [feats, offsets, flat]
        bundle = HostBundle(feats.shape[0], feats.ctypes.data_as(_fp),
                            offsets.ctypes.data_as(_i32p), flat.ctypes.data_as(_i32p))
    out = np.zeros(max(n, 1), dtype=np.int32)
    n_out = C.c_int(0)
    rc = lib.smvs_host_select_neighbors(views, C.c_int(n),
        C.byref(bundle) if bundle is not None else None, C.c_int(view),
        C.c_int(num_neighbors), out.ctypes.data_as(_i32p), C.byref(n_out))
    if rc != 0:
        raise _capi.SmvsError(rc, lib.smvs_host_last_error().decode())
    return [int(x) for x in out[:n_out.value]]


def surface_script(inputs, init_scale, ops, init_depth=None, delete_every=3, device=None):
    """smvs_amd::Surface on its own (csrc/host/surface.cc, mirror of
    lib/surface.cc): create + a script of operations (1 expand,
    2 subdivide_patches, 3 fill_patches_from_depth, 4 remove_isolated_patches,
    5 delete every delete_every-th valid patch + remove_nodes_without_patch).
    device=None: the host mirror, no device involved; device=k: the same
    script on the surface of a device context (csrc/surface.hip), result
    downloaded (plus valid_patches as the last operation reported it)."""
    lib = load()
    keep = []
    main, _, _, bundle = _marshal(inputs, keep)
    h, w = np.asarray(inputs["images"][0]).shape[:2]
    cap_n, cap_p = (w + 2) * (h + 2), (w + 1) * (h + 1)
    nodes = np.zeros(cap_n * 4); nv = np.zeros(cap_n, np.uint8); pv = np.zeros(cap_p, np.uint8)
    info = np.zeros(6, np.int32)
    ops_a = np.asarray(list(ops) + [0], dtype=np.int32)
    depth = None if init_depth is None else np.ascontiguousarray(init_depth, dtype=np.float32)
    dptr = depth.ctypes.data_as(_fp) if depth is not None else None
    if device is None:
        rc = lib.smvs_host_surface_script(C.byref(main), C.byref(bundle), dptr,
            C.c_int(init_scale), ops_a.ctypes.data_as(_i32p), C.c_int(len(ops)),
            C.c_int(delete_every), info.ctypes.data_as(_i32p),
            nodes.ctypes.data_as(C.POINTER(C.c_double)), nv.ctypes.data_as(_u8p),
            pv.ctypes.data_as(_u8p))
    else:
        rc = lib.smvs_host_surface_script_device(C.byref(main), C.byref(bundle), dptr,
            C.c_int(init_scale), ops_a.ctypes.data_as(_i32p), C.c_int(len(ops)),
            C.c_int(delete_every), C.c_int(device), info.ctypes.data_as(_i32p),
            nodes.ctypes.data_as(C.POINTER(C.c_double)), nv.ctypes.data_as(_u8p),
            pv.ctypes.data_as(_u8p))
    if rc != 0:
        raise _capi.SmvsError(rc, lib.smvs_host_last_error().decode())
    scale, npx, npy, sx, sy = (int(x) for x in info[:5])
    nn, npatch = (npx + 1) * (npy + 1), npx * npy
    out = dict(scale=scale, npx=npx, npy=npy, start_x=sx, start_y=sy,
               nodes=nodes[:4 * nn].reshape(nn, 4).copy(), node_valid=nv[:nn].copy(),
               patch_valid=pv[:npatch].copy())
    if device is not None:
        out["valid_patches"] = int(info[5])
    return out


def surface_maps(inputs, init_scale, init_depth=None, device=0):
    """DepthOptimizer(main, subs, surface, opts).get_depth() / get_normals()
    without optimize() (lib/depth_optimizer.h:53-61)."""
    lib = load()
    keep = []
    main, subs, n_subs, b = _marshal(inputs, keep)
    h, w = main.height, main.width
    depth = np.zeros((h, w), dtype=np.float32)
    normals = np.zeros((h, w, 3), dtype=np.float32)
    init = None if init_depth is None else np.ascontiguousarray(init_depth, dtype=np.float32)
    rc = lib.smvs_host_surface_maps(C.byref(main), subs, n_subs, C.byref(b),
        init.ctypes.data_as(_fp) if init is not None else None, C.c_int(init_scale),
        C.c_int(device), depth.ctypes.data_as(_fp), normals.ctypes.data_as(_fp))
    if rc != 0:
        raise _capi.SmvsError(rc, lib.smvs_host_last_error().decode())
    return depth, normals


def depth_range(inputs, view_index=0):
    """SGMStereo::fill_depth_range_for_view on the host mirror."""
    lib = load()
    keep = []
    _, _, _, bundle = _marshal(inputs, keep)
    v = _view(inputs["images"][view_index], inputs["cams"][view_index],
              inputs["view_ids"][view_index], keep)
    out = np.zeros(2, np.float32)
    if lib.smvs_host_depth_range(C.byref(v), C.byref(bundle), out.ctypes.data_as(_fp)) != 0:
        raise _capi.SmvsError(-1, lib.smvs_host_last_error().decode())
    return out


def view_reprojection(inputs, src, dst):
    """CameraInfo::fill_reprojection between two views at their image sizes."""
    lib = load()
    keep = []
    a = _view(inputs["images"][src], inputs["cams"][src], inputs["view_ids"][src], keep)
    b = _view(inputs["images"][dst], inputs["cams"][dst], inputs["view_ids"][dst], keep)
    M = np.zeros(9, np.float32); t = np.zeros(3, np.float32)
    if lib.smvs_host_reprojection(C.byref(a), C.byref(b), M.ctypes.data_as(_fp),
                                  t.ctypes.data_as(_fp)) != 0:
        raise _capi.SmvsError(-1, lib.smvs_host_last_error().decode())
    return M, t


def sgm_image(inputs, view_index=0, halvings=1):
    """StereoView::get_byte_image + rescale_half_size (the SGM input image)."""
    lib = load()
    keep = []
    v = _view(inputs["images"][view_index], inputs["cams"][view_index],
              inputs["view_ids"][view_index], keep)
    out = np.zeros(v.width * v.height, np.uint8)
    w, h = C.c_int(0), C.c_int(0)
    if lib.smvs_host_sgm_image(C.byref(v), C.c_int(halvings), out.ctypes.data_as(_u8p),
                               C.byref(w), C.byref(h)) != 0:
        raise _capi.SmvsError(-1, lib.smvs_host_last_error().decode())
    return out[:w.value * h.value].reshape(h.value, w.value).copy()


def block_multiply(H9, node_stride, x):
    """smvs_amd::BlockStencilMatrix::multiply (BlockSparseMatrix<4>::multiply,
    lib/block_sparse_matrix.h:276-298) on the host: y = A x for the block
    stencil H9 [N][9][16].  No device involved."""
    lib = load()
    H9 = np.ascontiguousarray(H9, dtype=np.float64)
    x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1)
    n = x.size // 4
    y = np.zeros(4 * n)
    dp = C.POINTER(C.c_double)
    rc = lib.smvs_host_block_multiply(C.c_int(n), C.c_int(node_stride),
                                      H9.ctypes.data_as(dp), x.ctypes.data_as(dp),
                                      y.ctypes.data_as(dp))
    if rc != 0:
        raise _capi.SmvsError(rc, lib.smvs_host_last_error().decode())
    return y


# ------------------------------------------------------------ loop recording
def record_loops(on=True):
    """smvs_host_record_loops: while on, every Newton batch of this thread's
    optimize() calls leaves a device-resident clone of its loop's start state
    (smvs_ctx_clone_loop_state); starting a recording drops what the recorder
    still owns."""
    load().smvs_host_record_loops(1 if on else 0)


def take_recorded_loops():
    """-> [dict(handle, params, scale, iter)] of this thread's recording; the
    contexts now belong to the caller (smvs_ctx_destroy each)."""
    lib = load()
    n = lib.smvs_host_recorded_loops(None, None, None, None, 0)
    if n == 0:
        return []
    ctxs = (C.c_void_p * n)()
    prm = (_capi.LoopParams * n)()
    scales = (C.c_int * n)()
    iters = (C.c_int * n)()
    lib.smvs_host_recorded_loops(ctxs, prm, scales, iters, n)
    lib.smvs_host_release_recorded_loops(0)
    return [dict(handle=C.c_void_p(ctxs[i]), params=prm[i], scale=scales[i], iter=iters[i])
            for i in range(n)]


# ---------------------------------------------------------------- embeddings
def last_embeddings():
    """{name: array} of what the last optimize() of this thread wrote into its
    main view (smvs_host_embedding_names / smvs_host_embedding): the results and,
    with SMVS_DEBUG_LVL >= 2, the reference's intermediate embeddings."""
    lib = load()
    buf = C.create_string_buffer(4096)
    lib.smvs_host_embedding_names(buf, 4096)
    out = {}
    for name in buf.value.decode().split("\n"):
        if not name:
            continue
        whc = (C.c_int * 3)()
        n = lib.smvs_host_embedding(name.encode(), None, C.c_longlong(0), whc)
        if n <= 0:
            continue
        a = np.zeros(n, dtype=np.float32)
        if lib.smvs_host_embedding(name.encode(), a.ctypes.data_as(_fp), C.c_longlong(n), whc) == 0:
            out[name] = a.reshape(whc[1], whc[0], whc[2]).squeeze()
    return out
