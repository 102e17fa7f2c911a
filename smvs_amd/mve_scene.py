"""Writer / reader for the MVE scene files the host mirror consumes
(csrc/host/scene_io.cc): views/view_XXXX.mve/{meta.ini, <embedding>.mvei},
synth_0.out.  Test and demo tooling: it turns smvs_amd.synth scenes into a
directory smvs_host_reconstruct_scene can run on.  Formats as remembered from
MVE ([MVE-unverified], tests/golden/README.md)."""
import os
import struct

import numpy as np

SIGNATURE = b"\x89MVE_IMAGE\n"
TYPE_UINT8, TYPE_FLOAT = 1, 9


def save_mvei(path, array):
    a = np.ascontiguousarray(array)
    if a.ndim == 2:
        a = a[:, :, None]
    if a.dtype == np.uint8:
        t = TYPE_UINT8
    elif a.dtype == np.float32:
        t = TYPE_FLOAT
    else:
        raise ValueError("mvei: uint8 or float32 only")
    h, w, c = a.shape
    with open(path, "wb") as f:
        f.write(SIGNATURE)
        f.write(struct.pack("<iiii", w, h, c, t))
        f.write(a.tobytes())


def load_mvei(path):
    with open(path, "rb") as f:
        if f.read(len(SIGNATURE)) != SIGNATURE:
            raise ValueError("not an .mvei file: %s" % path)
        w, h, c, t = struct.unpack("<iiii", f.read(16))
        dtype = {TYPE_UINT8: np.uint8, TYPE_FLOAT: np.float32}[t]
        data = np.frombuffer(f.read(), dtype=dtype)
    return data.reshape(h, w, c) if c > 1 else data.reshape(h, w)


def write_view(scene_dir, view_id, cam, image_u8, embedding="undistorted"):
    d = os.path.join(scene_dir, "views", "view_%04d.mve" % view_id)
    os.makedirs(d, exist_ok=True)
    R = np.asarray(cam.R, dtype=np.float32).reshape(9)
    t = np.asarray(cam.t, dtype=np.float32).reshape(3)
    with open(os.path.join(d, "meta.ini"), "w") as f:
        f.write("# MVE view meta data is stored in INI-file syntax.\n\n[camera]\n")
        f.write("focal_length = %.9g\npixel_aspect = 1\nprincipal_point = 0.5 0.5\n"
                % float(np.float32(cam.flen)))
        f.write("rotation = %s\n" % " ".join("%.9g" % float(x) for x in R))
        f.write("translation = %s\n\n" % " ".join("%.9g" % float(x) for x in t))
        f.write("[view]\nid = %d\nname = %04d\n" % (view_id, view_id))
    save_mvei(os.path.join(d, embedding + ".mvei"), np.asarray(image_u8, dtype=np.uint8))
    return d


def write_bundle(scene_dir, cams, features, refs):
    """features (F, 3); refs[f] = list of view ids that see feature f."""
    with open(os.path.join(scene_dir, "synth_0.out"), "w") as f:
        f.write("drews 1.0\n%d %d\n" % (len(cams), len(features)))
        for cam in cams:
            R = np.asarray(cam.R, dtype=np.float64).reshape(3, 3)
            t = np.asarray(cam.t, dtype=np.float64).reshape(3)
            f.write("%.9g 0 0\n" % float(cam.flen))
            for row in R:
                f.write("%.9g %.9g %.9g\n" % tuple(row))
            f.write("%.9g %.9g %.9g\n" % tuple(t))
        for p, r in zip(np.asarray(features, dtype=np.float32), refs):
            f.write("%.9g %.9g %.9g\n255 255 255\n" % tuple(float(x) for x in p))
            f.write("%d %s\n" % (len(r), " ".join("%d %d 0" % (v, i) for i, v in enumerate(r))))


def write_scene(scene_dir, inputs, embedding="undistorted"):
    """A smvs_amd.synth.pipeline_inputs dict as an MVE scene (view ids = list
    positions; every feature seen by every view)."""
    os.makedirs(os.path.join(scene_dir, "views"), exist_ok=True)
    for i, (cam, img) in enumerate(zip(inputs["cams"], inputs["images"])):
        write_view(scene_dir, i, cam, img, embedding)
    n = len(inputs["cams"])
    write_bundle(scene_dir, inputs["cams"], inputs["features"],
                 [list(range(n))] * len(inputs["features"]))
    return scene_dir
