"""Writer / reader for the MVE scene files the host mirror consumes
(csrc/host/scene_io.cc): views/view_XXXX.mve/{meta.ini, <embedding>.mvei},
synth_0.out.  Test and demo tooling: it turns smvs_amd.synth scenes into a
directory smvs_host_reconstruct_scene can run on.  Formats as remembered from
MVE ([MVE-unverified], tests/golden/README.md)."""
import os
import struct
import zlib

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


# ------------------------------------------------------------------ PNG
# An independent (pure Python + zlib) writer / reader of the PNG container, the
# counterpart of csrc/host/png_io.cc in the tests: every scanline filter, the
# colour types with 8-bit samples, Adam7.
_PNG_SIG = b"\x89PNG\r\n\x1a\n"
_ADAM7 = [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2),
          (0, 1, 1, 2)]


def _png_chunk(kind, data):
    return struct.pack(">I", len(data)) + kind + data \
        + struct.pack(">I", zlib.crc32(kind + data) & 0xFFFFFFFF)


def _paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)


def _filter_rows(rows, bpp, filters):
    """rows: (n, stride) uint8 -> filtered bytes; filters: per-row type or None
    (cycle through all five)."""
    out = bytearray()
    prev = np.zeros(rows.shape[1], np.int32)
    for y, row in enumerate(rows.astype(np.int32)):
        ft = filters[y % len(filters)]
        left = np.concatenate([np.zeros(bpp, np.int32), row[:-bpp]]) if bpp < row.size \
            else np.zeros_like(row)
        upleft = np.concatenate([np.zeros(bpp, np.int32), prev[:-bpp]]) if bpp < row.size \
            else np.zeros_like(row)
        if ft == 0:
            pred = np.zeros_like(row)
        elif ft == 1:
            pred = left
        elif ft == 2:
            pred = prev
        elif ft == 3:
            pred = (left + prev) >> 1
        else:
            pred = np.array([_paeth(int(a), int(b), int(c))
                             for a, b, c in zip(left, prev, upleft)], np.int32)
        out.append(ft)
        out += ((row - pred) & 255).astype(np.uint8).tobytes()
        prev = row
    return bytes(out)


def save_png(path, array, filters=(0, 1, 2, 3, 4), interlace=False, palette=False,
             idat_chunk=1 << 20):
    """8-bit PNG of an (h, w) / (h, w, c) uint8 array, c in 1..4; `palette`
    stores an RGB image with at most 256 colours as colour type 3."""
    a = np.ascontiguousarray(array, dtype=np.uint8)
    if a.ndim == 2:
        a = a[:, :, None]
    h, w, c = a.shape
    plte = b""
    if palette:
        colours, index = np.unique(a.reshape(-1, 3), axis=0, return_inverse=True)
        if c != 3 or len(colours) > 256:
            raise ValueError("palette PNG: RGB with at most 256 colours")
        plte = _png_chunk(b"PLTE", colours.astype(np.uint8).tobytes())
        a = index.reshape(h, w, 1).astype(np.uint8)
        ctype, bpp = 3, 1
    else:
        ctype, bpp = {1: 0, 2: 4, 3: 2, 4: 6}[c], c
    raw = b""
    for (x0, y0, dx, dy) in (_ADAM7 if interlace else [(0, 0, 1, 1)]):
        sub = a[y0::dy, x0::dx]
        if sub.shape[0] and sub.shape[1]:
            raw += _filter_rows(sub.reshape(sub.shape[0], -1), bpp, list(filters))
    z = zlib.compress(raw, 6)
    with open(path, "wb") as f:
        f.write(_PNG_SIG)
        f.write(_png_chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0,
                                                1 if interlace else 0)))
        f.write(plte)
        for i in range(0, len(z), idat_chunk):
            f.write(_png_chunk(b"IDAT", z[i:i + idat_chunk]))
        f.write(_png_chunk(b"IEND", b""))


def load_png(path):
    """Reader for the non-interlaced files csrc/host/png_io.cc writes."""
    data = open(path, "rb").read()
    assert data[:8] == _PNG_SIG
    pos, idat, hdr = 8, b"", None
    while pos < len(data):
        n, kind = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0] \
            == zlib.crc32(kind + body) & 0xFFFFFFFF
        if kind == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif kind == b"IDAT":
            idat += body
        pos += 12 + n
    w, h, depth, ctype, _, _, interlace = hdr
    assert depth == 8 and interlace == 0
    c = {0: 1, 2: 3, 4: 2, 6: 4}[ctype]
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, w * c + 1)
    assert (raw[:, 0] == 0).all()      # the writer uses filter type 0 only
    out = raw[:, 1:].reshape(h, w, c)
    return out if c > 1 else out.reshape(h, w)


def write_view(scene_dir, view_id, cam, image_u8, embedding="undistorted", container="mvei"):
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
    if container == "png":      # what makescene leaves: <embedding>.png
        save_png(os.path.join(d, embedding + ".png"), np.asarray(image_u8, dtype=np.uint8))
    elif container == "jpg":    # ... and original.jpg, the camera's own file
        from PIL import Image   # (test helper: Pillow is the independent codec of the tests)
        Image.fromarray(np.asarray(image_u8, dtype=np.uint8)).save(
            os.path.join(d, embedding + ".jpg"), format="JPEG", quality=95, subsampling=2)
    else:
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


def write_scene(scene_dir, inputs, embedding="undistorted", container="mvei"):
    """A smvs_amd.synth.pipeline_inputs dict as an MVE scene (view ids = list
    positions; every feature seen by every view)."""
    os.makedirs(os.path.join(scene_dir, "views"), exist_ok=True)
    for i, (cam, img) in enumerate(zip(inputs["cams"], inputs["images"])):
        write_view(scene_dir, i, cam, img, embedding, container)
    n = len(inputs["cams"])
    write_bundle(scene_dir, inputs["cams"], inputs["features"],
                 [list(range(n))] * len(inputs["features"]))
    return scene_dir
