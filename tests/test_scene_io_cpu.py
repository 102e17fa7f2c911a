"""MVE scene I/O of the host mirror (csrc/host/scene_io.cc):
meta.ini, synth_0.out and .mvei files written by smvs_amd/mve_scene.py are read
back by the C++ readers.  Every format detail is [MVE-unverified] (MVE is not
in the reference tree); what these tests pin is that writer and reader agree
and that the reader reproduces MVE's list semantics (index = view id, holes =
null views)."""
import os

import numpy as np
import pytest

from smvs_amd import synth, mve_scene


@pytest.fixture(scope="module")
def scene(tmp_path_factory):
    from smvs_amd import host
    host.load()
    d = str(tmp_path_factory.mktemp("scene"))
    inputs = synth.pipeline_inputs("plane", 96, 64, 3, n_features=50)
    mve_scene.write_scene(d, inputs)
    return d, inputs


def test_scene_views_and_bundle_are_read_back(scene):
    from smvs_amd import host
    d, inputs = scene
    info = host.scene_info(d)
    assert info["present"].tolist() == [1, 1, 1, 1]
    assert info["n_features"] == 50
    for i, cam in enumerate(inputs["cams"]):
        assert info["flen"][i] == np.float32(cam.flen)
        assert np.array_equal(info["rot"][i], np.asarray(cam.R, np.float32).reshape(9))
        assert np.array_equal(info["trans"][i], np.asarray(cam.t, np.float32).reshape(3))
        assert (info["width"][i], info["height"][i]) == (96, 64)


def test_view_list_index_is_the_view_id(tmp_path):
    """A scene with ids 0, 2, 5: the list has six entries, three of them null
    (mve::Scene::init_views); a view without the embedding reports size 0."""
    from smvs_amd import host
    inputs = synth.pipeline_inputs("plane", 64, 48, 2, n_features=10)
    d = str(tmp_path)
    for vid, k in ((0, 0), (2, 1), (5, 2)):
        mve_scene.write_view(d, vid, inputs["cams"][k], inputs["images"][k])
    os.remove(os.path.join(d, "views", "view_0002.mve", "undistorted.mvei"))
    info = host.scene_info(d)
    assert info["present"].tolist() == [1, 0, 1, 0, 0, 1]
    assert info["width"].tolist() == [64, 0, 0, 0, 0, 64]
    assert info["n_features"] == -1          # no synth_0.out
    with pytest.raises(Exception):
        host.scene_info(os.path.join(d, "nowhere"))


@pytest.mark.parametrize("dtype,channels", [(np.uint8, 3), (np.uint8, 1), (np.float32, 1),
                                            (np.float32, 3)])
def test_mvei_round_trip(tmp_path, dtype, channels):
    from smvs_amd import host
    rng = np.random.default_rng(3)
    a = (rng.random((37, 53, channels)) * 255).astype(dtype)
    if channels == 1:
        a = a[:, :, 0]
    src, dst = str(tmp_path / "a.mvei"), str(tmp_path / "b.mvei")
    mve_scene.save_mvei(src, a)
    host.mvei_roundtrip(src, dst)
    assert open(src, "rb").read() == open(dst, "rb").read()
    assert np.array_equal(mve_scene.load_mvei(dst), a)
    with open(str(tmp_path / "bad.mvei"), "wb") as f:
        f.write(b"not an image")
    with pytest.raises(Exception):
        host.mvei_roundtrip(str(tmp_path / "bad.mvei"), dst)


# ------------------------------------------------------------------ PNG
# csrc/host/png_io.cc (reader / writer on zlib) against the independent
# pure-Python codec of smvs_amd/mve_scene.py.
@pytest.mark.parametrize("channels", [1, 2, 3, 4])
@pytest.mark.parametrize("interlace", [False, True])
def test_png_reader_decodes_every_filter_and_layout(tmp_path, channels, interlace):
    """Scanline filters None / Sub / Up / Average / Paeth cycling over the rows,
    grey, grey + alpha, RGB, RGBA, plain and Adam7, IDAT split into many
    chunks, odd sizes (Adam7 passes that are empty)."""
    from smvs_amd import host
    rng = np.random.default_rng(10 * channels + interlace)
    for h, w in ((37, 53), (1, 1), (3, 2), (8, 9), (64, 5)):
        a = rng.integers(0, 256, size=(h, w, channels), dtype=np.uint8)
        if h > 8:
            a[: h // 3] = a[0:1]                      # long runs for the inflater
        p = str(tmp_path / ("f_%d_%d.png" % (h, w)))
        mve_scene.save_png(p, a, filters=(0, 1, 2, 3, 4), interlace=interlace, idat_chunk=300)
        got = host.load_byte_image(p)
        assert np.array_equal(got.reshape(a.shape), a), (h, w)
    for ft in range(5):                               # one filter type for all rows
        a = rng.integers(0, 256, size=(19, 23, channels), dtype=np.uint8)
        p = str(tmp_path / ("one_%d.png" % ft))
        mve_scene.save_png(p, a, filters=(ft,), interlace=interlace)
        assert np.array_equal(host.load_byte_image(p).reshape(a.shape), a), ft


def test_png_palette_and_writer_and_errors(tmp_path):
    from smvs_amd import host
    rng = np.random.default_rng(77)
    pal = (rng.integers(0, 5, size=(20, 31, 3)) * 50).astype(np.uint8)
    p = str(tmp_path / "pal.png")
    mve_scene.save_png(p, pal, palette=True)
    assert np.array_equal(host.load_byte_image(p), pal)
    # the C++ writer, read back by the Python reader
    for c in (1, 2, 3, 4):
        a = rng.integers(0, 256, size=(33, 17, c), dtype=np.uint8)
        q = str(tmp_path / ("w%d.png" % c))
        host.save_png(q, a)
        assert np.array_equal(mve_scene.load_png(q).reshape(a.shape), a)
        assert np.array_equal(host.load_byte_image(q).reshape(a.shape), a)
    # corrupt files are errors, not garbage
    data = bytearray(open(p, "rb").read())
    data[40] ^= 0xFF                                     # breaks a chunk checksum
    bad = str(tmp_path / "bad.png")
    open(bad, "wb").write(bytes(data))
    with pytest.raises(Exception):
        host.load_byte_image(bad)
    open(bad, "wb").write(bytes(open(p, "rb").read()[:60]))   # truncated
    with pytest.raises(Exception):
        host.load_byte_image(bad)
    # 16-bit samples are refused (mve::ByteImage has none)
    import struct, zlib
    raw = b"".join(b"\x00" + bytes(2 * 4) for _ in range(3))
    sixteen = (b"\x89PNG\r\n\x1a\n" + mve_scene._png_chunk(b"IHDR", struct.pack(
        ">IIBBBBB", 4, 3, 16, 0, 0, 0, 0)) + mve_scene._png_chunk(b"IDAT", zlib.compress(raw))
        + mve_scene._png_chunk(b"IEND", b""))
    open(bad, "wb").write(sixteen)
    with pytest.raises(Exception):
        host.load_byte_image(bad)
    # a forged header (2^31 - 1 pixels a side, a few bytes of data) is refused
    # before any size is computed from it: no overflow, no multi-GB allocation
    for w, h in ((0x7FFFFFFF, 0x7FFFFFFF), (0x7FFFFFFF, 1), (1 << 21, 3), (1 << 16, 1 << 16)):
        forged = (b"\x89PNG\r\n\x1a\n" + mve_scene._png_chunk(b"IHDR", struct.pack(
            ">IIBBBBB", w, h, 8, 2, 0, 0, 1)) + mve_scene._png_chunk(b"IDAT", zlib.compress(raw))
            + mve_scene._png_chunk(b"IEND", b""))
        open(bad, "wb").write(forged)
        with pytest.raises(Exception, match="out of range"):
            host.load_byte_image(bad)


@pytest.mark.parametrize("shape", [(37, 53, 3), (64, 64, 1), (5, 2, 3), (101, 7, 1), (2, 2, 1)])
def test_rescale_half_size_gaussian_matches_oracle(oracle, shape):
    """mve::image::rescale_half_size_gaussian<uint8_t> [MVE-unverified M29]:
    the host mirror (4 x 4 weight table) against the oracle's restatement (the
    sixteen Accum::add calls written out), bit for bit, and both against the
    definition in float64 (a Gaussian-weighted mean of the clamped 4 x 4 window,
    sigma^2 = 0.75) to within the final rounding."""
    from smvs_amd import host
    rng = np.random.default_rng(shape[0] * 100 + shape[1])
    a = rng.integers(0, 256, size=shape, dtype=np.uint8)
    got = host.rescale_half_size_gaussian(a)
    want = oracle.rescale_half_size_gaussian(a)
    assert got.shape == ((shape[0] + 1) // 2, (shape[1] + 1) // 2, shape[2])
    assert np.array_equal(got, want)
    w = np.array([np.exp(-4.5 / 1.5), np.exp(-2.5 / 1.5), np.exp(-0.5 / 1.5)])
    k = np.array([[w[0], w[1], w[1], w[0]], [w[1], w[2], w[2], w[1]],
                  [w[1], w[2], w[2], w[1]], [w[0], w[1], w[1], w[0]]])
    pad = np.pad(a.astype(np.float64), ((1, 2), (1, 2), (0, 0)), mode="edge")
    for y in range(got.shape[0]):
        for x in range(got.shape[1]):
            ref = (pad[2 * y:2 * y + 4, 2 * x:2 * x + 4] * k[:, :, None]).sum((0, 1)) / k.sum()
            assert np.all(np.abs(got[y, x] - ref) <= 0.5 + 1e-4)


def test_scene_with_png_embeddings_is_read(tmp_path):
    """What makescene leaves in a view directory: undistorted.png.  The scene
    reader reports its size from the IHDR chunk; a view that holds both
    containers prefers the .mvei; a file called .jpg that is no JPEG does not
    count as an image."""
    from smvs_amd import host
    inputs = synth.pipeline_inputs("plane", 96, 64, 2, n_features=20)
    d = str(tmp_path)
    mve_scene.write_scene(d, inputs, container="png")
    info = host.scene_info(d)
    assert info["present"].tolist() == [1, 1, 1]
    assert info["width"].tolist() == [96, 96, 96] and info["height"].tolist() == [64, 64, 64]
    v0 = os.path.join(d, "views", "view_0000.mve")
    got = host.load_byte_image(os.path.join(v0, "undistorted.png"))
    assert np.array_equal(got, np.asarray(inputs["images"][0], np.uint8))
    os.rename(os.path.join(d, "views", "view_0001.mve", "undistorted.png"),
              os.path.join(d, "views", "view_0001.mve", "undistorted.jpg"))
    info = host.scene_info(d)
    assert info["width"].tolist() == [96, 0, 96]


def _jpeg_test_image(h, w, c, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    a = np.stack([127 + 100 * np.sin(xx / (7.0 + k)) * np.cos(yy / (5.0 + 2 * k))
                  for k in range(c)], -1) + rng.normal(0, 12, (h, w, c))
    a = np.clip(a, 0, 255).astype(np.uint8)
    return a if c > 1 else a[:, :, 0]


def test_jpeg_decoder_matches_libjpeg(tmp_path):
    """csrc/host/jpeg_io.cc against Pillow's decoder (libjpeg-turbo with
    libjpeg's default parameters: islow IDCT, fancy upsampling -- what
    mve::image::load_jpg_file uses [MVE-unverified M32]) -- BIT-identical on
    4:4:4 / 4:2:2 / 4:2:0 / 4:1:1 and grey images, sequential AND progressive
    (spectral selection + successive approximation), qualities 30 .. 95,
    standard and optimised Huffman tables, restart intervals, RGB stored without
    a colour transform, sizes from 1 x 1 to several MCUs that are no multiple of
    8 or 16."""
    PIL = pytest.importorskip("PIL.Image")
    from smvs_amd import host
    p = str(tmp_path / "t.jpg")
    checked = 0

    def check(a, **kw):
        nonlocal checked
        PIL.fromarray(a).save(p, format="JPEG", **kw)
        want = np.array(PIL.open(p))
        got = host.load_byte_image(p)
        assert got.shape == want.shape, (a.shape, kw)
        assert np.array_equal(got, want), (a.shape, kw, int(np.abs(got.astype(int) - want).max()))
        checked += 1

    for i, (h, w) in enumerate(((33, 47), (64, 64), (17, 9), (120, 200), (8, 8), (1, 1), (2, 35))):
        for sub in (0, 1, 2):
            for q, opt in ((30, False), (75, True), (95, False)):
                for progressive in (False, True):
                    check(_jpeg_test_image(h, w, 3, 10 * i + sub), quality=q, subsampling=sub,
                          optimize=opt, progressive=progressive)
        check(_jpeg_test_image(h, w, 1, i), quality=80)
        check(_jpeg_test_image(h, w, 1, i), quality=80, progressive=True)
    a = _jpeg_test_image(100, 130, 3, 99)
    check(a, quality=85, subsampling="4:1:1")
    check(a, quality=80, restart_marker_blocks=3)
    check(a, quality=80, restart_marker_rows=1, subsampling=2)
    check(a, quality=80, restart_marker_blocks=1, subsampling=1)
    check(a, quality=80, restart_marker_rows=1, subsampling=2, progressive=True)
    check(a, quality=80, restart_marker_blocks=2, progressive=True)
    check(a, quality=90, keep_rgb=True)
    assert checked >= 140


def test_jpeg_decoder_refuses_what_it_does_not_decode(tmp_path):
    PIL = pytest.importorskip("PIL.Image")
    from smvs_amd import host
    a = _jpeg_test_image(40, 52, 3, 5)
    p = str(tmp_path / "t.jpg")
    cmyk = PIL.fromarray(np.dstack([a, a[:, :, 0]])).convert("CMYK")
    cmyk.save(p, format="JPEG", quality=90)
    with pytest.raises(Exception, match="CMYK"):
        host.load_byte_image(p)
    PIL.fromarray(a).save(p, format="JPEG", quality=90)
    data = open(p, "rb").read()
    open(p, "wb").write(data[: len(data) // 2])                  # truncated
    with pytest.raises(Exception, match="truncated"):
        host.load_byte_image(p)
    # a forged frame header (65535 x 65535 x 3) is refused before anything is allocated
    sof = data.index(b"\xff\xc0")
    forged = bytearray(data)
    forged[sof + 5: sof + 9] = b"\xff\xff\xff\xff"
    open(p, "wb").write(bytes(forged))
    with pytest.raises(Exception, match="out of range"):
        host.load_byte_image(p)
    open(p, "wb").write(b"not a jpeg at all")
    with pytest.raises(Exception):
        host.load_byte_image(p)


def test_scene_with_jpeg_embeddings_is_read(tmp_path):
    """smvsrecon --image=original (app/smvsrecon.cc:41, 156): the embedding
    makescene keeps as original.jpg.  The scene reader reports its size from the
    frame header and loads the pixels libjpeg would."""
    PIL = pytest.importorskip("PIL.Image")
    from smvs_amd import host
    inputs = synth.pipeline_inputs("plane", 96, 64, 2, n_features=20)
    d = str(tmp_path)
    mve_scene.write_scene(d, inputs, embedding="original", container="jpg")
    info = host.scene_info(d, "original")
    assert info["present"].tolist() == [1, 1, 1]
    assert info["width"].tolist() == [96, 96, 96] and info["height"].tolist() == [64, 64, 64]
    v0 = os.path.join(d, "views", "view_0000.mve", "original.jpg")
    got = host.load_byte_image(v0)
    assert np.array_equal(got.reshape(np.array(PIL.open(v0)).shape), np.array(PIL.open(v0)))
