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
