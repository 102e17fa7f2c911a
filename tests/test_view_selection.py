"""SURVEY.md 8(f)-4, the part that needs no MVE checkout: the neighbour-view
selection smvs::ViewSelection (lib/view_selection.cc:14-161).  The C++ host
mirror (csrc/host/view_selection.cc, sort based) against the oracle's literal
restatement (oracle/smvs_oracle_scene.c, the reference's std::map /
std::multimap emulated by sorted insertion) on seeded random scenes: the
neighbour lists are index lists and have to be identical.  Host-only code: no
GPU involved."""
import numpy as np
import pytest


def _look_at(pos, target, roll=0.0):
    """world-to-camera rotation (rows = camera axes) and translation"""
    z = target - pos
    z = z / np.linalg.norm(z)
    up = np.array([0.0, 0.0, 1.0])
    x = np.cross(z, up)
    x = x / np.linalg.norm(x)
    y = np.cross(z, x)
    c, s = np.cos(roll), np.sin(roll)
    x, y = c * x + s * y, -s * x + c * y
    R = np.stack([x, y, z]).astype(np.float32)
    t = (-R @ pos).astype(np.float32)
    return R, t


def _scene(seed, n_views=24, n_features=1500, ring=True):
    rng = np.random.default_rng(seed)
    views = []
    for i in range(n_views):
        if ring:
            a = 2 * np.pi * i / n_views + 0.05 * rng.standard_normal()
            pos = np.array([4 * np.cos(a), 4 * np.sin(a), 0.4 * rng.standard_normal()])
        else:
            pos = 4 * rng.standard_normal(3)
        target = 0.3 * rng.standard_normal(3)
        roll = np.pi if rng.random() < 0.1 else 0.05 * rng.standard_normal()
        R, t = _look_at(pos, target, roll)
        views.append(dict(present=True, id=i, flen=float(0.8 + 0.6 * rng.random()),
                          rot=R, trans=t, has_image=True,
                          width=int(rng.choice([640, 800, 1024])),
                          height=int(rng.choice([480, 600, 768]))))
    feats = (0.8 * rng.standard_normal((n_features, 3))).astype(np.float32)
    refs = []
    for f in range(n_features):
        # a feature is seen by the views within a random angular window
        centre = rng.integers(0, n_views)
        width = rng.integers(2, n_views // 2)
        seen = sorted({int((centre + k) % n_views) for k in range(-width, width + 1)
                       if rng.random() < 0.8})
        refs.append(seen)
    return dict(views=views, features=feats, refs=refs)


@pytest.fixture(scope="module")
def host():
    from smvs_amd import host as h
    h.load()
    return h


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_bundle_based_selection_matches_oracle(host, oracle, seed):
    scene = _scene(seed)
    picked = 0
    for view in range(len(scene["views"])):
        for k in (1, 6, 10):
            got = host.select_neighbors(scene, view, k)
            want = oracle.select_neighbors(scene, view, k)
            assert got == want, (view, k)
            assert len(got) <= k and view not in got
            picked += len(got)
    assert picked > 0   # (the scenes do share features)


@pytest.mark.parametrize("seed,ring", [(11, True), (12, False), (13, False)])
def test_position_based_selection_matches_oracle(host, oracle, seed, ring):
    scene = _scene(seed, n_views=30, ring=ring)
    total = 0
    for view in range(len(scene["views"])):
        got = host.select_neighbors(scene, view, 6, use_bundle=False)
        want = oracle.select_neighbors(scene, view, 6, use_bundle=False)
        assert got == want, view
        total += len(got)
        # nearest first
        centre = lambda v: -np.asarray(v["rot"], np.float32).T @ np.asarray(v["trans"], np.float32)
        d = [np.linalg.norm(centre(scene["views"][view]) - centre(scene["views"][i])) for i in got]
        assert d == sorted(d)
    assert total > 0


def test_selection_edge_cases(host, oracle):
    """Views without a camera, without the image embedding, holes in the view
    list, two cameras at exactly the same distance (the std::map keeps the
    later one), fewer than 11 shared features, a main view without image."""
    scene = _scene(5, n_views=16, n_features=600)
    v = scene["views"]
    v[3]["flen"] = 0.0                 # no camera: never a candidate
    v[5]["has_image"] = False          # no embedding: skipped by the bundle path
    v[7]["present"] = False            # null View::Ptr
    # views 8 and 9 share one camera: exactly the same distance from everybody,
    # and std::map<float, size_t>::operator[] keeps the later one
    v[9]["rot"] = v[8]["rot"]
    v[9]["trans"] = v[8]["trans"]
    # view 12 shares only a handful of features with anybody
    scene["refs"] = [[i for i in r if i != 12] for r in scene["refs"]]
    for f in range(8):
        scene["refs"][f] = sorted(set(scene["refs"][f]) | {12, 0})
    for view in range(16):
        if view == 7:
            continue
        for use_bundle in (True, False):
            got = host.select_neighbors(scene, view, 6, use_bundle=use_bundle)
            want = oracle.select_neighbors(scene, view, 6, use_bundle=use_bundle)
            assert got == want, (view, use_bundle)
            assert 3 not in got and 7 not in got
            if view not in (8, 9):
                assert 8 not in got
            if use_bundle:
                assert 5 not in got and 12 not in got
    assert host.select_neighbors(scene, 5, 6) == []       # main view without image
    assert oracle.select_neighbors(scene, 5, 6) == []
    assert host.select_neighbors(scene, 12, 6) == []      # <= 10 shared features
