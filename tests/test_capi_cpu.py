"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads
and exports every symbol include/smvs_hip.h declares; argument errors are
reported as status codes (no compute without a GPU)."""
import ctypes as C

import numpy as np
import pytest


@pytest.fixture(scope="module")
def lib():
    from smvs_amd import build as hip_build, _capi
    hip_build.build()
    return _capi.load()


def test_exports_every_declared_symbol(lib):
    from smvs_amd import _capi
    syms = _capi.declared_symbols()
    assert len(syms) >= 30
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing


def test_rccl_library_exports_its_header():
    """include/smvs_rccl.h <-> csrc/libsmvs_rccl.so (the lighting all-reduce
    over RCCL): every declared entry point is exported; no call without a GPU."""
    import os, re
    from smvs_amd import build as hip_build, _capi
    path = hip_build.build_rccl()
    text = open(os.path.join(_capi.HERE, "..", "include", "smvs_rccl.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    syms = sorted(set(re.findall(r"\b(smvs_[a-z_0-9]+)\s*\(", text)))
    assert syms == ["smvs_comm_create", "smvs_comm_destroy", "smvs_comm_ranks",
                    "smvs_comm_unique_id", "smvs_light_allreduce"]
    lib = C.CDLL(path)
    assert not [s for s in syms if not hasattr(lib, s)]
    # argument errors are status codes
    assert lib.smvs_comm_unique_id(None) == -1
    assert lib.smvs_light_allreduce(None, None, 0) == -1
    assert lib.smvs_comm_destroy(None) == 0
    assert lib.smvs_comm_ranks(None, None, None) == -1


def test_product_package_does_not_import_oracle():
    import os, re
    root = os.path.join(os.path.dirname(__file__), "..", "smvs_amd")
    for dirpath, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cc", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", text, re.M), f
                assert "liboracle" not in text and "libsmvs_oracle" not in text, f


def test_argument_errors_are_status_codes(lib):
    lib.smvs_last_error.restype = C.c_char_p
    h = C.c_void_p()
    # null output pointer
    assert lib.smvs_ctx_create(0, 64, 64, 2, None) == -1
    # bad neighbour count is rejected before any HIP call
    assert lib.smvs_ctx_create(0, 64, 64, 0, C.byref(h)) == -1
    assert lib.smvs_ctx_create(0, 64, 64, 999, C.byref(h)) == -1
    assert b"n_subs" in lib.smvs_last_error()
    assert lib.smvs_ctx_destroy(None) == 0
    assert lib.smvs_gn_construct(None, C.c_double(0.1), C.c_double(0.0), None, None) == -1
    assert lib.smvs_cg_solve(None, 200, C.c_double(-1), C.c_double(1e-3), None, None) == -1
    assert lib.smvs_ctx_sgm_init_depth(None, None, 0, 0, C.c_float(5.0), 5, None) == -1
    assert lib.smvs_topology_subviews(None, None, 0, None) == -1


def test_no_gpu_fails_loudly(lib):
    if lib.smvs_device_count() > 0:
        pytest.skip("GPU present")
    h = C.c_void_p()
    rc = lib.smvs_ctx_create(0, 64, 64, 2, C.byref(h))
    assert rc < 0
    from smvs_amd import ViewContext
    from smvs_amd._capi import SmvsError
    with pytest.raises(SmvsError):
        ViewContext(64, 64, 2)


def test_synth_problem_shapes():
    from smvs_amd import synth
    prob = synth.make_problem(96, 64, 2, scale=2)
    s = prob["surf"]
    assert s["nodes"].shape == ((s["npx"] + 1) * (s["npy"] + 1), 4)
    assert prob["views"]["grad"].shape == (64, 96, 2)
    assert prob["views"]["subs"][0][1].shape == (64, 96, 3)
    assert s["patch_valid"].sum() > 0
    g = synth.grid_for_scale(1920, 1080, 2)
    assert (g["npx"], g["npy"]) == (478, 268)
