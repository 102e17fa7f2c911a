"""The reference-side binding shown in INTEGRATION.md section 2 is real code:
the block is extracted from the document, compiled against a declaration-only
mock of the reference types it touches (tests/integration_stub/) and linked
against libsmvs_hip.so -- every C-ABI call in it exists with that signature."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_integration_stub_compiles_and_links(tmp_path):
    from smvs_amd import build as hip_build
    lib = hip_build.build()
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```cpp\n(// lib/depth_optimizer_hip\.cc.*?)```", text, flags=re.S)
    assert len(blocks) == 1
    src = tmp_path / "depth_optimizer_hip.cc"
    src.write_text(blocks[0])
    out = tmp_path / "libbinding.so"
    cmd = ["g++", "-std=c++11", "-Wall", "-Werror", "-fPIC", "-shared",
           "-I", os.path.join(ROOT, "include"),
           "-I", os.path.join(ROOT, "tests", "integration_stub"),
           str(src), "-o", str(out), "-L", os.path.dirname(lib), "-lsmvs_hip",
           "-Wl,-rpath," + os.path.dirname(lib), "-Wl,--no-undefined-version"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    # every smvs_* symbol the binding references is exported by the library
    nm = subprocess.run(["nm", "-D", "--undefined-only", str(out)], capture_output=True,
                        text=True).stdout
    wanted = set(re.findall(r"\b(smvs_[a-z_0-9]+)\b", nm))
    assert {"smvs_ctx_create", "smvs_ctx_set_surface", "smvs_gn_run_loop",
            "smvs_get_nodes", "smvs_ctx_upload_sub", "smvs_surface_expand",
            "smvs_surface_remove_isolated_patches", "smvs_surface_delete_unseen_patches",
            "smvs_topology_cut_boundaries", "smvs_surface_info"} <= wanted
    exported = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True,
                              text=True).stdout
    for sym in wanted:
        assert re.search(r"\b%s\b" % sym, exported), sym


def test_reference_signature_classes_compile_and_link(tmp_path):
    """INTEGRATION.md section 1b is real code as well: GaussNewtonStep /
    ConjugateGradient with the reference's signatures, compiled against the
    host mirror's headers and linked with libsmvs_host.so."""
    from smvs_amd import build as hip_build
    lib = hip_build.build()
    host_dir = os.path.join(os.path.dirname(lib), "host")
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```cpp\n(// newton_step_with_reference_calls\.cc.*?)```", text, flags=re.S)
    assert len(blocks) == 1
    src = tmp_path / "newton_step_with_reference_calls.cc"
    src.write_text(blocks[0])
    out = tmp_path / "libstep.so"
    cmd = ["g++", "-std=c++17", "-Wall", "-Werror", "-fPIC", "-shared", "-I", host_dir,
           str(src), "-o", str(out), "-L", host_dir, "-lsmvs_host",
           "-L", os.path.dirname(lib), "-lsmvs_hip",
           "-Wl,-rpath," + host_dir, "-Wl,-rpath," + os.path.dirname(lib), "-Wl,--no-undefined"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    nm = subprocess.run(["nm", "-D", "-C", "--undefined-only", str(out)], capture_output=True,
                        text=True).stdout
    assert "smvs_amd::GaussNewtonStep::construct(" in nm
    assert "smvs_amd::ConjugateGradient::solve(" in nm


_COMPAT_TU = r"""
// a translation unit written against the reference's names (namespace smvs)
#include "smvs_compat.h"
#include <vector>

struct Identity : public smvs::ConjugateGradient::Functor   // conjugate_gradient.h:44-50
{
    std::size_t n;
    explicit Identity(std::size_t n) : n(n) {}
    smvs::ConjugateGradient::Vector multiply(smvs::ConjugateGradient::Vector const& x) const
    { return x; }
    std::size_t input_size(void) const { return n; }
    std::size_t output_size(void) const { return n; }
};

double residual_norm2(smvs::ConjugateGradient::Functor const& A,
    smvs::ConjugateGradient::Vector const& x, smvs::ConjugateGradient::Vector const& b)
{
    smvs::ConjugateGradient::Vector const Ax = A.multiply(x);
    double s = 0.0;
    for (std::size_t i = 0; i < b.size(); ++i)
        s += (Ax[i] - b[i]) * (Ax[i] - b[i]);
    return s;
}

int options_of_the_reference(void)
{
    smvs::DepthOptimizer::Options opts;        // depth_optimizer.h:30-42
    opts.regularization = 0.01;
    opts.num_iterations = 5;
    opts.min_scale = 2;
    opts.debug_lvl = 1;
    opts.use_shading = true;
    smvs::ConjugateGradient::Options cg;       // conjugate_gradient.h:29-35
    cg.max_iterations = 200;
    smvs::BlockStencilMatrix H;
    H.num_nodes = 4; H.node_stride = 2;
    H.blocks.assign(4 * 9 * 16, 0.0);
    Identity I(16);
    smvs::ConjugateGradient::Vector x(16, 1.0);
    return (int)H.multiply(x).size() + (int)residual_norm2(I, x, x) + opts.min_scale
        + cg.max_iterations;
}
"""


def test_reference_namespace_alias_and_functor_interface(tmp_path):
    """csrc/host/smvs_compat.h: `namespace smvs = smvs_amd;` -- code written
    against the reference's names compiles unchanged, Functor has the
    reference's three members (multiply included) and BlockStencilMatrix
    implements it."""
    from smvs_amd import build as hip_build
    lib = hip_build.build()
    host_dir = os.path.join(os.path.dirname(lib), "host")
    src = tmp_path / "uses_reference_names.cc"
    src.write_text(_COMPAT_TU)
    out = tmp_path / "libuses.so"
    cmd = ["g++", "-std=c++17", "-Wall", "-Werror", "-fPIC", "-shared", "-I", host_dir,
           str(src), "-o", str(out), "-L", host_dir, "-lsmvs_host",
           "-L", os.path.dirname(lib), "-lsmvs_hip",
           "-Wl,-rpath," + host_dir, "-Wl,-rpath," + os.path.dirname(lib), "-Wl,--no-undefined"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    nm = subprocess.run(["nm", "-D", "-C", "--undefined-only", str(out)], capture_output=True,
                        text=True).stdout
    assert "smvs_amd::BlockStencilMatrix::multiply(" in nm


def test_host_block_multiply_matches_oracle():
    """smvs_amd::BlockStencilMatrix::multiply on the host against the oracle's
    restatement of BlockSparseMatrix<4>::multiply (block_sparse_matrix.h:276-298)
    on a constructed system: the same products in the same order -- bit-identical."""
    import numpy as np
    from smvs_amd import synth, host
    from oracle import pyoracle
    prob = synth.make_problem(96, 64, 2, 2, noise=0.01)
    orc = pyoracle.OracleProblem(prob["surf"], prob["views"])
    act = prob["surf"]["node_valid"].copy()
    act[::5] = 0          # absent rows / columns as well
    ref = orc.gn_construct(act, 0.01)
    rng = np.random.default_rng(3)
    x = rng.standard_normal(ref["g"].size)
    want = orc.spmv(ref["H9"], ref["present"], x)
    got = host.block_multiply(ref["H9"], prob["surf"]["npx"] + 1, x)
    assert np.array_equal(got, want.reshape(-1))
    assert np.linalg.norm(got) > 0
