"""Builds libsmvs_hip.so (HIP kernels + C ABI) in-tree with hipcc for gfx950."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libsmvs_hip.so")
SOURCES = ["ctx.hip", "gn_construct.hip", "cg.hip", "cg_resident.hip", "update.hip", "sgm.hip",
           "scale.hip", "topology.hip", "mesh.hip", "pool.hip", "surface.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
         "-Wall", "-Wno-unused-function"]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + ["common.h", os.path.join("host", "topo_math.h"),
                                                      os.path.join("host", "surface_math.h")]]
    deps.append(os.path.join(HERE, "..", "include", "smvs_hip.h"))
    return any(os.path.getmtime(d) > t for d in deps)


HOST_DIR = os.path.join(CSRC, "host")
HOST_LIB = os.path.join(HOST_DIR, "libsmvs_host.so")
HOST_SOURCES = ["camera.cc", "stereo_view.cc", "surface.cc", "sgm_stereo.cc",
                "depth_optimizer.cc", "view_selection.cc", "view_queue.cc", "conjugate_gradient.cc",
                "gauss_newton_step.cc", "scene_io.cc", "png_io.cc", "jpeg_io.cc", "pinned_images.cc", "host_capi.cc"]


def _host_stale():
    if not os.path.exists(HOST_LIB):
        return True
    t = os.path.getmtime(HOST_LIB)
    deps = [os.path.join(HOST_DIR, f) for f in os.listdir(HOST_DIR)
            if f.endswith((".cc", ".h"))]
    deps += [LIB, os.path.join(HERE, "..", "include", "smvs_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build_host(force=False, verbose=False):
    """C++ host mirror (DepthOptimizer / Surface / StereoView / SGMStereo)
    on top of the C ABI."""
    if not force and not _host_stale():
        return HOST_LIB
    cxx = os.environ.get("CXX", "g++")
    # -ffp-contract=off: the float routines that must round like the reference
    # (fill_reprojection, the grid surgery, topo_math.h) may not be fused into
    # FMAs whatever -march the caller's CXX / CXXFLAGS select
    cmd = [cxx, "-std=c++17", "-O3", "-ffp-contract=off", "-fPIC", "-Wall", "-pthread",
           "-shared", "-o", HOST_LIB] \
        + [os.path.join(HOST_DIR, s) for s in HOST_SOURCES] \
        + ["-L" + CSRC, "-lsmvs_hip", "-lz", "-Wl,-rpath,$ORIGIN/..",
           "-Wl,-rpath,/opt/rocm/lib"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return HOST_LIB


RCCL_LIB = os.path.join(CSRC, "libsmvs_rccl.so")


def build_rccl(force=False, verbose=False):
    """libsmvs_rccl.so: the lighting all-reduce over RCCL (include/smvs_rccl.h);
    links libsmvs_hip.so and librccl."""
    src = os.path.join(CSRC, "rccl_light.hip")
    deps = [src, LIB, os.path.join(CSRC, "common.h"),
            os.path.join(HERE, "..", "include", "smvs_rccl.h")]
    if not force and os.path.exists(RCCL_LIB) \
            and all(os.path.getmtime(d) <= os.path.getmtime(RCCL_LIB) for d in deps):
        return RCCL_LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + ["-shared", "-o", RCCL_LIB, src, "-L" + CSRC, "-lsmvs_hip",
                             "-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,$ORIGIN",
                             "-Wl,-rpath,/opt/rocm/lib"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return RCCL_LIB


def build(force=False, verbose=False):
    lib = _build_hip(force, verbose)
    build_host(force, verbose)
    build_rccl(force, verbose)
    return lib


def _build_hip(force=False, verbose=False):
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    for s in SOURCES:
        obj = os.path.join(CSRC, s.replace(".hip", ".o"))
        objs.append(obj)
        cmd = [hipcc] + FLAGS + ["-c", os.path.join(CSRC, s), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs \
        + ["-Wl,-rpath,/opt/rocm/lib"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
