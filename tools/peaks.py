#!/usr/bin/env python3
"""Measured peaks of the GPU this runs on (the HBM and FP64
figures the rooflines divide by are measured, not assumed).

    python tools/peaks.py [--build-only] [--out FILE]

Builds tools/libpeaks.so from tools/peaks.hip (hipcc, gfx950) and prints one
JSON object: copy / read / write GB/s over 1 GiB, FP64 TFLOP/s of the vector
FMA, of v_mfma_f64_16x16x4_f64 and of v_mfma_f64_4x4x4_4b_f64, issue and
dependent-latency cycles, whether a VALU wave and an MFMA wave on one SIMD
overlap, and the lane layouts of the two FP64 MFMA shapes.
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "peaks.hip")
LIB = os.path.join(HERE, "libpeaks.so")


def build(force=False):
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
                           "-shared", "-o", LIB, SRC, "-Wl,-rpath,/opt/rocm/lib"])
    return LIB


def _decode_probe4(p):
    """p[la][lb] = mask of result lanes that are non-zero when A is one-hot in
    lane la and B one-hot in lane lb.  Returns per-lane (block, i, k) of A,
    (block, k, j) of B and (block, i, j) of D, inferred from the products."""
    pairs = {}
    for la in range(64):
        for lb in range(64):
            m = p[la * 64 + lb]
            if m:
                pairs[(la, lb)] = [l for l in range(64) if (m >> l) & 1]
    return {"%d,%d" % k: v for k, v in sorted(pairs.items())}


def run(device=0):
    lib = C.CDLL(build())
    out = (C.c_double * 32)()
    p4 = (C.c_ulonglong * 4096)()
    p16 = (C.c_ulonglong * (4 * 4096))()
    rc = lib.peaks_run(device, out, 32, p4, p16)
    if rc != 0:
        raise RuntimeError("peaks_run failed: %d" % rc)
    o = list(out)
    res = {
        "cus": int(o[0]), "clock_ghz": o[1],
        "hbm_copy_GBps": o[2], "hbm_read_GBps": o[3], "hbm_write_GBps": o[4],
        "fp64_valu_fma_TFLOPs": {"1_wave_per_simd": o[5], "2": o[6], "4": o[7]},
        "fp64_valu_fma_dependent_cycles": o[8],
        "v_rcp_f64_cycles_per_instr": o[9],
        "fp64_mfma_16x16x4_TFLOPs": {"1_wave_per_simd": o[10], "2": o[11]},
        "fp64_mfma_16x16x4_issue_cycles": o[12],
        "fp64_mfma_16x16x4_dependent_cycles": o[13],
        "fp64_mfma_4x4x4_TFLOPs": {"1_wave_per_simd": o[14], "2": o[15]},
        "fp64_mfma_4x4x4_issue_cycles": o[16],
        "fp64_mfma_4x4x4_dependent_cycles": o[17],
        "valu_plus_mfma_same_simd_ms": {"both": o[18], "valu_alone": o[19],
                                        "mfma_alone": o[20]},
    }
    # cycles per VALU FMA instruction per SIMD at 4 waves / SIMD
    if o[7] > 0:
        res["fp64_valu_fma_cycles_per_instr"] = 128.0 * o[0] * 4 * o[1] * 1e9 / (o[7] * 1e12)
    res["layout_4x4x4"] = _decode_probe4(list(p4))
    lay16 = {}
    for la in range(64):
        for lb in range(64):
            for r in range(4):
                m = p16[(la * 64 + lb) * 4 + r]
                if m:
                    lay16["%d,%d,r%d" % (la, lb, r)] = [l for l in range(64) if (m >> l) & 1]
    res["layout_16x16x4_samples"] = {k: lay16[k] for k in list(sorted(lay16))[:64]}
    res["layout_16x16x4_count"] = len(lay16)
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--build-only", action="store_true")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    build(force=a.build_only)
    if a.build_only:
        sys.exit(0)
    r = run()
    s = json.dumps(r)
    if a.out:
        with open(a.out, "w") as f:
            f.write(s)
    short = {k: v for k, v in r.items() if not k.startswith("layout")}
    print(json.dumps(short, indent=1))
