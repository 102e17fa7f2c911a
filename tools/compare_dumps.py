#!/usr/bin/env python3
"""Where do the C++ host + HIP optimiser and the oracle part ways?  Runs both
on the same inputs with their state dumps switched on (SMVS_DUMP_DIR /
ORC_DUMP_DIR), finds the first stage whose validity differs and re-runs the
oracle's cut_boundaries on the DEVICE side's pre-cut state: if that reproduces
the device's post-cut state the two topology implementations agree and the
difference comes from the (rounding-level) difference of the Newton results.
    compare_dumps.py W H N [--sgm] [--shading]"""
import glob, os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from smvs_amd import synth, host
import smvs_amd
from oracle import pyoracle as oracle


def load(path):
    raw = open(path, "rb").read()
    scale, npx, npy = np.frombuffer(raw[:12], np.int32)
    nn, npatch = (npx + 1) * (npy + 1), npx * npy
    o = 12
    nodes = np.frombuffer(raw[o:o + 32 * nn], np.float64).reshape(nn, 4); o += 32 * nn
    nv = np.frombuffer(raw[o:o + nn], np.uint8); o += nn
    pv = np.frombuffer(raw[o:o + npatch], np.uint8); o += npatch
    vis = np.frombuffer(raw[o:o + 4 * npatch], np.uint32)
    return dict(scale=int(scale), npx=int(npx), npy=int(npy), nodes=nodes, node_valid=nv,
                patch_valid=pv, patch_vis=vis)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    w, h, n = int(args[0]), int(args[1]), int(args[2])
    use_sgm, shading = "--sgm" in sys.argv, "--shading" in sys.argv
    oracle.lib().orc_set_threads(max(1, min(os.cpu_count() or 1, 64)))
    inp = synth.pipeline_inputs("sphere", w, h, n, flen=1.2)
    d_dev, d_orc = tempfile.mkdtemp(), tempfile.mkdtemp()
    os.environ["SMVS_DUMP_DIR"] = d_dev
    os.environ["ORC_DUMP_DIR"] = d_orc
    sgm = sgm_o = None
    if use_sgm:
        sgm = host.sgm_depth(inp, sgm_scale=1)
        sgm_o = oracle.sgm_depth_for_view(inp, sgm_scale=1, roundtrip=True)
    host.optimize(inp, min_scale=2, use_shading=shading, sgm_depth=sgm)
    oracle.optimize(inp, regularization=0.01, num_iterations=5, min_scale=2,
                    use_shading=shading, sgm_depth=sgm_o)
    names = sorted(os.path.basename(p) for p in glob.glob(d_dev + "/*.bin"))
    order = lambda nm: (-int(nm[1]), int(nm.split("_i")[1][0]), 0 if "newton" in nm else 1)
    names.sort(key=order)
    for nm in names:
        a, b = load(os.path.join(d_dev, nm)), load(os.path.join(d_orc, nm))
        same_pv = np.array_equal(a["patch_valid"], b["patch_valid"])
        both = (a["node_valid"] != 0) & (b["node_valid"] != 0)
        dn = np.abs(a["nodes"][both] - b["nodes"][both])
        rel = dn[:, 0].max() / np.abs(b["nodes"][both][:, 0]).max() if both.any() else 0
        print("%-22s patches %6d / %6d  validity equal %s  max |f| diff rel %.2e"
              % (nm, a["patch_valid"].sum(), b["patch_valid"].sum(), same_pv, rel))
        if same_pv:
            continue
        diff = np.nonzero(a["patch_valid"] != b["patch_valid"])[0]
        print("  first differing stage; patches", diff[:10], "device valid",
              a["patch_valid"][diff[:10]], "oracle valid", b["patch_valid"][diff[:10]])
        if "cut" not in nm:
            break
        pre = nm.replace("cut", "newton")
        A0, B0 = load(os.path.join(d_dev, pre)), load(os.path.join(d_orc, pre))
        scale = A0["scale"]
        ctx = smvs_amd.ViewContext(w, h, n)
        for v, img in enumerate(inp["images"]):
            ctx.upload_image(v - 1, img)
        ctx.set_scale(scale)
        grads = [ctx.download_planes(v - 1)[0] for v in range(n + 1)]
        images = [img.astype(np.float32) / np.float32(255.0) for img in inp["images"]]
        cams = inp["cams"]
        Ms, ts = zip(*[synth.reprojection(cams[0], c) for c in cams[1:]])
        g = synth.grid_for_scale(w, h, scale)
        for label, S in (("device", A0), ("oracle", B0)):
            surf = dict(g); surf.update(width=w, height=h, scale=scale, nodes=S["nodes"].copy(),
                                        node_valid=S["node_valid"].copy(),
                                        patch_valid=S["patch_valid"].copy(),
                                        patch_vis=S["patch_vis"].copy())
            tp = oracle.TopologyProblem(surf, images, grads, np.array(Ms), np.array(ts),
                                        cams[0].flen)
            mse = tp.patch_mse()
            tp.cut_boundaries()
            tgt = a if label == "device" else b
            print("  oracle cut_boundaries on the %s pre-cut state reproduces the %s post-cut "
                  "state: %s" % (label, label, np.array_equal(tp.patch_valid, tgt["patch_valid"])))
            for p in diff[:4]:
                print("    patch %d: mse (pre-cut %s state) %.17g, vis %#x" % (p, label, mse[p],
                      S["patch_vis"][p]))
        ctx.close()
        break


if __name__ == "__main__":
    main()
