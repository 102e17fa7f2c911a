#!/usr/bin/env python3
"""How the active set of the bench workload evolves inside one Newton batch:
per step the active nodes / patches, the PCG iterations, how the active nodes
cluster (occupied 16x8 node tiles, the SpMV's work unit) and the HIP-event
time of every kernel class.  Feeds the "cost proportional to the active set"
work."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench
import smvs_amd

TX, TY = 16, 8


def tile_stats(active, stride, rows):
    a = active.reshape(rows, stride)
    ty = (rows + TY - 1) // TY
    tx = (stride + TX - 1) // TX
    pad = np.zeros((ty * TY, tx * TX), np.uint8)
    pad[:rows, :stride] = a
    t = pad.reshape(ty, TY, tx, TX).sum(axis=(1, 3))
    occ = int((t > 0).sum())
    return occ, tx * ty, float(t[t > 0].mean()) / (TX * TY) if occ else 0.0


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else None
    prob = bench.make_problem(0, "--small" in sys.argv)
    surf = prob["surf"]
    ctx = smvs_amd.ViewContext(surf["width"], surf["height"], bench.NSUBS)
    ctx.set_views(prob["views"])
    ctx.set_surface(surf)
    ctx.save_nodes()
    bench.run_steps(ctx, prob, 10)
    stride = surf["npx"] + 1
    rows = surf["npy"] + 1
    steps = []
    ctx.set_nodes(surf["nodes"])
    ctx.set_active(None)
    ctx.synchronize()
    n_init = int(surf["node_valid"].sum())
    n_act = n_init
    ctx.profile(True)
    k = 0
    while n_act > n_init // 20 and k < 60:
        act, n_before = ctx.get_active()
        occ, tiles, fill = tile_stats(act, stride, rows)
        ctx.profile_reset()
        t0 = time.perf_counter()
        st = ctx.run_loop(bench.REG, max_newton_steps=1, reset_active=False)
        dt = time.perf_counter() - t0
        prof = ctx.profile_get()
        k += 1
        n_act = st["final_active_nodes"]
        steps.append(dict(step=k, active_nodes=n_before, active_patches=st["active_patch_steps"],
                          cg_iterations=st["linear_iterations"], wall_ms=1e3 * dt,
                          tiles_occupied=occ, tiles_total=tiles, tile_fill=fill,
                          kernels={kk: [round(v[0], 4), int(v[1])] for kk, v in prof.items()}))
        print("step %2d: nodes %6d patches %6d cg %3d  tiles %4d/%4d fill %.2f  | HIP-event us: patch %.0f  "
              "resident solve (assembly inside) %.0f  re-activation %.0f  loop begin + end of step %.0f" % (
                  k, n_before, st["active_patch_steps"], st["linear_iterations"], occ, tiles, fill,
                  1e3 * prof["patch"][0], 1e3 * prof["cg_resident"][0],
                  1e3 * prof["reactivate"][0], 1e3 * prof["misc"][0]))
    res = dict(valid_nodes=n_init, valid_patches=int(surf["patch_valid"].sum()), steps=steps)
    if out_path:
        with open(out_path, "w") as f:
            json.dump(res, f)
    ctx.close()


if __name__ == "__main__":
    main()
