#!/usr/bin/env python3
"""Randomised parity sweep (not part of the test suite): random image sizes,
neighbour counts, scales, shading on / off, partially active sets; one Newton
step and one whole Newton loop on the GPU against the oracle.
Usage: fuzz_parity.py [cases [seed [big]]] -- big = 1: grids of 5 k .. 60 k
nodes (scales 1-3, images 300 .. 900 px wide), i.e. 10 .. 120 tiles of the
resident solver with its grid-wide exchange; default: single-tile grids."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import smvs_amd
from smvs_amd import synth
from oracle import pyoracle as oracle

def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300)

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
big = len(sys.argv) > 3 and sys.argv[3] == "1"
oracle.lib().orc_set_threads(max(1, (os.cpu_count() or 2) // 2))
worst = dict(H=0, g=0, P=0, x=0)
for case in range(n_cases):
    scale = int(rng.integers(1, 4 if big else 6))
    ps = 1 << scale
    if big:
        w = int(rng.integers(300, 900)); h = int(rng.integers(200, 700))
    else:
        w = int(rng.integers(6 * ps + 8, 14 * ps + 40)); h = int(rng.integers(5 * ps + 8, 10 * ps + 40))
    n_subs = int(rng.integers(1, 9))
    shading = bool(rng.integers(0, 2))
    light_reg = float(rng.choice([0.0, 0.5])) if shading else 0.0
    prob = synth.make_problem(w, h, n_subs, scale, shading=shading, noise=float(rng.choice([0.01, 0.03])), seed=int(rng.integers(1, 10000)))
    surf = prob["surf"]
    if surf["patch_valid"].sum() < 4:
        print("case %d skipped (no valid patches)" % case); continue
    lighting = prob["lighting"] if shading else None
    ctx = smvs_amd.ViewContext(w, h, n_subs)
    ctx.set_views(prob["views"]); ctx.set_surface(surf)
    orc = oracle.OracleProblem(surf, prob["views"])
    active = surf["node_valid"].copy()
    active[rng.random(active.size) < rng.choice([0.0, 0.3, 0.7])] = 0
    ctx.set_active(active)
    n_gpu = ctx.gn_construct(0.01, light_reg, lighting)
    H9, g, P = ctx.gn_download()
    ref = orc.gn_construct(active, 0.01, light_reg, lighting)
    assert n_gpu == ref["active_patches"], (case, n_gpu, ref["active_patches"])
    eH, eg, eP = rel(H9, ref["H9"]), rel(g, ref["g"]), rel(P, ref["P"])
    it, info = ctx.cg_solve()
    x = ctx.cg_x()
    xr, itr, infor = orc.cg_solve(H9, ref["present"], P, -g, 200, 0.01 * np.linalg.norm(g), 1e-3)
    ex = rel(x, xr) if np.linalg.norm(xr) > 0 else 0.0
    ok_it = (it, info) == (itr, infor)
    n_act, _, nan = ctx.update_and_reactivate(0.15, False)
    new_active, n_ref, _ = orc.update_and_reactivate(xr, active)
    a_gpu, _ = ctx.get_active()
    same_active = np.array_equal(a_gpu, new_active)
    print("case %2d: %4dx%-4d scale %d subs %d shading %d act %5d | H %.1e g %.1e P %.1e x %.1e it %s active %s"
          % (case, w, h, scale, n_subs, shading, int(active.sum()), eH, eg, eP, ex, ok_it, same_active))
    # Hard: the system, the iteration count and the new active set.  Soft: x --
    # both sides solve the SAME system (the oracle gets the device's H, g, P),
    # so ex is what the different association of the sums does to CG; on the
    # ill-conditioned systems of tiny scale-1 surfaces the iterates drift apart
    # within the solver's 1e-3 tolerance.  Counted, not asserted.
    # (`it`: same iteration count and return code.  A flip -- the termination
    # test of a long solve decided the other way, DESIGN.md section 5 -- moves
    # x by up to the solver's tolerance and may change the active set; counted)
    assert eH < 1e-8 and eg < 1e-8 and eP < 1e-5, case
    assert ok_it or abs(it - itr) <= max(2, 0.05 * itr), (case, it, itr)
    assert same_active or not ok_it, case
    assert ex < (1e-2 if ok_it else 1e-1), (case, ex)
    worst["single_solve_iteration_flips"] = worst.get("single_solve_iteration_flips", 0) + (not ok_it)
    worst["x_above_1e-8"] = worst.get("x_above_1e-8", 0) + (ex > 1e-8)
    for k, v in (("H", eH), ("g", eg), ("P", eP), ("x", ex)):
        worst[k] = max(worst[k], v)
    # ---- the whole Newton loop (fused assembly + resident solver, launch-ahead)
    # against the oracle's loop run step by step: same control flow, same CG
    # iteration total, depth to 1e-6
    ctx.set_surface(surf)
    orc = oracle.OracleProblem(surf, prob["views"])
    max_steps = int(rng.integers(1, 7))
    stats = ctx.run_loop(0.01, light_reg, lighting, max_newton_steps=max_steps)
    act = surf["node_valid"].copy()
    n_init = int(act.sum()); n_act = n_init; steps = 0; its = 0; psteps = 0
    while steps < max_steps and n_act > n_init // 20:
        steps += 1
        ref = orc.gn_construct(act, 0.01, light_reg, lighting)
        psteps += ref["active_patches"]
        xr, itr, _ = orc.cg_solve(ref["H9"], ref["present"], ref["P"], -ref["g"], 200,
                                  0.01 * np.linalg.norm(ref["g"]), 1e-3)
        its += itr
        act, n_act, _ = orc.update_and_reactivate(xr, act)
    ed = rel(ctx.depth_map(), orc.depth_map())
    same = (stats["newton_steps"], stats["active_patch_steps"],
            stats["final_active_nodes"]) == (steps, psteps, n_act)
    # The CG iteration totals are expected to be identical as well; where a
    # long solve of an ill-conditioned (tiny, scale-1) system ends an iteration
    # or two apart -- the termination test is a discrete decision on sums
    # whose association differs from the reference's sequential chain
    #  -- x moves at the level of the solver's tolerance
    # and the case is reported, with the north-star bound on the depth.
    its_gpu = stats["linear_iterations"]
    print("         loop: %d steps (max %d), CG iterations %d%s, depth %.1e, control flow %s"
          % (steps, max_steps, its, "" if its_gpu == its else " (DEVICE %d)" % its_gpu,
             ed, same))
    assert (same and ed < 1e-4) or its_gpu != its, (case, stats, steps, psteps, n_act, its)
    assert ed < 1e-3, (case, ed)
    worst["loop_control_flow_differs"] = worst.get("loop_control_flow_differs", 0) + (not same)
    worst["depth_above_1e-6"] = worst.get("depth_above_1e-6", 0) + (ed > 1e-6)
    worst["iteration_mismatches"] = worst.get("iteration_mismatches", 0) + (its_gpu != its)
    worst["loop_depth"] = max(worst.get("loop_depth", 0), ed)
    ctx.close()
print("cases", n_cases)
print("worst", worst)
