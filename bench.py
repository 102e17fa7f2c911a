#!/usr/bin/env python3
"""Headline benchmark: Gauss-Newton iterations/s x active patches on a
1920x1080 reference view with 8 neighbours (BASELINE.json configs[1]:
synthetic textured sphere, -o2, basic photometric optimizer).

A "step" is one pass of the Newton loop of lib/depth_optimizer.cc:219-304
(construct + PCG solve + node update + re-activation) at scale 2 over the
evolving active set; when the loop ends (active <= initial/20, :220) the
surface is reset to the same perturbed start and the loop restarts.
value = sum over timed steps of active patches / wall time, inputs resident
in HBM.  One process per GPU; ranks work on independent reference views
(weak scaling, no data-path collective).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H, NSUBS, SCALE = 1920, 1080, 8, 2
REG = 0.01            # 0.01 * alpha (app/smvsrecon.cc:712), alpha = 1
NOISE = 0.002


def make_problem(rank, small=False):
    from smvs_amd import synth
    if small:
        return synth.make_problem(480, 270, NSUBS, SCALE, noise=NOISE, seed=2000 + rank)
    return synth.make_problem(W, H, NSUBS, SCALE, noise=NOISE, seed=2000 + rank)


def run_steps(ctx, prob, steps):
    """Run exactly `steps` Newton steps; returns (active patch-steps, CG its)."""
    done = 0
    patch_steps = 0
    cg_its = 0
    need_reset = False
    while done < steps:
        if need_reset:
            ctx.set_nodes(prob["surf"]["nodes"])
        st = ctx.run_loop(REG, max_newton_steps=min(200, steps - done),
                          reset_active=True)
        done += st["newton_steps"]
        patch_steps += st["active_patch_steps"]
        cg_its += st["linear_iterations"]
        need_reset = True
        if st["newton_steps"] == 0:
            raise RuntimeError("Newton loop made no progress")
    return patch_steps, cg_its


def cpu_baseline(prob):
    """Oracle (CPU restatement, 1 thread): one full Newton step of the same
    workload with every valid node active (the first step of a batch),
    about 15 s on one core."""
    from oracle import pyoracle
    surf = prob["surf"]
    active = np.ascontiguousarray(surf["node_valid"], dtype=np.uint8)
    orc = pyoracle.OracleProblem(surf, prob["views"])
    t0 = time.perf_counter()
    ref = orc.gn_construct(active, REG)
    t1 = time.perf_counter()
    x, it, _ = orc.cg_solve(ref["H9"], ref["present"], ref["P"], -ref["g"], 200,
                            0.01 * np.linalg.norm(ref["g"]), 1e-3)
    t2 = time.perf_counter()
    orc.update_and_reactivate(x, active)
    t3 = time.perf_counter()
    return dict(value=ref["active_patches"] / (t3 - t0),
                unit="active-patch-steps/s", cores=1, kind="port",
                sample="1 Newton step, all %d patches active (first step of a batch "
                       "of the %dx%d / %d-neighbour workload): construct %.2fs, "
                       "PCG %d it %.2fs, update %.2fs"
                       % (ref["active_patches"], W, H, NSUBS, t1 - t0, it,
                          t2 - t1, t3 - t2))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--small", action="store_true", help="480x270 debug size")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--views-in-flight", type=int, default=1,
                    help="reference views processed concurrently per GPU (own context, "
                         "stream and host thread each); the headline number uses 1")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    import torch  # device plumbing + torch.distributed only
    import smvs_amd
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if smvs_amd.device_count() < 1:
        raise RuntimeError("bench.py needs a GPU")

    prob = make_problem(rank, args.small)
    surf = prob["surf"]
    w, h = surf["width"], surf["height"]
    ctx = smvs_amd.ViewContext(w, h, NSUBS, device=local_rank)
    ctx.set_views(prob["views"])
    ctx.set_surface(surf)

    def barrier():
        ctx.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    # optional: more reference views in flight on this GPU (same inputs, own
    # context / stream / host thread each); every view runs the K steps
    extra = []
    for _ in range(max(args.views_in_flight, 1) - 1):
        c = smvs_amd.ViewContext(w, h, NSUBS, device=local_rank)
        c.set_views(prob["views"]); c.set_surface(surf)
        extra.append(c)

    def run_all(steps):
        import threading
        res = [None] * (1 + len(extra))
        def work(i, c):
            res[i] = run_steps(c, prob, steps)
        th = [threading.Thread(target=work, args=(i + 1, c)) for i, c in enumerate(extra)]
        [t.start() for t in th]
        work(0, ctx)
        [t.join() for t in th]
        return sum(r[0] for r in res), sum(r[1] for r in res)

    run_all(args.warmup)
    for c in [ctx] + extra:
        c.set_nodes(surf["nodes"])
        c.synchronize()
    barrier()
    t0 = time.perf_counter()
    patch_steps, cg_its = run_all(args.steps)
    for c in extra:
        c.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    for c in extra:
        c.close()

    # whole-job aggregate: units summed over ranks, time = max over ranks
    from smvs_amd import shard
    dev = torch.device("cuda", local_rank) if dist is not None else None
    patch_steps, elapsed = shard.aggregate_throughput(patch_steps, elapsed, dist, dev)
    cg_its, _ = shard.aggregate_throughput(cg_its, 0.0, dist, dev)

    # ---- roofline of the dominant kernel: same steps, HIP-event timed ----
    roof = None
    cpu = None
    if rank == 0:
        ctx.set_nodes(surf["nodes"])
        ctx.profile(True)
        ctx.profile_reset()
        prof_patch_steps, prof_cg_its = run_steps(ctx, prob, args.steps)
        prof = ctx.profile_get()
        ctx.profile(False)
        name, (ms, cnt) = max(prof.items(), key=lambda kv: kv[1][0])
        n_nodes = ctx.num_nodes
        kernels = {k: dict(ms=round(v[0], 3), launches=int(v[1]),
                           avg_us=round(1e3 * v[0] / max(v[1], 1), 2))
                   for k, v in prof.items()}
        # Algorithmic bytes / flops per launch.  The CG
        # kernels are also launched as no-ops after convergence (the host
        # learns the iteration count late): their average duration is taken
        # over the launches that did work (= CG iterations of the profiled
        # steps), with the no-op time left in the numerator.
        per_node = {"cg_spmv": 5 * 128 + 2 * 32 + 2 * 32,      # H upper half; z, d_old; Ad, d_new
                    "cg_update": 5 * 32 + 128 + 3 * 32 + 2}    # x d r Ad b; P; x r z; mask
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "traffic_r1.json")
        if os.path.exists(tfile):
            with open(tfile) as f:
                traffic = json.load(f).get(name)
        if name in per_node:
            bytes_per_launch = per_node[name] * n_nodes
            work_launches = max(prof_cg_its, 1)
            avg_s = ms * 1e-3 / work_launches
            achieved = bytes_per_launch / avg_s / 1e9
            roof = dict(kernel=name, bound="hbm", achieved=round(achieved, 1),
                        peak=8000.0, unit="GB/s", frac=round(achieved / 8000.0, 4),
                        traffic=traffic, bytes_per_launch=bytes_per_launch,
                        avg_us=round(avg_s * 1e6, 2), launches_with_work=work_launches,
                        kernels=kernels)
        else:
            # gn_patch_kernel: FP64 arithmetic (VALU + v_mfma_f64).  0.132 MFLOP
            # per active patch is what the factored formulation executes at
            # S = 8 neighbours, P = 16 samples, counted from the SQ instruction
            # counters (profiles/r1_patch_kernel_counters.txt);
            # SURVEY 8(d)'s 0.50 MFLOP prices the reference's unfactored rows.
            # Peak: 78.6 TFLOP/s, MI355X's FP64 rate (vector and matrix alike).
            flops = 0.132e6 * prof_patch_steps / max(cnt, 1)
            avg_s = ms * 1e-3 / cnt
            achieved = flops / avg_s / 1e12
            roof = dict(kernel=name, bound="mfma", achieved=round(achieved, 3),
                        peak=78.6, unit="TFLOP/s", frac=round(achieved / 78.6, 4),
                        traffic=traffic, flops_per_launch=flops,
                        avg_us=round(avg_s * 1e6, 2), kernels=kernels)
        if not args.no_cpu_baseline and world == 1:   # N = 1 only
            cpu = cpu_baseline(prob)

    if rank == 0:
        value = patch_steps / elapsed
        out = {
            "metric": "Gauss-Newton iters/sec x active patches, 1920x1080 ref view, 8 neighbours",
            "value": value, "unit": "active-patch-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "configs[1]: %dx%d synthetic textured sphere, 1 ref + %d "
                                   "neighbours, -o2 (scale 2, %d patches), basic photometric "
                                   "optimizer, %s"
                                   % (w, h, NSUBS, int(surf["patch_valid"].sum()),
                                      "one reference view per GPU" if args.views_in_flight <= 1
                                      else "%d reference views in flight per GPU" % args.views_in_flight),
                       "regularization": REG,
                       "cg_iterations_per_step": cg_its / max(args.steps * world * max(args.views_in_flight, 1), 1),
                       "views_in_flight_per_gpu": max(args.views_in_flight, 1)},
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
