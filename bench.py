#!/usr/bin/env python3
"""Headline benchmark: Gauss-Newton iterations/s x active patches on a
1920x1080 reference view with 8 neighbours (BASELINE.json configs[1]:
synthetic textured sphere, -o2, basic photometric optimizer) over BASELINE.md's
own timed region -- ALL Newton loops of ALL scales of one
DepthOptimizer::optimize (lib/depth_optimizer.cc:219-304 entered once per
batch, :53-162).

A "step" is one pass over the Newton loops of one optimize(): each of its
batches (15 at 1920x1080: scales 6 .. 2, three batches per scale) runs
construct + PCG solve + node update + re-activation until its active set is
spent (:220).  The start state of every batch -- the scale's planes of all nine
views, cameras, surface, visibility masks -- is recorded from a real optimize()
during set-up (smvs_ctx_clone_loop_state: 15 device-resident contexts) and the
timed region replays exactly those loops; the set-up asserts that the replay
reproduces the optimize()'s own batch log (Newton steps, active patch-steps, CG
iterations per batch).  What lies between the batches in optimize() (topology
tests, grid surgery, scale space) is outside the reference's loop timers and
outside this region; whole views per second are in `secondary`.
value = sum over the timed passes of active patch-steps / wall time, inputs
resident in HBM.  One process per GPU; ranks work on independent reference
views (weak scaling, no data-path collective).

    python bench.py --gpus N --steps K --warmup W [--repeats R] [--config 1|5]

With N > 1 and no torch.distributed environment the script re-executes itself
under `python -m torch.distributed.run` with N ranks on 127.0.0.1.  After W
warm-up steps the K-step region (barrier + device synchronisation on both
sides, time = max over ranks) is measured R times; `value` / `ms_per_step` are
the MEDIAN repeat, min and max are in "timing".

--workload newton_steps is the headline of rounds 1-5 (the scale-2 loop of the
same view replayed from a perturbed start surface; a step = one Newton step):
now `secondary.scale2_replay` of the default run.

--config 5 (BASELINE.json configs[4]): every rank holds --views-per-rank
reference views (64 over 8 GPUs), shading-aware (-S): per view the
GlobalLighting normal equations are accumulated on the device, optionally
summed over the lock-step round of views with an RCCL all-reduce on the device
buffers ("--shared-lighting"; the reference fits per view, which is the parity
default), solved, and the scale-2 Newton loop runs with the shading term.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import platform
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H, NSUBS, SCALE = 1920, 1080, 8, 2
REG = 0.01            # 0.01 * alpha (app/smvsrecon.cc:712), alpha = 1
NOISE = 0.002

# Assumed peaks (MI355X data sheet; MI355X_MICROARCH.md): what `frac` divides by.
HBM_PEAK_GBPS = 8000.0
FP64_PEAK_TFLOPS = 78.6
# Per-unit algorithmic work:
#  * per active patch, FP64 flops the factored construction executes at S = 8
#    neighbours, P = 16 samples (SQ instruction counters,
#    profiles/r3_patch_kernel_counters.txt: 1.081e5 executed);
#    SURVEY.md 8(d) prices the reference's unfactored rows at 0.50 MFLOP
#  * per node and CG iteration, bytes of the upper-half block stencil + vectors
FLOP_PER_PATCH = 0.1081e6
FLOP_PER_PATCH_SURVEY = 0.50e6
BYTES_PER_PATCH = 4.2e3
# The resident PCG reads its input once per solve (1,152 + 128 B per live patch,
# 64 B per node for x and b) and then exchanges per iteration: one two-level
# all-reduce whose floor is two dependent cross-CU hand-offs at the idle price
# of MI355X_MICROARCH.md's handoff-1to1 row (0.8-1.1 us each).
RESIDENT_BYTES_PER_PATCH = 144 * 8 + 128    # csrc/common.h PATCH_H_STRIDE doubles + the 16 of g
RESIDENT_BYTES_PER_NODE = 64
EXCHANGE_FLOOR_S = 2 * 1.1e-6
CG_BYTES = {"cg_spmv": 5 * 128 + 2 * 32 + 2 * 32,      # H upper half; z, d_old; Ad, d_new
            "cg_update": 5 * 32 + 128 + 3 * 32 + 2}    # x d r Ad b; P; x r z; mask


def make_problem(rank, small=False, shading=False):
    from smvs_amd import synth
    w, h = (480, 270) if small else (W, H)
    return synth.make_problem(w, h, NSUBS, SCALE, noise=NOISE, seed=2000 + rank,
                              shading=shading)


def run_steps(ctx, prob, steps, lighting=None):
    """Run exactly `steps` Newton steps; returns (active patch-steps, CG its)."""
    done = 0
    patch_steps = 0
    cg_its = 0
    need_reset = False
    while done < steps:
        if need_reset:
            ctx.restore_nodes()   # the start surface, resident in HBM
        st = ctx.run_loop(REG, lighting=lighting,
                          max_newton_steps=min(200, steps - done), reset_active=True)
        done += st["newton_steps"]
        patch_steps += st["active_patch_steps"]
        cg_its += st["linear_iterations"]
        need_reset = True
        if st["newton_steps"] == 0:
            raise RuntimeError("Newton loop made no progress")
    return patch_steps, cg_its


# ------------------------------------------ the Newton loops of one optimize()
PATCH_FLOPS_FILE = os.path.join(ROOT, "profiles", "patch_flops_r6.json")
SAMPLES_PER_PATCH = {0: 1, 1: 4, 2: 16, 3: 16, 4: 64, 5: 64, 6: 256}   # sampled pixels (gn_construct.hip)


def scene_inputs(rank, small=False):
    """The view of `secondary.optimize` / --workload optimize; ranks get
    different textures and features (independent reference views)."""
    from smvs_amd import synth
    w, h = (480, 270) if small else (W, H)
    return synth.pipeline_inputs("sphere", w, h, NSUBS, flen=1.2, seed=1234 + rank)


class LoopReplay:
    """All Newton loops of ONE DepthOptimizer::optimize, resident: the start
    state of every batch as its own device context (smvs_ctx_clone_loop_state,
    recorded while a real optimize() ran), replayed with the parameters the
    optimizer used.  run_pass() is the bench's "step"."""

    def __init__(self, inp, device=0):
        import ctypes as C
        from smvs_amd import host, _capi
        self.C = C
        self.lib = _capi.load()
        kw = dict(regularization=REG, num_iterations=5, min_scale=SCALE, device=device,
                  want_maps=False)
        host.optimize(inp, **kw)          # warm: library, pools, the resident plan
        host.record_loops(True)
        try:
            r = host.optimize(inp, **kw)
        finally:
            host.record_loops(False)
        self.log = r["log"]
        self.loops = host.take_recorded_loops()
        if len(self.loops) != len(self.log):
            raise RuntimeError("recorded %d loops for %d batches" % (len(self.loops), len(self.log)))
        geom = (C.c_int * 6)()
        for L in self.loops:
            _capi.check(self.lib.smvs_surface_info(L["handle"], geom, None))
            L["nodes"] = (geom[2] + 1) * (geom[3] + 1)
            L["patches"] = geom[2] * geom[3]
        self.stats = _capi.LoopStats()
        self.check = _capi.check

    def run_loop(self, L):
        C = self.C
        self.check(self.lib.smvs_ctx_restore_nodes(L["handle"]))
        self.check(self.lib.smvs_gn_run_loop(L["handle"], C.byref(L["params"]),
                                             C.byref(self.stats)))
        s = self.stats
        return s.newton_steps, s.active_patch_steps, s.linear_iterations

    def run_pass(self):
        """-> (active patch-steps, CG iterations, Newton steps) of one pass."""
        ps = its = ns = 0
        for L in self.loops:
            n, p, i = self.run_loop(L)
            ns += n; ps += p; its += i
        return ps, its, ns

    def verify(self):
        """The replay is the optimize() it was recorded from: per batch the
        same Newton steps, active patch-steps and CG iterations."""
        for L, e in zip(self.loops, self.log):
            got = self.run_loop(L)
            want = (e["newton_steps"], e["active_patch_steps"], e["cg_iterations"])
            if got != want or L["scale"] != e["scale"]:
                raise RuntimeError("replayed loop (scale %d, iteration %d) ran %r, the "
                                   "optimize() it was recorded from %r"
                                   % (L["scale"], L["iter"], got, want))

    def synchronize(self):
        for L in self.loops:
            self.check(self.lib.smvs_ctx_synchronize(L["handle"]))

    def profile_pass(self):
        """One more pass with HIP-event timing of every kernel class on the
        contexts' streams (untimed region) -> per batch (scale, nodes, Newton
        steps, active patch-steps, CG iterations, {class: (ms, launches)})."""
        C = self.C
        from smvs_amd import _capi
        out = []
        for L in self.loops:
            h = L["handle"]
            self.check(self.lib.smvs_profile_enable(h, 1))
            self.check(self.lib.smvs_profile_reset(h))
            n, p, i = self.run_loop(L)
            ms = (C.c_double * 8)(); cnt = (C.c_longlong * 8)()
            self.check(self.lib.smvs_profile_get(h, ms, cnt))
            self.check(self.lib.smvs_profile_enable(h, 0))
            out.append(dict(scale=L["scale"], nodes=L["nodes"], newton_steps=n,
                            active_patch_steps=p, cg_iterations=i,
                            kernels={k: (ms[j], int(cnt[j]))
                                     for j, k in enumerate(_capi.K_NAMES)}))
        return out

    def close(self):
        for L in self.loops:
            self.lib.smvs_ctx_destroy(L["handle"])
        self.loops = []


def patch_flops_table():
    """FP64 flops the patch kernel executes per active patch, per kernel form
    (= per samples-per-patch), from SQ instruction counters of one optimize()
    (tools/patch_flops.sh -> profiles/patch_flops_r6.json).  Without the file:
    round 3's constant for the 16-sample form, scaled by the samples."""
    try:
        with open(PATCH_FLOPS_FILE) as f:
            t = json.load(f)
        return {int(k): float(v) for k, v in t["flop_per_patch_by_scale"].items()}, \
            "profiles/patch_flops_r6.json (rocprofv3 --pmc SQ_INSTS_VALU_*_F64 / MFMA, committed)"
    except (OSError, KeyError, ValueError):
        per16 = FLOP_PER_PATCH
        return {2: per16, 3: per16, 4: 4 * per16, 5: 4 * per16, 6: 16 * per16}, \
            "estimate: round-3 counters of the 16-sample form x samples per patch / 16"


def loops_roofline(replay, steps, ms_per_step, with_peaks=True):
    """Roofline of the timed region's dominant kernel (the one with the largest
    share of the Newton loops of one optimize()), per_kernel for patch and
    solve, by_scale split."""
    batches = replay.profile_pass()
    peaks = measured_peaks() if with_peaks else None
    hbm_measured = (peaks or {}).get("hbm_read_GBps")
    flops_by_scale, flops_source = patch_flops_table()
    tot = {}
    for b in batches:
        for k, (ms, cnt) in b["kernels"].items():
            a = tot.setdefault(k, [0.0, 0])
            a[0] += ms; a[1] += cnt
    kernels = {k: dict(ms=round(v[0], 3), launches=int(v[1]),
                       avg_us=round(1e3 * v[0] / max(v[1], 1), 2)) for k, v in tot.items()}
    loop_ms = 1e3 * sum(e["loop_seconds"] for e in replay.log)
    busy_ms = sum(v[0] for v in tot.values())
    name = max(tot.items(), key=lambda kv: kv[1][0])[0]

    def hbm_fracs(achieved_GBps):
        out = dict(frac=round(achieved_GBps / HBM_PEAK_GBPS, 4))
        if hbm_measured:
            out["frac_of_measured_read_peak"] = round(achieved_GBps / hbm_measured, 4)
            out["measured_read_peak_GBps"] = hbm_measured
        return out

    lines = {}
    # Resident PCG, one launch per solve.  `achieved` follows the contract:
    # ALGORITHMIC bytes per launch -- SURVEY.md 8(d)'s per-unit figure for the PCG
    # phase (bytes per node and CG iteration: the upper half of H, P and the
    # vectors, what a streaming solver moves; 1,154 B with this repo's symmetric
    # storage, the survey prices the full stencil at 1.66 KB) x the nodes x the
    # iterations of the launch -- over the launch's duration.  The kernel holds H
    # in registers, so it moves almost none of them (design_bytes_per_launch:
    # its one pass over the per-patch systems + x and b) and the figure comes out
    # ABOVE the HBM peak: what bounds the kernel is the latency of its one
    # grid-wide exchange per iteration (frac_of_latency_floor).
    ms_k, cnt_k = tot.get("cg_resident", (0.0, 0))
    if cnt_k > 0:
        cg_bytes_node = CG_BYTES["cg_spmv"] + CG_BYTES["cg_update"]
        alg = sum(cg_bytes_node * b["nodes"] * b["cg_iterations"] for b in batches)
        byts = sum(RESIDENT_BYTES_PER_PATCH * b["active_patch_steps"]
                   + RESIDENT_BYTES_PER_NODE * b["nodes"] * b["kernels"]["cg_resident"][1]
                   for b in batches)
        its = sum(b["cg_iterations"] for b in batches)
        avg_s = 1e-3 * ms_k / cnt_k
        ach = alg / cnt_k / avg_s / 1e9
        moved = byts / cnt_k / avg_s / 1e9
        lines["cg_resident"] = dict(
            bound="hbm", achieved=round(ach, 1), peak=HBM_PEAK_GBPS, unit="GB/s",
            **hbm_fracs(ach), bytes_per_launch=int(alg / cnt_k),
            bytes_per_node_and_iteration=cg_bytes_node, avg_us=round(1e6 * avg_s, 2),
            launches=int(cnt_k), iterations_per_launch=round(its / cnt_k, 1),
            design_bytes_per_launch=int(byts / cnt_k), design_GBps=round(moved, 1),
            frac_of_what_the_design_moves=round(moved / HBM_PEAK_GBPS, 4),
            us_per_iteration_incl_prologue=round(1e3 * ms_k / max(its, 1), 2),
            latency_floor_us_per_iteration=round(1e6 * EXCHANGE_FLOOR_S, 2),
            frac_of_latency_floor=round(its * EXCHANGE_FLOOR_S / (1e-3 * ms_k), 4),
            note="whole PCG solve in one launch, H assembled into registers from the "
                 "per-patch systems (%d B per live patch + %d B per node, read once = "
                 "design_bytes_per_launch).  achieved / frac price the launch at the "
                 "algorithmic bytes of SURVEY.md 8(d) (%d B per node and iteration x nodes x "
                 "iterations): above 1 because a resident H is not re-read, NOT a claim of "
                 "HBM utilisation -- the kernel's HBM traffic is design_GBps "
                 "(frac_of_what_the_design_moves) and its bound is the latency of one "
                 "two-hop grid-wide exchange per iteration: frac_of_latency_floor prices an "
                 "iteration at two idle cross-CU hand-offs (MI355X_MICROARCH.md, 1.1 us each)"
                 % (RESIDENT_BYTES_PER_PATCH, RESIDENT_BYTES_PER_NODE, cg_bytes_node))
    ms_k, cnt_k = tot.get("patch", (0.0, 0))
    if cnt_k > 0:
        flops = sum(flops_by_scale.get(b["scale"], FLOP_PER_PATCH) * b["active_patch_steps"]
                    for b in batches)
        avg_s = 1e-3 * ms_k / cnt_k
        ach = flops / cnt_k / avg_s / 1e12
        by = {}
        for b in batches:
            d = by.setdefault(str(b["scale"]), [0.0, 0.0])
            d[0] += flops_by_scale.get(b["scale"], FLOP_PER_PATCH) * b["active_patch_steps"]
            d[1] += 1e-3 * b["kernels"]["patch"][0]
        lines["patch"] = dict(
            bound="fp64-issue", achieved=round(ach, 3), peak=FP64_PEAK_TFLOPS, unit="TFLOP/s",
            frac=round(ach / FP64_PEAK_TFLOPS, 4), flops_per_launch=flops / cnt_k,
            avg_us=round(1e6 * avg_s, 2), launches=int(cnt_k),
            flop_per_patch_by_scale={str(k): v for k, v in sorted(flops_by_scale.items())},
            flop_per_patch_source=flops_source,
            frac_with_survey_flops=round(
                sum(FLOP_PER_PATCH_SURVEY / 16.0 * SAMPLES_PER_PATCH.get(b["scale"], 16)
                    * b["active_patch_steps"] for b in batches)
                / cnt_k / avg_s / 1e12 / FP64_PEAK_TFLOPS, 4),
            note="frac = FP64 flops the factored kernel EXECUTES (SQ counters per kernel "
                 "form) over the peak: a pipe utilisation; SURVEY.md 8(d) prices the "
                 "reference's unfactored rows at 0.50 MFLOP per 16-sample patch "
                 "(frac_with_survey_flops, above the executed figure by the factor the "
                 "6 x 6 reformulation saves)",
            frac_by_scale={k: round(v[0] / v[1] / 1e12 / FP64_PEAK_TFLOPS, 4)
                           for k, v in by.items() if v[1] > 0})
    # committed counter traffic of the dominant kernel (own rocprofv3 --pmc passes)
    traffic = traffic_source = None
    for tf in ("traffic_r6.json", "traffic_r5.json"):
        tfile = os.path.join(ROOT, "profiles", tf)
        if os.path.exists(tfile):
            with open(tfile) as f:
                traffic = json.load(f).get(name)
            traffic_source = ("profiles/" + tf + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in "
                              "their own passes, committed; not measured in this run)")
            break
    by_scale = {}
    for b, e in zip(batches, replay.log):
        d = by_scale.setdefault(str(b["scale"]), dict(active_patch_steps=0, newton_steps=0,
                                                      cg_iterations=0, kernel_ms={}))
        d["active_patch_steps"] += b["active_patch_steps"]
        d["newton_steps"] += b["newton_steps"]
        d["cg_iterations"] += b["cg_iterations"]
        for k, (ms, cnt) in b["kernels"].items():
            if cnt:
                d["kernel_ms"][k] = round(d["kernel_ms"].get(k, 0.0) + ms, 3)
    # T_min / T of the whole region, T_min = sum over the steps of
    # max(F_construct / pi, B_construct / beta) + N_cg B_cg / beta with the measured
    # per-batch patches, nodes and iterations
    beta, pi = HBM_PEAK_GBPS * 1e9, FP64_PEAK_TFLOPS * 1e12
    cgb = CG_BYTES["cg_spmv"] + CG_BYTES["cg_update"]
    t_exec = t_survey = 0.0
    for b in batches:
        t_cg = b["cg_iterations"] * cgb * b["nodes"] / beta
        b_con = b["active_patch_steps"] * BYTES_PER_PATCH / beta
        t_exec += max(flops_by_scale.get(b["scale"], FLOP_PER_PATCH) * b["active_patch_steps"] / pi,
                      b_con) + t_cg
        t_survey += max(FLOP_PER_PATCH_SURVEY / 16.0 * SAMPLES_PER_PATCH.get(b["scale"], 16)
                        * b["active_patch_steps"] / pi, b_con) + t_cg
    region = dict(t_min_ms_executed_flops=round(1e3 * t_exec, 3),
                  t_min_ms_survey_flops=round(1e3 * t_survey, 3),
                  frac_executed_flops=round(1e3 * t_exec / ms_per_step, 4),
                  frac_survey_flops=round(1e3 * t_survey / ms_per_step, 4),
                  note="SURVEY.md 8(d)'s T_min / T for one pass: construction at the FP64 peak "
                       "(executed flops of the factored kernel / the survey's 0.50 MFLOP per "
                       "16-sample patch) + the PCG streamed from HBM at %d B per node and "
                       "iteration; above 1 where the resident solver does not stream" % cgb)
    out = dict(kernel=name, traffic=traffic, traffic_source=traffic_source, kernels=kernels,
               region_model=region,
               region_share={k: round(v[0] / max(busy_ms, 1e-9), 4) for k, v in tot.items() if v[1]},
               region_kernel_ms=round(busy_ms, 3), region_wall_ms=round(ms_per_step, 3),
               per_kernel=lines, by_scale=by_scale,
               peaks_assumed=dict(hbm_GBps=HBM_PEAK_GBPS, fp64_TFLOPs=FP64_PEAK_TFLOPS),
               peaks_measured=peaks,
               note="kernel = the class with the largest share of the region's kernel time "
                    "(HIP events on the contexts' streams, one profiled pass); top-level "
                    "bound / achieved / peak / frac are that kernel's")
    if name in lines:
        out.update({k: lines[name][k] for k in ("bound", "achieved", "peak", "unit", "frac")})
        out["avg_us"] = lines[name]["avg_us"]
    return out


# --------------------------------------------------------------- CPU baseline
def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or "unknown"


def _oracle_flags():
    try:
        with open(os.path.join(ROOT, "oracle", "Makefile")) as f:
            for line in f:
                if line.startswith("CFLAGS"):
                    return "gcc " + line.split(":=", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(small=False):
    """The oracle (CPU restatement of the reference; SSE2 two-lane inner loop
    like lib/gauss_newton_step.cc:252-333) on the SAME region as `value`: the
    Newton loops of one whole DepthOptimizer::optimize of the same scene, timed
    by the oracle's own loop timers (orc_opt_log.loop_seconds around
    lib/depth_optimizer.cc:219-304).  One optimize() gives both numbers:
      value      ONE thread -- the reference's per-view behaviour (it
                 parallelises over views, app/smvsrecon.cc:658-733, not inside
                 one) -- on a bounded sample: the first batch of every scale
                 (5 of the 15 batches, a third of the patch-steps in the
                 region's own proportions);
      all_cores  OpenMP over the patches of the view (up to 64 threads) in the
                 other ten batches -- and in everything between the batches,
                 which neither side's timers count."""
    from oracle import pyoracle
    L = pyoracle.lib()
    nproc = os.cpu_count() or 1
    th = min(nproc, 64)
    inp = scene_inputs(0, small)
    L.orc_set_threads(th)
    L.orc_set_first_batch_loop_threads(1)
    t = time.perf_counter()
    try:
        r = pyoracle.optimize(inp, regularization=REG, num_iterations=5, min_scale=SCALE)
    finally:
        L.orc_set_first_batch_loop_threads(0)
        L.orc_set_threads(1)
    wall = time.perf_counter() - t
    first = [e for e in r["log"] if e["iter"] == 0]
    rest = [e for e in r["log"] if e["iter"] != 0]
    aps1 = sum(e["active_patch_steps"] for e in first)
    s1 = sum(e["loop_seconds"] for e in first)
    w, h = (480, 270) if small else (W, H)
    out = dict(value=aps1 / s1, unit="active-patch-steps/s", cores=1, kind="port",
               nproc=nproc, cpu_model=_cpu_model(), compiler_flags=_oracle_flags(),
               active_patch_steps=int(aps1), newton_loop_s=round(s1, 2),
               by_scale={str(e["scale"]): dict(active_patch_steps=int(e["active_patch_steps"]),
                                               newton_steps=int(e["newton_steps"]),
                                               cg_iterations=int(e["cg_iterations"]),
                                               loop_s=round(e["loop_seconds"], 3))
                         for e in first},
               sample="Newton loops (the oracle's own loop timers) of the first batch of "
                      "every scale (%d .. %d: %d of %d batches, %d of %d active patch-steps) "
                      "of one whole optimize() of the %dx%d / %d-neighbour sphere scene, "
                      "--no-sgm, 1 thread: the same scene, region and units as `value`"
                      % (max(e["scale"] for e in first), SCALE, len(first), len(r["log"]),
                         aps1, sum(e["active_patch_steps"] for e in r["log"]), w, h, NSUBS))
    if rest:
        apsN = sum(e["active_patch_steps"] for e in rest)
        sN = sum(e["loop_seconds"] for e in rest)
        out["all_cores"] = dict(value=apsN / sN, unit="active-patch-steps/s", threads=th,
                                active_patch_steps=int(apsN), newton_loop_s=round(sN, 2),
                                sample="the other %d batches of the same optimize(), OpenMP "
                                       "over the patches of the view" % len(rest))
    out["optimize_wall_s"] = round(wall, 1)
    return out


# ------------------------------------------------------------------- roofline
def measured_peaks():
    """tools/peaks.py on this GPU (HBM copy / read, FP64 VALU and MFMA)."""
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import peaks
        r = peaks.run()
        return dict(hbm_copy_GBps=round(r["hbm_copy_GBps"], 1),
                    hbm_read_GBps=round(r["hbm_read_GBps"], 1),
                    hbm_write_GBps=round(r["hbm_write_GBps"], 1),
                    fp64_valu_fma_TFLOPs=round(r["fp64_valu_fma_TFLOPs"]["4"], 2),
                    fp64_mfma_16x16x4_TFLOPs=round(r["fp64_mfma_16x16x4_TFLOPs"]["2"], 2),
                    fp64_valu_and_mfma_share_one_pipe=bool(
                        r["valu_plus_mfma_same_simd_ms"]["both"]
                        > 0.9 * (r["valu_plus_mfma_same_simd_ms"]["valu_alone"]
                                 + r["valu_plus_mfma_same_simd_ms"]["mfma_alone"])))
    except Exception as e:  # the peaks are a report, never a reason to fail
        return dict(error=str(e))


def profile_pass(ctx, prob, steps, lighting=None):
    """The same steps once more, one Newton step per call with HIP-event
    timing of every kernel class on the context's stream: per-step active
    sets, CG iterations and kernel times (untimed region)."""
    surf = prob["surf"]
    ctx.set_nodes(surf["nodes"])
    ctx.set_active(None)
    ctx.profile(True)
    ctx.profile_reset()
    n_init = int(surf["node_valid"].sum())
    per_step = []
    done = 0
    n_act = n_init
    while done < steps:
        if n_act <= n_init // 20:
            ctx.set_nodes(surf["nodes"])
            ctx.set_active(None)
            n_act = n_init
        st = ctx.run_loop(REG, lighting=lighting, max_newton_steps=1, reset_active=False)
        if st["newton_steps"] == 0:
            raise RuntimeError("Newton loop made no progress")
        per_step.append((n_act, st["active_patch_steps"], st["linear_iterations"]))
        n_act = st["final_active_nodes"]
        done += 1
    prof = ctx.profile_get()
    ctx.profile(False)
    return per_step, prof


def roofline(ctx, prob, steps, ms_per_step, lighting=None, with_peaks=True):
    per_step, prof = profile_pass(ctx, prob, steps, lighting)
    kernels = {k: dict(ms=round(v[0], 3), launches=int(v[1]),
                       avg_us=round(1e3 * v[0] / max(v[1], 1), 2))
               for k, v in prof.items()}
    name, (ms, cnt) = max(prof.items(), key=lambda kv: kv[1][0])
    n_nodes = ctx.num_nodes
    patch_steps = sum(p for _, p, _ in per_step)
    cg_its = sum(c for _, _, c in per_step)
    # HBM bytes per launch from the PMC counters (FETCH_SIZE / WRITE_SIZE in
    # their own rocprofv3 passes, tools/collect_profiles.sh): a committed
    # measurement of this command on an earlier box, NOT taken in this run
    traffic = traffic_source = None
    for tf in ("traffic_r5.json", "traffic_r4.json", "traffic_r3.json", "traffic_r2.json",
               "traffic_r1.json"):
        tfile = os.path.join(ROOT, "profiles", tf)
        if os.path.exists(tfile):
            with open(tfile) as f:
                traffic = json.load(f).get(name)
            traffic_source = "profiles/" + tf + " (rocprofv3 --pmc, committed; not measured in this run)"
            break
    # ---- T_min of the whole step over the measured time ----
    #   T_min = sum_steps [ max(F_construct / pi, B_construct / beta)
    #                       + N_cg * B_cg / beta ]
    # with the measured per-step active patches, active nodes and CG iterations;
    # F = the flops the factored construction executes (the survey's 0.50 MFLOP
    # per patch prices the unfactored rows and is reported beside it).
    beta = HBM_PEAK_GBPS * 1e9
    pi = FP64_PEAK_TFLOPS * 1e12
    peaks = measured_peaks() if with_peaks else None
    hbm_measured = (peaks or {}).get("hbm_read_GBps")   # tools/peaks.py on this box

    def hbm_fracs(achieved_GBps):
        """Fractions of the assumed 8 TB/s and of the read peak measured here."""
        out = dict(frac=round(achieved_GBps / HBM_PEAK_GBPS, 4))
        if hbm_measured:
            out["frac_of_measured_read_peak"] = round(achieved_GBps / hbm_measured, 4)
            out["measured_read_peak_GBps"] = hbm_measured
        return out
    cg_bytes_node = CG_BYTES["cg_spmv"] + CG_BYTES["cg_update"]
    t_min = t_min_survey = t_min_resident = 0.0
    for nodes_act, patches, its in per_step:
        b_construct = patches * BYTES_PER_PATCH / beta
        t_cg = its * cg_bytes_node * nodes_act / beta
        t_construct = max(patches * FLOP_PER_PATCH / pi, b_construct)
        t_min += t_construct + t_cg
        t_min_survey += max(patches * FLOP_PER_PATCH_SURVEY / pi, b_construct) + t_cg
        # the design that is built: H in registers, one pass over its input,
        # one exchange per iteration at its latency floor
        t_min_resident += t_construct + (RESIDENT_BYTES_PER_PATCH * patches
                                         + RESIDENT_BYTES_PER_NODE * n_nodes) / beta \
            + its * EXCHANGE_FLOOR_S
    t_meas = ms_per_step * 1e-3 * steps
    out = dict(kernel=name, traffic=traffic, traffic_source=traffic_source, kernels=kernels,
               step_frac=round(t_min_resident / t_meas, 4),
               step_t_min_us=round(1e6 * t_min_resident / steps, 1),
               step_frac_streaming_model=round(t_min / t_meas, 4),
               step_t_min_streaming_model_us=round(1e6 * t_min / steps, 1),
               step_frac_with_survey_flops=round(t_min_survey / t_meas, 4),
               step_note="step_frac = T_min / T with the measured per-step active "
                         "patches / nodes / CG iterations: construction %.3f MFLOP per "
                         "patch (executed by the factored kernel) at %.1f TFLOP/s, the "
                         "resident solver's one pass over its input (%d B per live patch + "
                         "%d B per node) at %.0f GB/s and %.1f us per CG iteration for "
                         "its one two-hop exchange (two idle cross-CU hand-offs, "
                         "MI355X_MICROARCH.md); step_frac_streaming_model is SURVEY.md "
                         "8(d)'s formula (%d B per active node and CG iteration streamed "
                         "from HBM), which the resident design no longer moves; the survey's "
                         "0.50 MFLOP per patch (unfactored rows) gives "
                         "step_frac_with_survey_flops"
                         % (FLOP_PER_PATCH / 1e6, FP64_PEAK_TFLOPS, RESIDENT_BYTES_PER_PATCH,
                            RESIDENT_BYTES_PER_NODE, HBM_PEAK_GBPS, 1e6 * EXCHANGE_FLOOR_S,
                            cg_bytes_node),
               peaks_assumed=dict(hbm_GBps=HBM_PEAK_GBPS, fp64_TFLOPs=FP64_PEAK_TFLOPS),
               peaks_measured=peaks)
    def cg_resident_line():
        # One launch per solve.  Inside the Newton loop the kernel assembles H,
        # g, P itself from the per-patch systems (1,152 + 128 B per live patch,
        # read once) and writes x and b; through smvs_cg_solve it reads the
        # assembled upper half of H, P and g instead.  Either way the matrix
        # then stays in registers.
        ms_k, cnt_k = prof["cg_resident"]
        fused = prof["assemble"][1] == 0
        if fused:
            bytes_per_launch = RESIDENT_BYTES_PER_PATCH * patch_steps / max(cnt_k, 1) + 64 * n_nodes
        else:
            bytes_per_launch = (5 * 128 + 128 + 32 + 32 + 32) * n_nodes
        avg_s = ms_k * 1e-3 / max(cnt_k, 1)
        achieved = bytes_per_launch / avg_s / 1e9
        return dict(bound="hbm", achieved=round(achieved, 1), peak=HBM_PEAK_GBPS,
                    unit="GB/s", **hbm_fracs(achieved),
                    bytes_per_launch=int(bytes_per_launch), avg_us=round(avg_s * 1e6, 2),
                    note="whole PCG solve in one launch (%.1f iterations on average, "
                         "system assembled in the kernel: %s): after the one pass over "
                         "its input the kernel is bound by the latency of its one "
                         "grid-wide exchange per iteration, not by HBM"
                         % (cg_its / max(cnt_k, 1), "yes" if fused else "no"))

    def patch_line():
        # gn_patch_kernel: FP64 arithmetic.  The vector FMA and v_mfma_f64 share
        # one pipe on gfx950 (tools/peaks.py), so the bound is FP64 issue, and
        # `frac` = executed FP64 flops / peak is a pipe UTILISATION (every
        # executed instruction counts as useful): an upper bound on the useful
        # fraction.
        ms_k, cnt_k = prof["patch"]
        flops = FLOP_PER_PATCH * patch_steps / max(cnt_k, 1)
        avg_s = ms_k * 1e-3 / max(cnt_k, 1)
        achieved = flops / avg_s / 1e12
        return dict(bound="fp64-issue", achieved=round(achieved, 3), peak=FP64_PEAK_TFLOPS,
                    unit="TFLOP/s", frac=round(achieved / FP64_PEAK_TFLOPS, 4),
                    fp64_util=round(achieved / FP64_PEAK_TFLOPS, 4),
                    flops_per_launch=flops, avg_us=round(avg_s * 1e6, 2))

    # the two kernels that share a Newton step about evenly are both described;
    # the top-level fields are those of the one with the larger share
    lines = {}
    if prof["patch"][1] > 0:
        lines["patch"] = patch_line()
    if prof.get("cg_resident", (0, 0))[1] > 0:
        lines["cg_resident"] = cg_resident_line()
    out["per_kernel"] = lines
    if name in lines:
        out.update(lines[name])
    elif name in CG_BYTES:
        # launches after convergence are no-ops: their time stays in the
        # numerator, the denominator counts the launches that did work
        bytes_per_launch = CG_BYTES[name] * n_nodes
        work = max(cg_its, 1)
        avg_s = ms * 1e-3 / work
        achieved = bytes_per_launch / avg_s / 1e9
        out.update(bound="hbm", achieved=round(achieved, 1), peak=HBM_PEAK_GBPS,
                   unit="GB/s", **hbm_fracs(achieved),
                   bytes_per_launch=bytes_per_launch, avg_us=round(avg_s * 1e6, 2),
                   launches_with_work=work)
    out["cg_iterations"] = cg_its
    out["cg_launch_pairs"] = int(prof["cg_spmv"][1])
    out["cg_resident_solves"] = int(prof.get("cg_resident", (0, 0))[1])
    return out



# ----------------------------------------------------------------- secondary
def secondary_workloads(args):
    """Beside the headline: (1) BASELINE.md's own timed region -- all Newton
    loops of all scales of ONE DepthOptimizer::optimize (lib/depth_optimizer.cc:
    53-162) through the C++ host mirror: sum of active patch-steps / sum of
    in-loop seconds; (2) whole reference views per second through the C++
    ViewQueue (the per-view task of app/smvsrecon.cc:658-733: StereoViews,
    optional SGM front end, optimize, maps), one and several views in flight;
    (3) per-kernel times of the SGM front end with their HBM rooflines."""
    import ctypes as C
    from smvs_amd import synth, host, _capi
    w, h = (480, 270) if args.small else (W, H)
    inp = synth.pipeline_inputs("sphere", w, h, NSUBS, flen=1.2)
    out = {}
    # (1) one optimize(), warm (the first call pays library / pool start-up)
    host.optimize(inp, regularization=REG, num_iterations=5, min_scale=SCALE)
    t = time.perf_counter()
    r = host.optimize(inp, regularization=REG, num_iterations=5, min_scale=SCALE)
    wall = time.perf_counter() - t
    aps = sum(e["active_patch_steps"] for e in r["log"])
    loop_s = sum(e["loop_seconds"] for e in r["log"])
    by_scale = {}
    for e in r["log"]:
        d = by_scale.setdefault(str(e["scale"]), dict(active_patch_steps=0, loop_ms=0.0,
                                                      newton_steps=0, cg_iterations=0))
        d["active_patch_steps"] += e["active_patch_steps"]
        d["loop_ms"] = round(d["loop_ms"] + 1e3 * e["loop_seconds"], 3)
        d["newton_steps"] += e["newton_steps"]
        d["cg_iterations"] += e["cg_iterations"]
    out["optimize"] = dict(
        value=aps / loop_s, unit="active-patch-steps/s",
        active_patch_steps=aps, newton_loop_ms=round(1e3 * loop_s, 3),
        optimize_wall_ms=round(1e3 * wall, 1), batches=len(r["log"]), by_scale=by_scale,
        note="all Newton loops of all scales (init .. %d) of one optimize() of the "
             "%dx%d / %d-neighbour sphere scene, --no-sgm: sum of active patch-steps over "
             "the sum of the loops' wall time (BASELINE.md's timed region)" % (SCALE, w, h, NSUBS))
    # (2) views per second, whole per-view pipeline
    views = {}
    for label, sgm_scale in (("no_sgm", None), ("sgm", 1)):
        # (warm-up: library, page-locked pools, the workers' contexts of this mode -- and
        # the GPU's clocks: this section follows half a minute of CPU-only work (the CPU
        # baseline), and one view at a time leaves the GPU idle a fifth of the time, so the
        # busiest configuration runs first)
        host.optimize_views(inp, 16, regularization=REG, min_scale=SCALE, sgm_scale=sgm_scale,
                            views_in_flight=8)
        for in_flight in (8, 4, 1):
            jobs = 4 if in_flight == 1 else 3 * in_flight
            best = None
            for rep in range(2):
                v = host.optimize_views(inp, jobs, regularization=REG, min_scale=SCALE,
                                        sgm_scale=sgm_scale, views_in_flight=in_flight)
                if best is None or v["views_per_s"] > best["views_per_s"]:
                    best = v
            views["%s_in_flight_%d" % (label, in_flight)] = dict(
                views_per_s=round(best["views_per_s"], 2), views=jobs,
                mean_task_ms=round(1e3 * float(np.mean(best["job_seconds"])), 1))
    out["views_per_s"] = dict(per_gpu=views,
        note="whole per-view tasks (9 x StereoView::create, [SGM front end,] optimize() of "
             "all scales, depth + normal maps) through smvs_amd::ViewQueue on one GPU, "
             "%dx%d, %d neighbours; the better of two runs of each entry" % (w, h, NSUBS))
    # (3) SGM front end kernels
    lib = _capi.load()
    host.sgm_depth(inp, sgm_scale=1)
    lib.smvs_sgm_profile(1, None, None)
    host.sgm_depth(inp, sgm_scale=1)
    ms = (C.c_double * 8)(); cnt = (C.c_longlong * 8)()
    lib.smvs_sgm_profile(0, ms, cnt)
    sw, sh = (w + 1) // 2, (h + 1) // 2
    cells = sw * sh * 128
    names = ["census", "warp", "cost", "paths", "wta", "lr_check", "merge", "bilateral"]
    # algorithmic HBM bytes per launch (one run_sgm at sw x sh x 128):
    bytes_per = dict(paths=cells * (8 * 1 + 8 * 1),   # per direction: cost in, path byte (L - C) out
                     cost=cells * 2,                   # warped plane in, cost out
                     warp=cells * 1,                   # warped plane out (neighbour image cached)
                     wta=cells * 9)                    # cost + eight path bytes in (S = 8 C + sum, on the fly)
    sgm = {}
    for i, nme in enumerate(names):
        if cnt[i] == 0:
            continue
        avg_s = 1e-3 * ms[i] / cnt[i]
        line = dict(avg_us=round(1e6 * avg_s, 1), launches=int(cnt[i]))
        if nme in bytes_per:
            ach = bytes_per[nme] / avg_s / 1e9
            line.update(bound="hbm", bytes_per_launch=int(bytes_per[nme]),
                        achieved=round(ach, 1), peak=HBM_PEAK_GBPS, unit="GB/s",
                        frac=round(ach / HBM_PEAK_GBPS, 4))
        sgm[nme] = line
    out["sgm_front_end"] = dict(kernels=sgm, size=[sw, sh, 128],
        note="HIP-event times of one reconstruct_sgm_depth_for_view (4 x run_sgm); "
             "algorithmic bytes per launch: paths 8 x (1 B cost in + 1 B path byte out) per "
             "cell, wta (sum of the eight path bytes + 8 C, winner-takes-all) 9 B in per cell, "
             "cost 1 B in + 1 B out, warp 1 B out")
    return out

# ------------------------------------------------- whole views on N GPUs
def _views_inputs(small, shading):
    from smvs_amd import synth
    w, h = (480, 270) if small else (W, H)
    lighting = None
    if shading:
        rng = np.random.default_rng(3000)
        lighting = np.zeros(16); lighting[0] = 0.9
        lighting[1:4] = rng.uniform(-0.2, 0.2, 3)
    return synth.pipeline_inputs("sphere", w, h, NSUBS, flen=1.2, lighting=lighting), w, h


def views_worker(args):
    """`bench.py --views-worker`: one process of the multi-GPU views workload.
    Runs --views whole per-view tasks (9 x StereoView::create, SGM front end,
    optimize() of all scales [-S with --shading], depth + normal maps;
    app/smvsrecon.cc:658-733) through the C++ ViewQueue on devices
    [--first-device, --first-device + --num-devices), --views-in-flight per
    GPU.  After its warm-up it waits for the go file of --sync-dir so that all
    workers of a measurement start together.  Prints one JSON line."""
    from smvs_amd import host
    inp, w, h = _views_inputs(args.small, args.shading)
    kw = dict(regularization=REG, min_scale=SCALE, sgm_scale=1, use_shading=args.shading,
              first_device=args.first_device, num_devices=args.num_devices)
    host.optimize_views(inp, max(2, args.num_devices * args.views_in_flight),
                        views_in_flight=args.views_in_flight, **kw)
    if args.sync_dir:
        open(os.path.join(args.sync_dir, "ready_%d" % args.worker_id), "w").close()
        go = os.path.join(args.sync_dir, "go")
        t_wait = time.perf_counter()
        while not os.path.exists(go):
            if time.perf_counter() - t_wait > 300:
                raise RuntimeError("views worker: no go signal")
            time.sleep(0.0005)
    t0 = time.time()
    v = host.optimize_views(inp, args.views, views_in_flight=args.views_in_flight, **kw)
    t1 = time.time()
    aps = sum(e["active_patch_steps"] for lg in v["logs"] for e in lg)
    print(json.dumps(dict(worker=args.worker_id, views=args.views, t_start=t0, t_end=t1,
                          seconds=t1 - t0, views_per_s=args.views / (t1 - t0),
                          mean_task_ms=round(1e3 * float(np.mean(v["job_seconds"])), 1),
                          active_patch_steps=int(aps))))


def multi_gpu_views(args, n_gpus, shading):
    """What can fail to scale (app/smvsrecon.cc:658-733: one task per reference
    view): whole per-view tasks on the N GPUs of this node, measured two ways --
    (a) N processes, one per GPU, started together (how the driver's ranks
    run), (b) ONE process driving all N GPUs through ViewQueue(N,
    views_in_flight), the reference's single process with a thread pool.  Each
    in fresh child processes after the distributed job has ended: a failure
    there cannot take the headline with it."""
    import tempfile
    per_gpu_views = max(args.views_per_rank, 1)
    in_flight = max(args.views_in_flight, 4)
    base = [sys.executable, os.path.abspath(__file__), "--views-worker",
            "--views-in-flight", str(in_flight)]
    if args.small:
        base.append("--small")
    if shading:
        base.append("--shading")
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
              "GROUP_RANK", "LOCAL_WORLD_SIZE", "ROLE_RANK", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    out = dict(views_in_flight_per_gpu=in_flight, views_per_gpu=per_gpu_views,
               shading=bool(shading), nproc=os.cpu_count(),
               note="whole per-view tasks (9 x StereoView::create, SGM front end, "
                    "optimize() of all scales%s, depth + normal maps), %dx%d, %d neighbours"
                    % (" with -S (SH lighting fit + shading residual)" if shading else "",
                       *((480, 270) if args.small else (W, H)), NSUBS))

    def run(workers):
        """workers: list of (first_device, num_devices, views)."""
        with tempfile.TemporaryDirectory() as d:
            procs = []
            for i, (first, num, views) in enumerate(workers):
                cmd = base + ["--first-device", str(first), "--num-devices", str(num),
                              "--views", str(views), "--sync-dir", d, "--worker-id", str(i)]
                procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE,
                                              stderr=subprocess.PIPE, text=True))
            t_wait = time.perf_counter()
            while not all(os.path.exists(os.path.join(d, "ready_%d" % i))
                          for i in range(len(workers))):
                if any(p.poll() is not None for p in procs) \
                        or time.perf_counter() - t_wait > 240:
                    for p in procs:
                        p.kill()
                    errs = [p.communicate()[1][-300:] for p in procs]
                    raise RuntimeError("views worker failed to start: %r" % errs)
                time.sleep(0.01)
            open(os.path.join(d, "go"), "w").close()
            res = []
            for p in procs:
                so, se = p.communicate(timeout=600)
                if p.returncode != 0:
                    raise RuntimeError("views worker: %s" % se[-500:])
                res.append(json.loads(so.strip().splitlines()[-1]))
        span = max(r["t_end"] for r in res) - min(r["t_start"] for r in res)
        total = sum(r["views"] for r in res)
        return dict(views=total, seconds=round(span, 3), views_per_s=round(total / span, 2),
                    views_per_s_per_gpu=round(total / span / n_gpus, 2),
                    per_worker_views_per_s=[round(r["views_per_s"], 2) for r in res],
                    mean_task_ms=round(float(np.mean([r["mean_task_ms"] for r in res])), 1),
                    host_threads=len(workers) + n_gpus * in_flight)

    try:
        out["one_process_per_gpu"] = run([(g, 1, per_gpu_views) for g in range(n_gpus)])
    except Exception as e:
        out["one_process_per_gpu"] = dict(error=repr(e))
    try:
        out["one_process_view_queue"] = run([(0, n_gpus, per_gpu_views * n_gpus)])
    except Exception as e:
        out["one_process_view_queue"] = dict(error=repr(e))
    if n_gpus > 1:
        try:
            out["single_gpu_reference"] = run([(0, 1, per_gpu_views)])
        except Exception as e:
            out["single_gpu_reference"] = dict(error=repr(e))
    return out


def optimize_workload(args):
    """--workload optimize: a "step" is one whole DepthOptimizer::optimize of one
    reference view (all scales, init .. 2) through the C++ host mirror; value =
    sum of active patch-steps over the sum of the Newton loops' wall time
    (BASELINE.md's timed region), ms_per_step = wall time of one optimize()."""
    from smvs_amd import synth, host
    import smvs_amd
    if smvs_amd.device_count() < 1:
        raise RuntimeError("bench.py needs a GPU")
    w, h = (480, 270) if args.small else (W, H)
    inp = synth.pipeline_inputs("sphere", w, h, NSUBS, flen=1.2)
    for _ in range(max(args.warmup, 1)):
        host.optimize(inp, regularization=REG, num_iterations=5, min_scale=SCALE)
    aps = 0
    loop_s = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        r = host.optimize(inp, regularization=REG, num_iterations=5, min_scale=SCALE)
        aps += sum(e["active_patch_steps"] for e in r["log"])
        loop_s += sum(e["loop_seconds"] for e in r["log"])
    wall = time.perf_counter() - t0
    print(json.dumps({
        "metric": "Gauss-Newton iters/sec x active patches over all Newton loops of optimize(), "
                  "1920x1080 ref view, 8 neighbours",
        "value": aps / loop_s, "unit": "active-patch-steps/s", "n_gpus": 1,
        "steps": args.steps, "warmup": max(args.warmup, 1),
        "ms_per_step": 1e3 * wall / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "configs[1], whole optimize(): %dx%d synthetic textured sphere, "
                               "1 ref + %d neighbours, --no-sgm, scales init .. %d, 5 iterations "
                               "per scale; a step = one optimize()" % (w, h, NSUBS, SCALE),
                   "regularization": REG,
                   "newton_loop_ms_per_optimize": 1e3 * loop_s / args.steps,
                   "active_patch_steps_per_optimize": aps / args.steps},
        "roofline": None, "cpu_baseline": None}))


def scale2_replay(args):
    """`--workload newton_steps` in a child process: the headline of rounds
    1-5 (scale-2 Newton steps from a perturbed start surface)."""
    cmd = [sys.executable, os.path.abspath(__file__), "--workload", "newton_steps",
           "--steps", "20", "--warmup", "5", "--repeats", "30", "--no-cpu-baseline",
           "--no-secondary", "--no-peaks"]
    if args.small:
        cmd.append("--small")
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    line = json.loads([l for l in res.stdout.splitlines() if l.strip().startswith("{")][-1])
    roof = line.get("roofline") or {}
    return dict(value=line["value"], unit=line["unit"], ms_per_step=line["ms_per_step"],
                steps=line["steps"], cg_iterations_per_step=line["config"]["cg_iterations_per_step"],
                per_kernel=roof.get("per_kernel"), step_frac=roof.get("step_frac"),
                note="a step = one Newton step of the scale-2 loop of the same view, replayed "
                     "from a perturbed start surface (`python bench.py --workload newton_steps`): "
                     "fewer CG iterations per step than optimize()'s own scale-2 batches")


# ----------------------------------------------------------------------- main
def respawn_under_torchrun(args):
    """`python bench.py --gpus N` without a torch.distributed environment:
    become N ranks (one per GPU) on this node."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--repeats", type=int, default=100,
                    help="how often the K-step region is measured (median reported)")
    ap.add_argument("--config", type=int, default=1, choices=(1, 5),
                    help="1: BASELINE configs[1] (headline); 5: configs[4], views sharded "
                         "over the GPUs, shading-aware")
    ap.add_argument("--views-per-rank", type=int, default=8, help="--config 5 only")
    ap.add_argument("--shared-lighting", action="store_true",
                    help="--config 5: RCCL all-reduce of the SH normal equations over "
                         "the lock-step round (default: per-view lighting, the "
                         "reference's behaviour)")
    ap.add_argument("--small", action="store_true", help="480x270 debug size")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-only", action="store_true",
                    help="print the CPU baseline JSON and exit (how the main run obtains it: "
                         "in a child process, after the GPU phase)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary workloads (whole optimize(), views per second, "
                         "SGM kernels)")
    ap.add_argument("--no-peaks", action="store_true",
                    help="skip the peak microbenchmarks (profiling runs)")
    ap.add_argument("--workload", default="optimize_loops",
                    choices=("optimize_loops", "newton_steps", "optimize"),
                    help="optimize_loops (default, the headline): BASELINE.md's timed region "
                         "-- all Newton loops of all scales of one optimize() of configs[1], "
                         "replayed from their recorded start states (a step = one pass over "
                         "them); newton_steps: the scale-2 loop replayed from a perturbed start "
                         "surface (a step = one Newton step; the headline of rounds 1-5, "
                         "secondary.scale2_replay of the default run); optimize: --steps whole "
                         "optimize() calls timed by the C++ loop timers (secondary.optimize)")
    ap.add_argument("--views-worker", action="store_true",
                    help="internal: one process of the multi-GPU views workload")
    ap.add_argument("--first-device", type=int, default=0, help="--views-worker")
    ap.add_argument("--num-devices", type=int, default=1, help="--views-worker")
    ap.add_argument("--views", type=int, default=8, help="--views-worker")
    ap.add_argument("--shading", action="store_true", help="--views-worker: -S")
    ap.add_argument("--sync-dir", default="", help="--views-worker")
    ap.add_argument("--worker-id", type=int, default=0, help="--views-worker")
    ap.add_argument("--no-multi-gpu-views", action="store_true",
                    help="N > 1: skip the whole-view throughput measurement over the GPUs")
    ap.add_argument("--views-in-flight", type=int, default=1,
                    help="reference views processed concurrently per GPU (own context, "
                         "stream and host thread each); the headline number uses 1")
    args = ap.parse_args()

    if args.views_worker:
        return views_worker(args)
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(respawn_under_torchrun(args))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    if args.workload == "optimize":
        return optimize_workload(args)

    shading = args.config == 5
    loops_mode = args.config == 1 and args.workload == "optimize_loops"
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline(args.small)))
        return
    if loops_mode and args.views_in_flight > 1:
        raise SystemExit("--views-in-flight > 1: use --workload newton_steps (the replayed "
                         "loops run one view per GPU; views in flight are measured as whole "
                         "views in secondary.views_per_s)")
    prob = None if loops_mode else make_problem(rank, args.small, shading=shading)

    import torch  # device plumbing + torch.distributed only
    import smvs_amd
    from smvs_amd import shard
    dist = None
    if smvs_amd.device_count() < 1:
        raise RuntimeError("bench.py needs a GPU")
    # (SMVS_DEVICE_MAP: logical -> physical devices, csrc/common.h; the library
    # maps its own device indices, torch gets the physical one.  With
    # SMVS_BENCH_DIST_BACKEND=gloo the ranks may then share a GPU -- the
    # rehearsal of the multi-rank path on a one-GPU box, tests/test_gpu_front.py;
    # RCCL itself refuses two ranks on one device)
    dev_map = [int(x) for x in os.environ.get("SMVS_DEVICE_MAP", "").split(",") if x.strip()]
    torch_index = dev_map[local_rank] if local_rank < len(dev_map) else local_rank
    backend = os.environ.get("SMVS_BENCH_DIST_BACKEND", "nccl")
    torch.cuda.set_device(torch_index)
    device = torch.device("cuda", torch_index)
    if world > 1:
        import torch.distributed as dist
        with shard.quiet_stdout():   # (RCCL's version banner)
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=device)
            else:
                dist.init_process_group(backend)
            dist.barrier()

    replay = None
    if loops_mode:
        # set-up: one optimize() of this rank's view, every Newton batch's start
        # state kept as its own resident context; the replay must reproduce the
        # optimize()'s batch log
        w, h = (480, 270) if args.small else (W, H)
        replay = LoopReplay(scene_inputs(rank, args.small), device=local_rank)
        replay.verify()
        replay_log = list(replay.log)
        surf = None
    else:
        surf = prob["surf"]
        w, h = surf["width"], surf["height"]
    n_views = max(args.views_per_rank, 1) if args.config == 5 else 1
    n_views = max(n_views, max(args.views_in_flight, 1))

    # The views of this rank: own context (device buffers + stream) each.  In
    # --config 5 they are distinct reference views in cost: same image planes,
    # individually perturbed start surfaces.
    ctxs, starts = [], []
    for v in range(0 if loops_mode else n_views):
        c = smvs_amd.ViewContext(w, h, NSUBS, device=local_rank)
        c.set_views(prob["views"])
        s = dict(surf)
        if v > 0 and args.config == 5:
            rng = np.random.default_rng(7000 + 97 * rank + v)
            nodes = surf["nodes"].copy()
            nodes[:, 0] *= 1.0 + 0.5 * NOISE * rng.standard_normal(nodes.shape[0])
            s["nodes"] = nodes
        c.set_surface(s)
        c.save_nodes()   # the start surface stays resident in HBM
        ctxs.append(c)
        starts.append(dict(prob, surf=s))
    ctx = ctxs[0] if ctxs else None

    def barrier():
        for c in ctxs:
            c.synchronize()
        if replay is not None:
            replay.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    # --shared-lighting: the one collective of the configuration, natively over
    # RCCL (include/smvs_rccl.h); torch.distributed only hands out the id
    native = shard.NativeComm(local_rank, dist) if (args.config == 5
                                                    and args.shared_lighting) else None

    def fit_lighting(round_ctxs):
        """light_optimizer.cc:22-55 for the views of one lock-step round."""
        for c in round_ctxs:
            c.light_accumulate_dev()
        if native is not None:
            native.allreduce_lighting(round_ctxs)
        return [shard.solve_lighting(*c.light_download()) for c in round_ctxs]

    def run_all(steps):
        """K steps on this rank: the default workload K passes over the recorded
        Newton loops of one optimize(); --workload newton_steps on the
        (views-in-flight) contexts concurrently, --config 5 view after view
        (lock step across ranks)."""
        if replay is not None:
            ps = its = 0
            for _ in range(steps):
                p, i, _n = replay.run_pass()
                ps += p; its += i
            return ps, its
        if args.config == 5:
            done = ps = its = 0
            v = 0
            while done < steps:
                c, p = ctxs[v % n_views], starts[v % n_views]
                c.restore_nodes()
                lighting = fit_lighting([c])[0]
                st = c.run_loop(REG, lighting=lighting,
                                max_newton_steps=min(200, steps - done), reset_active=True)
                if st["newton_steps"] == 0:
                    raise RuntimeError("Newton loop made no progress")
                done += st["newton_steps"]
                ps += st["active_patch_steps"]
                its += st["linear_iterations"]
                v += 1
            return ps, its
        import threading
        use = ctxs[:max(args.views_in_flight, 1)]
        res = [None] * len(use)

        def work(i, c):
            res[i] = run_steps(c, starts[i], steps)
        th = [threading.Thread(target=work, args=(i, c)) for i, c in enumerate(use) if i > 0]
        [t.start() for t in th]
        work(0, use[0])
        [t.join() for t in th]
        return sum(r[0] for r in res), sum(r[1] for r in res)

    run_all(args.warmup)
    repeats = []
    for _ in range(max(args.repeats, 1)):
        for c in ctxs:
            c.restore_nodes()
        barrier()
        t0 = time.perf_counter()
        patch_steps, cg_its = run_all(args.steps)
        barrier()
        elapsed = time.perf_counter() - t0
        # whole-job aggregate: units summed over ranks, time = max over ranks
        dev = device if dist is not None and backend == "nccl" else None
        units, secs = shard.aggregate_throughput(patch_steps, elapsed, dist, dev)
        its, _ = shard.aggregate_throughput(cg_its, 0.0, dist, dev)
        repeats.append((units / secs, secs, units, its))
    repeats.sort(key=lambda r: r[0])
    value, secs, units, cg_total = repeats[len(repeats) // 2]

    roof = None
    if rank == 0 and replay is not None:
        roof = loops_roofline(replay, args.steps, 1e3 * secs / args.steps,
                              with_peaks=not args.no_peaks)
    elif rank == 0:
        lighting = fit_lighting([ctx])[0] if args.config == 5 else None
        if args.config == 5:
            ctx.set_nodes(surf["nodes"])
        roof = roofline(ctx, prob, args.steps, 1e3 * secs / args.steps, lighting,
                        with_peaks=not args.no_peaks)

    # --shared-lighting: what one RCCL round of the lighting normal equations
    # costs (2,176 bytes all-reduced in place on the device buffers)
    rccl_round = None
    if native is not None:
        try:
            ctx.light_accumulate_dev()
            barrier()
            t0 = time.perf_counter()
            rounds = 50
            for _ in range(rounds):
                native.allreduce_lighting([ctx])
            torch.cuda.synchronize()
            rccl_round = dict(us_per_round=round(1e6 * (time.perf_counter() - t0) / rounds, 1),
                              rounds=rounds, world=world, bytes=272 * 8,
                              note="smvs_light_allreduce (device sum + ncclAllReduce + copy "
                                   "back) of one view's buffer per rank, host-timed")
        except Exception as e:
            rccl_round = dict(error=repr(e))

    # world > 1: what RCCL itself says about the job -- a native communicator over
    # all ranks (include/smvs_rccl.h), ncclCommCount, and one all-reduce of a
    # known pattern through the lighting buffer (rank r contributes (r + 1) x
    # pattern: the total is world (world + 1) / 2 x pattern only if every rank's
    # buffer was summed).  After the timed region: a failure here is reported,
    # never a reason to lose the headline.
    rccl_proof = None
    proof_hung = False
    # (not in the one-GPU rehearsal over gloo: RCCL refuses ranks that share a device)
    if world > 1 and backend == "nccl" and os.environ.get("SMVS_BENCH_NO_RCCL_PROOF", "") == "":
        import threading
        box = {}

        def prove():
            try:
                comm = native if native is not None else shard.NativeComm(local_rank, dist)
                n_seen, r_seen = comm.ranks()
                box.update(n_ranks_seen_by_rccl=int(n_seen), rank_seen_by_rccl=int(r_seen))
                pattern = np.arange(1.0, 273.0)
                pc = ctx if ctx is not None else smvs_amd.ViewContext(w, h, NSUBS,
                                                                      device=local_rank)
                pc.light_upload((rank + 1) * pattern[:256], (rank + 1) * pattern[256:])
                comm.allreduce_lighting([pc])
                A_sum, b_sum = pc.light_download()
                if pc is not ctx:
                    pc.close()
                total = 0.5 * world * (world + 1)
                box["allreduce_summed_every_rank"] = bool(
                    np.array_equal(A_sum.reshape(-1), total * pattern[:256])
                    and np.array_equal(b_sum, total * pattern[256:]))
                if comm is not native:
                    comm.close()
            except Exception as e:
                box["error"] = repr(e)

        # in a thread with a deadline: a communicator that never forms must not
        # take the headline with it (the measurement is finished at this point)
        worker = threading.Thread(target=prove, daemon=True)
        worker.start()
        worker.join(float(os.environ.get("SMVS_BENCH_RCCL_PROOF_TIMEOUT", "90")))
        proof_hung = worker.is_alive()
        rccl_proof = dict(box)
        if proof_hung:
            rccl_proof["error"] = "no answer from RCCL within the deadline"

    # The distributed job ends here: contexts, communicator and process group
    # go away, the other ranks exit.  What follows runs on rank 0 alone, the
    # multi-GPU part in fresh child processes (it cannot take the headline
    # with it).
    with shard.quiet_stdout():
        if not proof_hung:
            if native is not None:
                native.close()
                native = None
            for c in ctxs:
                c.close()
            if replay is not None:
                replay.close()
        ctxs = []
        if dist is not None and not proof_hung:
            # (with a deadline as well: a rank whose RCCL check failed differently
            # from the others must not wait here for ranks that have left)
            import threading

            def leave():
                dist.barrier()
                dist.destroy_process_group()
            leaver = threading.Thread(target=leave, daemon=True)
            leaver.start()
            leaver.join(120.0 if rccl_proof is not None and "error" in rccl_proof else None)
            proof_hung = leaver.is_alive()
            dist = None
    if rank != 0:
        if proof_hung:
            os._exit(0)     # (a thread is stuck inside RCCL: no orderly exit)
        return

    # The GPU phase comes first (what the driver's samplers watch at the start
    # of the run); the CPU baseline forks one oracle process per view, so it
    # runs afterwards in a child process that never loads the HIP runtime.
    secondary = cpu = None
    if world > 1 or args.config == 5:
        secondary = {}
        if rccl_round is not None:
            secondary["rccl_lighting_round"] = rccl_round
        if rccl_proof is not None:
            secondary["rccl"] = rccl_proof
        if not args.no_multi_gpu_views and not args.no_secondary:
            try:
                smvs_amd._capi.load().smvs_release_workspaces()
                time.sleep(1.0)   # (the other ranks are leaving their GPUs)
                secondary["views_per_s"] = multi_gpu_views(args, world, args.config == 5)
            except Exception as e:
                secondary["views_per_s"] = dict(error=repr(e))
        secondary["scaling_note"] = ("the headline scales by construction (independent "
                                     "views, no data-path collective); views_per_s is the "
                                     "whole per-view task whose host side can fail to scale")
    if world == 1 and args.config == 1:
        if not args.no_secondary:
            try:
                secondary = secondary_workloads(args)
            except Exception as e:   # a report, never a reason to lose the headline
                secondary = dict(error=repr(e))
            if loops_mode and isinstance(secondary, dict):
                # the headline of rounds 1-5, in a child process of its own
                try:
                    secondary["scale2_replay"] = scale2_replay(args)
                except Exception as e:
                    secondary["scale2_replay"] = dict(error=repr(e))
        if not args.no_cpu_baseline:
            cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only"]
            if args.small:
                cmd.append("--small")
            res = subprocess.run(cmd, capture_output=True, text=True)
            try:
                cpu = json.loads(res.stdout.strip().splitlines()[-1])
            except Exception:
                cpu = dict(error=(res.stderr or res.stdout)[-500:])
    if rank == 0:
        views_in_flight = max(args.views_in_flight, 1) if args.config == 1 else 1
        steps_per_unit = args.steps * world * views_in_flight
        config_extra = {}
        if loops_mode:
            ns = sum(e["newton_steps"] for e in replay_log)
            workload = ("configs[1]: %dx%d synthetic textured sphere, 1 ref + %d neighbours, -o2, "
                        "basic photometric optimizer, --no-sgm: all Newton loops of all scales "
                        "(%d .. %d) of one DepthOptimizer::optimize -- %d batches, %d Newton "
                        "steps, %d active patch-steps per pass (a step = one pass), replayed "
                        "from the batches' recorded start states; one reference view per GPU"
                        % (w, h, NSUBS, max(e["scale"] for e in replay_log), SCALE,
                           len(replay_log), ns,
                           sum(e["active_patch_steps"] for e in replay_log)))
            config_extra = dict(batches_per_step=len(replay_log), newton_steps_per_step=ns,
                                replay_verified_against_optimize_log=True)
        else:
            workload = ("configs[1]: %dx%d synthetic textured sphere, 1 ref + %d neighbours, "
                        "-o2 (scale 2, %d patches), basic photometric optimizer, %s"
                        % (w, h, NSUBS, int(surf["patch_valid"].sum()),
                           "one reference view per GPU" if views_in_flight <= 1
                           else "%d reference views in flight per GPU" % views_in_flight))
        if args.config == 5:
            workload = ("configs[4]: %d reference views x %d neighbours at %dx%d (%d per GPU), "
                        "-S shading-aware (SH lighting fit per view%s + shading residual), "
                        "scale 2, views sharded over the GPUs"
                        % (n_views * world, NSUBS, w, h, n_views,
                           ", normal equations all-reduced over the lock-step round with "
                           "RCCL" if args.shared_lighting else ""))
        out = {
            "metric": "Gauss-Newton iters/sec x active patches, 1920x1080 ref view, 8 neighbours",
            "value": value, "unit": "active-patch-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * secs / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": dict({"workload": workload, "regularization": REG,
                            "cg_iterations_per_step": cg_total / max(steps_per_unit, 1),
                            "views_in_flight_per_gpu": views_in_flight}, **config_extra),
            "timing": {"repeats": len(repeats), "statistic": "median",
                       "value_min": repeats[0][0], "value_max": repeats[-1][0],
                       "ms_per_step_min": 1e3 * repeats[-1][1] / args.steps,
                       "ms_per_step_max": 1e3 * repeats[0][1] / args.steps},
            "roofline": roof, "cpu_baseline": cpu, "secondary": secondary,
        }
        if isinstance(rccl_proof, dict) and "n_ranks_seen_by_rccl" in rccl_proof:
            # (top level: the ranks RCCL counted, not the ranks the launcher claims)
            out["n_ranks_seen_by_rccl"] = rccl_proof["n_ranks_seen_by_rccl"]
        # BASELINE.md's own timed region (all Newton loops of all scales of one
        # optimize() of the same scene) beside the headline's scale-2 replay
        if isinstance(secondary, dict) and isinstance(secondary.get("optimize"), dict) \
                and "value" in secondary["optimize"]:
            out["value_optimize"] = secondary["optimize"]["value"]
            out["value_optimize_note"] = (
                "active patch-steps / second over ALL Newton loops (scales init .. %d) of "
                "one whole DepthOptimizer::optimize of the same scene, timed in place by "
                "the C++ host's loop timers (secondary.optimize) -- BASELINE.md's timed "
                "region; %s" % (SCALE, "`value` replays exactly these loops from resident "
                                "start states inside one barrier-bracketed region"
                                if loops_mode else
                                "`value` replays the scale-%d loop of that workload from a "
                                "resident start surface" % SCALE))
        shard.flush_c_stdio()   # (whatever a C library still holds goes out first)
        print(json.dumps(out))
        sys.stdout.flush()
        if proof_hung:
            os._exit(0)


if __name__ == "__main__":
    main()
