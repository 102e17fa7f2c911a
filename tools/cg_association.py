#!/usr/bin/env python3
"""How the iteration count of ConjugateGradient::solve (lib/conjugate_gradient.h:72-202)
on the ill-conditioned systems of the fuzz sweep depends on the ASSOCIATION of its dot
products -- CPU only, numpy + the oracle's SpMV.

The recurrence below is the reference's, operation for operation; only `dot` changes:

  seq     the oracle's own orc_vec_dot (sse_vector.cc: pairs summed sequentially)
  sumx    the products rounded to double as in the reference, their sum exact
          (math.fsum) -- what csrc/cg.hip computes since round 5 (TwoSum accumulators)
  pairx   adjacent products added first (the reference's `a0*b0 + a1*b1`), then exact
  exact   Dot2: every product exactly (Dekker / Veltkamp split), the sum by math.fsum --
          the correctly rounded dot product
  pair    numpy's pairwise sum of the rounded products
  blkN    N accumulators strided over the elements, then a binary tree: what a plain
          sum over N-thread blocks does (csrc/cg.hip before round 5 was blk512-like)

Result (profiles/r5_cg_association.txt): on these systems the count is sensitive to a
1-ulp change of a dot product -- EVERY association other than the reference's own ends
some solve 2 .. 7 iterations apart (outlier 6: 61 instead of 68 with blocked sums, the
very number the streaming solver produced on the GPU in round 4; outlier 0: 107 instead
of 101 with the correctly rounded dot product).  The exact sum of the rounded products
(sumx) is the one that stays within ONE iteration on all seven (six identical) ON THE
ORACLE'S OWN SYSTEM.  Part 1 of the output shows why that does not carry over to the
device: the oracle's own solve flips between the same two exits (68 / 61, 50 / 54, ...)
when g moves by 1e-12 relative, which is how far the device's construction is from the
oracle's.  The bound of tests/test_gpu_parity.py (FUZZ_OUTLIERS) is that spread.
"""
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smvs_amd import synth  # noqa: E402
from oracle import pyoracle as oracle  # noqa: E402

FUZZ_OUTLIERS = [
    dict(w=38, h=49, scale=1, n_subs=4, noise=0.03, seed=8446),
    dict(w=66, h=29, scale=2, n_subs=4, noise=0.03, seed=4778),
    dict(w=33, h=31, scale=1, n_subs=7, noise=0.01, seed=2627),
    dict(w=45, h=31, scale=1, n_subs=8, noise=0.01, seed=1440),
    dict(w=57, h=48, scale=1, n_subs=6, noise=0.03, seed=2853),
    dict(w=43, h=26, scale=1, n_subs=8, noise=0.01, seed=7132, light_reg=0.0),
    dict(w=53, h=33, scale=1, n_subs=4, noise=0.03, seed=4110),
]


def dot_seq(a, b):
    return oracle.vec_dot(a, b)


def dot_pair(a, b):
    return float(np.sum(a * b))


def two_prod(a, b):
    """p = fl(a * b) and e with p + e = a * b exactly (Dekker, Veltkamp split)."""
    p = a * b
    c = 134217729.0                      # 2^27 + 1
    ah = c * a; ah = ah - (ah - a); al = a - ah
    bh = c * b; bh = bh - (bh - b); bl = b - bh
    e = ((ah * bh - p) + ah * bl + al * bh) + al * bl
    return p, e


def dot_sumx(a, b):
    return math.fsum((a * b).tolist())


def dot_pairx(a, b):
    p = a * b
    return math.fsum((p[0::2] + p[1::2]).tolist())


def dot_exact(a, b):
    p, e = two_prod(a, b)
    return math.fsum(np.concatenate([p, e]).tolist())


def make_blocked(nb):
    def dot_blk(a, b):
        p = a * b
        acc = np.zeros(nb)
        for i in range(0, p.size, nb):
            seg = p[i:i + nb]
            acc[:seg.size] += seg
        while acc.size > 1:
            h = acc.size // 2
            acc = acc[:h] + acc[h:2 * h]
        return float(acc[0])
    return dot_blk


def pcg(orc, H9, present, P, b, dot, max_it=200, q_tol=1e-3):
    """conjugate_gradient.h:86-199 with the block-Jacobi preconditioner."""
    nn = b.size // 4
    Pm = P.reshape(nn, 4, 4)
    act = present.reshape(nn, 9)[:, 4].astype(bool)

    def prec(r):
        rr = r.reshape(nn, 4)
        zz = np.zeros((nn, 4))
        for bc in range(4):              # block_sparse_matrix.h:289-295: column by column
            zz += Pm[:, :, bc] * rr[:, bc:bc + 1]
        zz[~act] = 0
        return zz.reshape(-1)

    x = np.zeros(b.size); r = b.copy(); z = prec(r)
    rdr = dot(z, r); d = z.copy()
    q0 = -1.0 * dot(x, b + r)
    tol = 0.01 * math.sqrt(dot(b, b))
    it = 1
    while it < max_it:
        Ad = orc.spmv(H9, present, d)
        alpha = rdr / dot(d, Ad)
        x = x + alpha * d
        r = r - alpha * Ad
        if dot(r, r) < tol:
            break
        q1 = -1.0 * dot(x, b + r)
        if it * (q1 - q0) / q1 < q_tol:
            break
        q0 = q1
        z = prec(r)
        nrdr = dot(z, r)
        d = z + (nrdr / rdr) * d
        rdr = nrdr
        it += 1
    return it, x


def first_solve(case, dots):
    c = FUZZ_OUTLIERS[case]
    prob = synth.make_problem(c["w"], c["h"], c["n_subs"], c["scale"], shading=True,
                              noise=c["noise"], seed=c["seed"])
    surf, lighting, light_reg = prob["surf"], prob["lighting"], c.get("light_reg", 0.5)
    orc = oracle.OracleProblem(surf, prob["views"])
    ref = orc.gn_construct(surf["node_valid"].copy(), 0.01, light_reg, lighting)
    b = -ref["g"].reshape(-1)
    _, itr, _ = orc.cg_solve(ref["H9"], ref["present"], ref["P"], -ref["g"], 200,
                             0.01 * np.linalg.norm(ref["g"]), 1e-3)
    res = {name: pcg(orc, ref["H9"], ref["present"], ref["P"], b, dot)[0] for name, dot in dots}
    return b.size // 4, itr, res


DOTS = (("seq", dot_seq), ("sumx", dot_sumx), ("pairx", dot_pairx), ("exact", dot_exact),
        ("pair", dot_pair), ("blk64", make_blocked(64)),
        ("blk256", make_blocked(256)), ("blk512", make_blocked(512)))

def input_noise(case, trials=12, rel=1e-12, seed=1):
    """The oracle's own C solve of the outlier's first system with g perturbed by
    `rel` (relative, Gaussian) -- the level at which the device's g agrees with the
    oracle's.  Returns the iteration counts (the first one unperturbed)."""
    c = FUZZ_OUTLIERS[case]
    prob = synth.make_problem(c["w"], c["h"], c["n_subs"], c["scale"], shading=True,
                              noise=c["noise"], seed=c["seed"])
    surf, lighting, light_reg = prob["surf"], prob["lighting"], c.get("light_reg", 0.5)
    orc = oracle.OracleProblem(surf, prob["views"])
    ref = orc.gn_construct(surf["node_valid"].copy(), 0.01, light_reg, lighting)
    rng = np.random.default_rng(seed)
    counts = []
    for k in range(trials):
        g = ref["g"] * (1.0 + (0.0 if k == 0 else rel) * rng.standard_normal(ref["g"].shape))
        _, it, _ = orc.cg_solve(ref["H9"], ref["present"], ref["P"], -g, 200,
                                0.01 * np.linalg.norm(g), 1e-3)
        counts.append(it)
    return counts


if __name__ == "__main__":
    print("# tools/cg_association.py, part 1: the ORACLE's own solve (C, sequential sums) of the first "
          "system of each fuzz outlier with g perturbed by 1e-12 relative -- the count is not a "
          "function of the system alone")
    for ci in range(len(FUZZ_OUTLIERS)):
        print("outlier %d: %s" % (ci, " ".join("%3d" % c for c in input_noise(ci))), flush=True)
    print("# part 2: the reference's recurrence with its dot products summed in different associations")
    for ci in range(len(FUZZ_OUTLIERS)):
        nodes, itr, res = first_solve(ci, DOTS)
        print("outlier %d: %4d nodes, oracle (C) %3d | %s" % (ci, nodes, itr, "  ".join(
            "%s %3d" % kv for kv in res.items())), flush=True)
