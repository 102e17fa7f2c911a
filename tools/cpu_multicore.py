#!/usr/bin/env python3
"""CPU baseline the way the reference parallelises (one reference view per
core, app/smvsrecon.cc:49,558): C processes each run one full Newton step of
the bench workload with the oracle; throughput = C * patches / wall."""
import os, sys, time, multiprocessing as mp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench

prob = None

def work(_):
    t = time.perf_counter()
    r = bench.cpu_baseline(prob)
    return time.perf_counter() - t, r["value"]

if __name__ == "__main__":
    prob = bench.make_problem(0, False)
    for c in [int(a) for a in sys.argv[1:]] or [1, 32]:
        t = time.perf_counter()
        with mp.get_context("fork").Pool(c) as pool:
            res = pool.map(work, range(c))
        wall = time.perf_counter() - t
        patches = int(prob["surf"]["patch_valid"].sum())
        print("%3d cores: wall %.1f s, %.0f active-patch-steps/s (per-core mean %.0f)"
              % (c, wall, c * patches / wall, np.mean([v for _, v in res])))
