#!/usr/bin/env python3
"""Whole-view throughput on one GPU: T host threads, each running the C++
DepthOptimizer::optimize (own context + stream) on the same synthetic
1920x1080 / 8-neighbour scene.  ctypes releases the GIL during the calls."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from smvs_amd import synth, host

w = int(sys.argv[1]) if len(sys.argv) > 1 else 1920
h = int(sys.argv[2]) if len(sys.argv) > 2 else 1080
n = int(sys.argv[3]) if len(sys.argv) > 3 else 8
inp = synth.pipeline_inputs("sphere", w, h, n, flen=1.2)
host.optimize(inp, min_scale=2)  # warm-up (library load, first allocations)
for threads in (1, 2, 4, 8):
    views = 4 * threads
    todo = list(range(views))
    lock = threading.Lock()
    def work():
        while True:
            with lock:
                if not todo:
                    return
                todo.pop()
            host.optimize(inp, min_scale=2)
    t = time.perf_counter()
    th = [threading.Thread(target=work) for _ in range(threads)]
    [x.start() for x in th]; [x.join() for x in th]
    dt = time.perf_counter() - t
    print("threads %d: %d views in %.2f s = %.2f views/s" % (threads, views, dt, views / dt))
