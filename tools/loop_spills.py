#!/usr/bin/env python3
"""Scratch traffic, barriers and cross-lane operations inside the loops of the resident PCG
kernels (hipcc -S of cg_resident.hip): spill reloads in the solve loop cost a memory round
trip per iteration.  Usage: loop_spills.py [extra hipcc flags]"""
import os, subprocess, sys, tempfile
here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(here, "smvs_amd", "csrc", "cg_resident.hip")
out = os.path.join(tempfile.gettempdir(), "cg_resident.s")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S",
                       "--cuda-device-only", src, "-o", out] + sys.argv[1:], stderr=subprocess.DEVNULL)
s = open(out).read()
for name in ("ILb1ELb1ELb0E", "ILb1ELb1ELb1E", "ILb1ELb0ELb0E"):
    full = "_ZN8smvs_hip18cg_resident_kernel%sEEvNS_7ResArgsE" % name
    a = s.index(full + ":")
    body = s[a:s.index(".Lfunc_end", a)].split("\n")
    loops, cur = {}, None
    for i, l in enumerate(body):
        if "Loop Header: Depth=1" in l:
            cur = l.split(":")[0].strip()
            loops[cur] = [i, i]
        elif cur and "Header=" + cur[2:] in l:
            loops[cur][1] = i
    print(name, "(%d lines)" % len(body))
    for k, (b, e) in loops.items():
        seg = body[b:e]
        if sum("s_barrier" in x for x in seg) == 0:
            continue
        count = lambda w: sum(w in x for x in seg)
        print("  loop %s: %d lines, barriers %d, scratch_load %d, scratch_store %d, ds_bpermute %d, "
              "permlane swaps %d, dpp %d, buffer_store %d, waits on vmcnt %d"
              % (k, e - b, count("s_barrier"), count("scratch_load"), count("scratch_store"),
                 count("ds_bpermute"), count("permlane"), count("_dpp"), count("buffer_store"), count("vmcnt")))
