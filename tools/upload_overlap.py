#!/usr/bin/env python3
"""Prologue of a view on the device: nine 1920x1080 RGB images from page-locked
memory + the first scale's planes, synchronous uploads (smvs_ctx_upload_image:
DMA, conversion, wait -- nine times) against smvs_ctx_upload_image_async (DMA
on the copy stream, conversion where set_scale needs it).  Host wall time until
the context is synchronised, best of 7."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import smvs_amd
from smvs_amd import _capi
lib = _capi.load()
w, h, n = 1920, 1080, 8
rng = np.random.default_rng(0)
img = rng.integers(0, 256, size=(h, w, 3)).astype(np.uint8)
ptrs = []
for v in range(n + 1):
    p = C.c_void_p(); _capi.check(lib.smvs_pinned_alloc(C.c_size_t(img.size), C.byref(p)))
    C.memmove(p, img.ctypes.data, img.size); ptrs.append(p)
ctx = smvs_amd.ViewContext(w, h, n)
u8p = C.POINTER(C.c_uint8)
def run(fn):
    best = 1e9
    for rep in range(7):
        ctx.synchronize()
        t = time.perf_counter()
        for v in range(n + 1):
            _capi.check(fn(ctx.handle, v - 1, w, h, 3, C.cast(ptrs[v], u8p)))
        t_up = time.perf_counter()
        ctx.set_scale(6)
        ctx.synchronize()
        t1 = time.perf_counter()
        best = min(best, t1 - t)
        last = (t_up - t, t1 - t_up)
    return best, last
for name, fn in (("synchronous", lib.smvs_ctx_upload_image), ("asynchronous", lib.smvs_ctx_upload_image_async),
                 ("synchronous", lib.smvs_ctx_upload_image), ("asynchronous", lib.smvs_ctx_upload_image_async)):
    best, last = run(fn)
    print("%-12s uploads + set_scale(6): best %.3f ms (last: upload calls %.3f ms, set_scale + wait %.3f ms)%s"
          % (name, 1e3 * best, 1e3 * last[0], 1e3 * last[1],
             "  [SMVS_UPLOAD_STREAM=%s]" % os.environ["SMVS_UPLOAD_STREAM"] if os.environ.get("SMVS_UPLOAD_STREAM") else ""))
ctx.close()
