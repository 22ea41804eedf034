#!/usr/bin/env python3
"""Developer tool (GPU box): achieved HBM bandwidth of the plumbing kernels."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry

pkg = entry.load_package()
ctx = pkg.Context(0)


def timeit(fn, reps=10):
    fn()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.synchronize()
    return (time.perf_counter() - t0) / reps


n = 8192
px = n * 240 * 320
src, c1, c2 = ctx.alloc(px * 2), ctx.alloc(px), ctx.alloc(px)
t = timeit(lambda: ctx.deinterleave_c2(src.ptr, px, c1.ptr, c2.ptr))
print("deinterleave_c2   %d half-size CbCr planes: %.3f ms  %.0f GB/s (read+write %.2f GB)" % (n, t * 1e3, px * 4 / t / 1e9, px * 4 / 1e9))
for b in (src, c1, c2):
    b.free()
px = n * 270 * 428
y, cb, cr, rgb = ctx.alloc(px), ctx.alloc(px), ctx.alloc(px), ctx.alloc(px * 4)
for ch in (3, 4):
    t = timeit(lambda: ctx.ycbcr_to_rgb(y.ptr, cb.ptr, cr.ptr, px, rgb.ptr, channels=ch))
    print("ycbcr_to_rgb(%d)   %d cards: %.3f ms  %.0f GB/s (read+write %.2f GB)" % (ch, n, t * 1e3, px * (3 + ch) / t / 1e9, px * (3 + ch) / 1e9))
px = n * 640 * 480 // 4
t = timeit(lambda: ctx.deinterleave_rgba_to_r(rgb.ptr, y.ptr, (270 * 428 * n) // 4 * 4 // 4 * 4 // 4))
sz = (270 * 428 * n) // 4 * 4 // 4 * 4 // 4
print("rgba_to_r         %d px: %.3f ms  %.0f GB/s" % (sz, t * 1e3, sz * 5 / t / 1e9))
