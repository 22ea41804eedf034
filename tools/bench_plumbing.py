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

# ---- ranks 2 and 4: sessions, quality scores, blur ----
import numpy as np

B = 65536
yb = ctx.alloc(B * pkg.FRAME_BYTES)
res = ctx.alloc(B * 1024)
exp = ctx.alloc(B * pkg.EXPIRY_DTYPE.itemsize)
cards = ctx.alloc(B * pkg.CARD_BYTES)
ctx.synth_frames(0xCA4D10, 0, B, yb.ptr)
ctx.pipeline_expiry(yb.ptr, B, res.ptr, exp.ptr, cards.ptr)
ctx.synchronize()
for F in (8, 32):
    S = B // F
    out = ctx.alloc(S * 128)
    t = timeit(lambda: ctx.scan_sessions(res.ptr, exp.ptr, S, F, out.ptr, scan_expiry=True, frame_interval_ms=33))
    print("scan_sessions     %d sessions x %d frames: %.3f ms  %.1f M frames/s  %.0f GB/s of records read"
          % (S, F, t * 1e3, B / t / 1e6, B * 2616 / t / 1e9))
    out.free()
focus, bright = ctx.alloc(B * 4), ctx.alloc(B * 4)
for full in (False, True):
    t = timeit(lambda: ctx.scores(yb.ptr, B, focus.ptr, bright.ptr, use_full_image=full))
    roi = 428 * 270 if full else 142 * 90
    print("scores(full=%d)    %d frames: %.3f ms  %.1f M frames/s  %.0f GB/s of ROI pixels" % (full, B, t * 1e3, B / t / 1e6, B * roi / t / 1e9))
n = 8192
rgb = ctx.alloc(n * pkg.CARD_BYTES * 3)
sess = np.zeros(n, pkg.SESSION_DTYPE)
sess["n_offsets"] = 16
sess["offsets"] = (40 + 19 * np.arange(16)).astype(np.uint16)
sess["number_width"] = 18.0
sess["vseg_y_offset"] = 150
dsess = ctx.alloc(sess.nbytes)
dsess.upload(sess.view(np.uint8))
t = timeit(lambda: ctx.blur_cards(rgb.ptr, n, dsess.ptr, 4), reps=3)
print("blur_cards        %d cards (12 boxes each): %.3f ms  %.2f M cards/s" % (n, t * 1e3, n / t / 1e6))
