#!/usr/bin/env python3
"""Developer tool (GPU box): throughput and call time of the full pipeline (detect -> ... -> expiry) against the batch size,
device-resident buffers, three queues: where the 65 536-frame headline sits on the curve, and what a small batch costs.
usage: python tools/dev/batch_curve.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as entry

pkg = entry.load_package()
ctx = pkg.Context(0)
BMAX = 131072
y = ctx.alloc(BMAX * pkg.FRAME_BYTES)
res = ctx.alloc(BMAX * 1024)
cards = ctx.alloc(BMAX * pkg.CARD_BYTES)
exp = ctx.alloc(BMAX * pkg.EXPIRY_DTYPE.itemsize)
ctx.synth_frames(0xCA4D10, 0, BMAX, y.ptr)
print("%8s %12s %14s" % ("batch", "ms per call", "frames/s"))
for B in (1, 8, 64, 256, 1024, 4096, 16384, 32768, 65536, 131072):
    reps = max(4, min(200, int(2e5 // max(B, 256))))
    for _ in range(2):
        ctx.pipeline_expiry(y.ptr, B, res.ptr, exp.ptr, cards.ptr)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.pipeline_expiry(y.ptr, B, res.ptr, exp.ptr, cards.ptr)
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print("%8d %12.3f %14.0f" % (B, dt * 1e3, B / dt))
