#!/usr/bin/env python3
"""Developer tool (GPU box): wall-clock ms of dmz_hip_pipeline_expiry_batch on a resident batch.
usage: python tools/time_pipeline.py [batch] [reps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry

pkg = entry.load_package()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ctx = pkg.Context(0)
y = ctx.alloc(B * pkg.FRAME_BYTES)
res = ctx.alloc(B * 1024)
cards = ctx.alloc(B * pkg.CARD_BYTES)
exp = ctx.alloc(B * pkg.EXPIRY_DTYPE.itemsize)
ctx.synth_frames(0xCA4D10, 0, B, y.ptr)
ctx.pipeline_expiry(y.ptr, B, res.ptr, exp.ptr, cards.ptr)
ctx.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    ctx.pipeline_expiry(y.ptr, B, res.ptr, exp.ptr, cards.ptr)
ctx.synchronize()
dt = (time.perf_counter() - t0) / reps
print("B=%d %.3f ms/step %.0f frames/s" % (B, dt * 1e3, B / dt))
