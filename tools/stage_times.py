#!/usr/bin/env python3
"""Developer tool: per-stage hipEvent timings of the pipeline on one GPU.
usage: python tools/stage_times.py [batch] [reps]   (DMZ_HIP_LIB selects a library variant)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry

pkg = entry.load_package()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ctx = pkg.Context(0)
y = ctx.alloc(B * pkg.FRAME_BYTES)
res = ctx.alloc(B * 1024)
cards = ctx.alloc(B * pkg.CARD_BYTES)
exp = ctx.alloc(B * pkg.EXPIRY_DTYPE.itemsize)
ctx.synth_frames(0xCA4D10, 0, B, y.ptr)
ctx.pipeline_expiry(y.ptr, B, res.ptr, exp.ptr, cards.ptr)
ctx.set_profiling(True)
ctx.stage_times()
for _ in range(reps):
    ctx.pipeline_expiry(y.ptr, B, res.ptr, exp.ptr, cards.ptr)
t = ctx.stage_times()
tot = 0.0
out = []
for k, (ms, cnt) in t.items():
    out.append("%s %.3f" % (k, ms / reps))
    tot += ms / reps
print(os.environ.get("DMZ_HIP_LIB", "default"), "B=%d" % B, " ".join(out), "total %.3f ms  %.0f frames/s" % (tot, B / tot * 1e3))
