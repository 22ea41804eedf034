#!/usr/bin/env python3
"""Developer probe (GPU box): the expiry stages on the bench's synthetic frames with a library built with -DDMZ_XC_TIMING
(prints the phase timeline of one categorised workgroup).
usage: tools/dev/variant.sh expiry.hip -DDMZ_XC_TIMING tools/dev/expiry_timing.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as entry
pkg = entry.load_package()
B = int(os.environ.get("BATCH", "8192"))
ctx = pkg.Context(0)
y = ctx.alloc(B * pkg.FRAME_BYTES)
res = ctx.alloc(B * 1024)
exp = ctx.alloc(B * pkg.EXPIRY_DTYPE.itemsize)
ctx.synth_frames(0xCA4D10, 0, B, y.ptr)
for _ in range(3):
    ctx.pipeline_expiry(y.ptr, B, res.ptr, exp.ptr)
    ctx.synchronize()
