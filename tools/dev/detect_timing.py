#!/usr/bin/env python3
"""Developer probe (GPU box): detect on the bench's synthetic card frames with a library built with -DDMZ_DT_TIMING
(prints the phase timeline of one workgroup per kernel).  usage: tools/dev/variant.sh detect.hip -DDMZ_DT_TIMING tools/dev/detect_timing.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as entry
pkg = entry.load_package()
B = int(os.environ.get("BATCH", "8192"))
ctx = pkg.Context(0)
y = ctx.alloc(B * pkg.FRAME_BYTES)
res = ctx.alloc(B * 1024)
ctx.synth_frames(0xCA4D10, 0, B, y.ptr)
for _ in range(3):
    ctx.detect(y.ptr, B, res.ptr)
    ctx.synchronize()
