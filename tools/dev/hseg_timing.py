#!/usr/bin/env python3
"""Developer probe (GPU box): phase timeline of one k_hseg wave (library built with -DDMZ_HSEG_TIMING).
usage: tools/dev/variant.sh hseg.hip -DDMZ_HSEG_TIMING tools/dev/hseg_timing.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as entry

pkg = entry.load_package()
B = int(os.environ.get("BATCH", "16384"))
ctx = pkg.Context(0)
y = ctx.alloc(B * pkg.FRAME_BYTES)
res = ctx.alloc(B * 1024)
cards = ctx.alloc(B * pkg.CARD_BYTES)
exp = ctx.alloc(B * pkg.EXPIRY_DTYPE.itemsize)
ctx.synth_frames(0xCA4D10, 0, B, y.ptr)
for _ in range(3):
    ctx.pipeline_expiry(y.ptr, B, res.ptr, exp.ptr, cards.ptr)
    ctx.synchronize()
