#!/usr/bin/env python3
"""Developer probe driver (GPU box): the number scan alone (vseg -> hseg -> digits, one queue) on the bench's synthetic cards, for
libraries built with a probe macro.   usage: tools/dev/variant.sh digits.hip -DDMZ_DG_LOADWAIT tools/dev/scan_only.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as entry

pkg = entry.load_package()
B = int(os.environ.get("BATCH", "8192"))
ctx = pkg.Context(0)
y = ctx.alloc(B * pkg.FRAME_BYTES)
res = ctx.alloc(B * 1024)
cards = ctx.alloc(B * pkg.CARD_BYTES)
ctx.synth_frames(0xCA4D10, 0, B, y.ptr)
ctx.pipeline(y.ptr, B, res.ptr, cards.ptr)
ctx.synchronize()
for _ in range(int(os.environ.get("REPS", "1"))):
    ctx.scan_cards(cards.ptr, B, res.ptr)
    ctx.synchronize()
