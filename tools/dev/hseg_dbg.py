#!/usr/bin/env python3
"""Developer tool (GPU box): how many passes of the filtered hseg search ran in the ordered form, from a library built with
-DDMZ_HSEG_DBG.   usage: tools/dev/variant.sh hseg.hip -DDMZ_HSEG_DBG tools/dev/hseg_dbg.py"""
import ctypes as C
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
out = (C.c_ulonglong * 4)()
ctx.lib.dmz_dbg_hseg(out, 1)
ctx.synth_frames(0xCA4D10, 0, B, y.ptr)
ctx.pipeline_expiry(y.ptr, B, res.ptr, exp.ptr, cards.ptr)
ctx.synchronize()
ctx.lib.dmz_dbg_hseg(out, 0)
print("passes %d, in the ordered form %d (%.2f %%)" % (out[0], out[1], 100.0 * out[1] / max(1, out[0])))
