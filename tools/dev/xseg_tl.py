#!/usr/bin/env python3
"""Developer probe (GPU box): cycles per phase of k_expiry_seg, averaged over the waves that reach the end (-DDMZ_XSEG_TL).
usage: tools/dev/variant.sh expiry.hip -DDMZ_XSEG_TL tools/dev/xseg_tl.py"""
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
ctx.synth_frames(0xCA4D10, 0, B, y.ptr)
ctx.pipeline_expiry(y.ptr, B, res.ptr, exp.ptr, cards.ptr)
ctx.synchronize()
out = (C.c_ulonglong * 16)()
ctx.lib.dmz_dbg_xseg_tl(out, 1)
ctx.pipeline_expiry(y.ptr, B, res.ptr, exp.ptr, cards.ptr)
ctx.synchronize()
ctx.lib.dmz_dbg_xseg_tl(out, 0)
n = max(1, out[15])
names = ["rows", "column sums", "thresholds + rect sums", "pick", "groups", "regrid", "character rects", "between", "slash + emit"]
print("per wave: groups %.2f (with a character image <= 16 wide: %.2f), rects %.2f, batches of three %.2f, of four %.2f, batches saved by four where the image fits %.2f"
      % (out[11] / n, out[12] / n, out[13] / n, out[9] / n, out[10] / n, out[14] / n))
print("waves %d; cycles per wave: " % n + ", ".join("%s %.0f" % (names[i], out[i] / n) for i in range(9)) + "; total %.0f" % (sum(out[:9]) / n))
