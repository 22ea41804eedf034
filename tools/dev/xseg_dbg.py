#!/usr/bin/env python3
"""Developer tool (GPU box): cycle counts of the candidate pick of k_expiry_seg from a library built with -DDMZ_XSEG_DBG
(variants/xsdbg.so): stripes, stripes that took the library-order path, cycles of the watched rounds / the re-ordering /
the repeated rounds."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as entry

pkg = entry.load_package()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
ctx = pkg.Context(0)
y = ctx.alloc(B * pkg.FRAME_BYTES)
res = ctx.alloc(B * 1024)
cards = ctx.alloc(B * pkg.CARD_BYTES)
exp = ctx.alloc(B * pkg.EXPIRY_DTYPE.itemsize)
ctx.synth_frames(0xCA4D10, 0, B, y.ptr)
ctx.pipeline_expiry(y.ptr, B, res.ptr, exp.ptr, cards.ptr)
ctx.synchronize()
out = (C.c_ulonglong * 16)()
ctx.lib.dmz_dbg_xseg(out, 1)
ctx.pipeline_expiry(y.ptr, B, res.ptr, exp.ptr, cards.ptr)
ctx.synchronize()
ctx.lib.dmz_dbg_xseg(out, 0)
n, t = out[0], max(1, out[1])
print("stripes %d, re-ordered %d (%.1f %%); cycles per stripe: watched rounds %.0f; per re-ordered stripe: ordering + repeated rounds %.0f, "
      "levels %.2f, ended on the all-marks pass %.1f %%" % (n, out[1], 100.0 * out[1] / max(1, n), out[2] / max(1, n), out[3] / t, out[4] / t, 100.0 * out[7] / t))
print("partitions per re-ordered stripe %.1f, slots (64 positions) %.1f" % (out[8] / t, out[9] / t))
print("stripes that ran to the end: %d, cycles per such stripe (whole kernel): %.0f" % (out[6], out[5] / max(1, out[6])))
