#!/usr/bin/env python3
"""Developer probe (GPU box): per-WAVE timelines of a few k_digits workgroups (library built with -DDMZ_DG_TIMING -DDMZ_DG_TIMING2).
Per pooled column: start, conv tile-row 0, epilogue 0, conv tile-row 1, epilogue 1, ..., FC1 chunk done; cycles relative to the
workgroup's earliest wave at column 0.   usage: tools/dev/variant.sh digits.hip "-DDMZ_DG_TIMING -DDMZ_DG_TIMING2" tools/dev/dg_waves.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as entry
pkg = entry.load_package()
B = 16384
ctx = pkg.Context(0)
y = ctx.alloc(B * pkg.FRAME_BYTES); res = ctx.alloc(B * 1024); cards = ctx.alloc(B * pkg.CARD_BYTES); exp = ctx.alloc(B * pkg.EXPIRY_DTYPE.itemsize)
ctx.synth_frames(0xCA4D10, 0, B, y.ptr)
ctx.set_two_queues(False) if hasattr(ctx, "set_two_queues") else None
for _ in range(2):
    ctx.pipeline_expiry(y.ptr, B, res.ptr, exp.ptr, cards.ptr)
ctx.synchronize()
buf = (C.c_longlong * (4 * 4 * 48))()
ctx.lib.dmz_dbg_digits_waves(buf)
t = np.frombuffer(buf, np.int64).reshape(4, 4, 48)
names = ["start", "conv0", "epi0", "conv1", "epi1", "-", "-", "chunk"]
for g in range(4):
    if not t[g].any():
        continue
    t0 = t[g][:, 0].min()
    print("workgroup %d:" % (2049 + 4096 * g))
    for w in range(4):
        row = []
        for pc in range(5):
            row.append("c%d[" % pc + " ".join("%s %d" % (names[i], t[g][w][8 * pc + i] - t0) for i in (0, 1, 2, 3, 4, 7)) + "]")
        print("  wave %d: " % w + " ".join(row) + "  last column's barriers %d %d" % (t[g][w][40] - t0, t[g][w][41] - t0))
