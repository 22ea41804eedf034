#!/usr/bin/env python3
"""Developer probe (GPU box): phase timeline of one k_digits workgroup (library built with -DDMZ_DG_TIMING)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as entry
pkg = entry.load_package()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
ctx = pkg.Context(0)
y = ctx.alloc(B * pkg.FRAME_BYTES); res = ctx.alloc(B * 1024); cards = ctx.alloc(B * pkg.CARD_BYTES)
exp = ctx.alloc(B * pkg.EXPIRY_DTYPE.itemsize)
ctx.synth_frames(0xCA4D10, 0, B, y.ptr)
for _ in range(2):
    ctx.pipeline_expiry(y.ptr, B, res.ptr, exp.ptr, cards.ptr)
ctx.synchronize()
r = res.download(pkg.RESULT_DTYPE)
print("cycles since entry:", r[B // 2]["scores"].reshape(-1)[1:16].astype(int).tolist())
