#!/usr/bin/env python3
"""Developer tool (GPU box): latency of small batches through the C-ABI (device-resident and host buffers)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry

pkg = entry.load_package()
ctx = pkg.Context(0)
for B in (1, 8, 64, 512):
    y = ctx.alloc(B * pkg.FRAME_BYTES)
    res = ctx.alloc(B * 1024)
    exp = ctx.alloc(B * pkg.EXPIRY_DTYPE.itemsize)
    cards = ctx.alloc(B * pkg.CARD_BYTES)
    ctx.synth_frames(1, 0, B, y.ptr)
    hy = y.download(np.uint8)
    hres = np.zeros(B, pkg.RESULT_DTYPE)
    hexp = np.zeros(B, pkg.EXPIRY_DTYPE)
    for name, fn in (("device buffers", lambda: (ctx.pipeline_expiry(y.ptr, B, res.ptr, exp.ptr, cards.ptr), ctx.synchronize())),
                     ("host buffers  ", lambda: ctx.pipeline_expiry(hy, B, hres, hexp))):
        for _ in range(3):
            fn()
        t0 = time.perf_counter()
        reps = 50
        for _ in range(reps):
            fn()
        dt = (time.perf_counter() - t0) / reps
        print("B=%4d %s %8.1f us/call  %8.1f us/frame" % (B, name, dt * 1e6, dt * 1e6 / B))
