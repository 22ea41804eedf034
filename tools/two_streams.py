#!/usr/bin/env python3
"""Developer experiment (GPU box): one context on the whole batch against K contexts (= K HIP streams)
on 1/K of the batch each, run from K host threads -- do the MFMA/latency-bound kernels of one part
overlap the VALU-bound kernels of another?   usage: python tools/two_streams.py [batch] [reps] [K]"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry

pkg = entry.load_package()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
K = int(sys.argv[3]) if len(sys.argv) > 3 else 2


def setup(n):
    ctx = pkg.Context(0)
    bufs = (ctx.alloc(n * pkg.FRAME_BYTES), ctx.alloc(n * 1024), ctx.alloc(n * pkg.CARD_BYTES),
            ctx.alloc(n * pkg.EXPIRY_DTYPE.itemsize))
    ctx.synth_frames(0xCA4D10, 0, n, bufs[0].ptr)
    run(ctx, bufs, n, 1)
    return ctx, bufs


def run(ctx, bufs, n, r):
    y, res, cards, exp = bufs
    for _ in range(r):
        ctx.pipeline_expiry(y.ptr, n, res.ptr, exp.ptr, cards.ptr)
    ctx.synchronize()


one = setup(B)
t0 = time.perf_counter()
run(one[0], one[1], B, reps)
dt1 = (time.perf_counter() - t0) / reps
parts = [setup(B // K) for _ in range(K)]
t0 = time.perf_counter()
th = [threading.Thread(target=run, args=(c, b, B // K, reps)) for c, b in parts]
for t in th:
    t.start()
for t in th:
    t.join()
dtk = (time.perf_counter() - t0) / reps
print("B=%d one stream %.3f ms (%.0f frames/s)   %d streams %.3f ms (%.0f frames/s)  ratio %.3f"
      % (B, dt1 * 1e3, B / dt1, K, dtk * 1e3, B / dtk, dt1 / dtk))
