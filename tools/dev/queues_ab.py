#!/usr/bin/env python3
"""Developer probe (GPU box): the timed pipeline step on one queue against three (dmz_hip_set_two_queues), same box.
usage: tools/dev/queues_ab.py [batch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as entry
pkg = entry.load_package()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
ctx = pkg.Context(0)
y = ctx.alloc(B * pkg.FRAME_BYTES); res = ctx.alloc(B * 1024); exp = ctx.alloc(B * pkg.EXPIRY_DTYPE.itemsize)
cards = ctx.alloc(B * pkg.CARD_BYTES)
ctx.synth_frames(0xCA4D10, 0, B, y.ptr)
for rep in range(3):
    for name, two in (("three queues", True), ("one queue", False)):
        ctx.set_two_queues(two)
        ctx.pipeline_expiry(y.ptr, B, res.ptr, exp.ptr, cards.ptr); ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            ctx.pipeline_expiry(y.ptr, B, res.ptr, exp.ptr, cards.ptr)
        ctx.synchronize()
        print("%-13s %.3f ms per step" % (name, (time.perf_counter() - t0) / 3 * 1e3))
