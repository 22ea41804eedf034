#!/usr/bin/env python3
"""Developer probe (GPU box): one frame of the corpus through the device and the oracle, the hseg fields side by side.
usage: r6_hseg_case.py <seed> <corpus index> [flavour]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as entry
pkg = entry.load_package(); orc = entry.load_oracle(); o = orc.Oracle()
seed, idx = int(sys.argv[1], 0), int(sys.argv[2])
flav = int(sys.argv[3]) if len(sys.argv) > 3 else 0
ctx = pkg.Context(0)
n = 64
y = ctx.alloc(n * pkg.FRAME_BYTES); res = ctx.alloc(n * 1024); cards = ctx.alloc(n * pkg.CARD_BYTES)
ctx.synth_frames(seed, idx - 8, n, y.ptr)
ctx.set_reference_flavour(flav); o.set_reference_flavour(flav)
ctx.pipeline(y.ptr, n, res.ptr, cards.ptr); ctx.synchronize()
got = res.download(pkg.RESULT_DTYPE, n)
frames = y.download(np.uint8).reshape(n, 480, 640)
gc = cards.download(np.uint8).reshape(n, 270, 428)
for j in range(n):
    w, wcard = o.scan_frame(frames[j])
    g = got[j]
    same = np.array_equal(g["offsets"], w["offsets"]) and g["hseg_score"].view(np.uint32) == w["hseg_score"].view(np.uint32) and g["pattern_offset"] == w["pattern_offset"]
    if not same or j == 8:
        print("frame %d (%s): cards equal %s" % (idx - 8 + j, "SAME" if same else "DIFFERENT", np.array_equal(gc[j], wcard)))
        for name, r in (("device", g), ("oracle", w)):
            print("  %s flags %d vseg_score %r (%08x)" % (name, r["flags"], float(r["vseg_score"]), int(r["vseg_score"].view(np.uint32))))
            print("  %s y_offset %d pattern %d n_offsets %d offsets %s pattern_offset %d hseg_score %r (%08x) number_width %r" % (
                name, r["vseg_y_offset"], r["pattern_type"], r["n_offsets"], r["offsets"].tolist(), r["pattern_offset"],
                float(r["hseg_score"]), int(r["hseg_score"].view(np.uint32)), float(r["number_width"])))
        if not same:
            np.save("gpurun_out/r6_hseg_case_card.npy", wcard)
            yo = int(w["vseg_y_offset"])
            print("  oracle.best_n_hseg on the strip:", o.best_n_hseg(wcard[yo:yo + 27], int(w["pattern_type"])))
