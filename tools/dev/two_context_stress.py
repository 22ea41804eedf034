#!/usr/bin/env python3
"""Developer stress test (GPU box): two contexts on ONE GPU, each driven by its own thread -- context A rectifies the same
frames again and again (dmz_hip_transform_batch: k_homography + k_warp) while context B keeps the expiry CNN running
(dmz_hip_scan_expiry_batch).  Every pass of A is compared with its first: the transient fault round 5 found in k_homography
beside another queue's kernels (DESIGN_LOG.md, round 5) would show as cards that move.  usage: two_context_stress.py [frames] [passes]
(DMZ_HIP_LIB selects a library variant, e.g. one built with -DDMZ_HOMOGRAPHY_NOCHECK)"""
import hashlib
import os
import sys
import threading

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as entry

pkg = entry.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 150
a, b = pkg.Context(0), pkg.Context(0)
# A: frames + detected records; B: its own cards / records with expiry work
ya, ra, ca = a.alloc(n * pkg.FRAME_BYTES), a.alloc(n * 1024), a.alloc(n * pkg.CARD_BYTES)
a.synth_frames(0xCA4D10, 0, n, ya.ptr)
a.detect(ya.ptr, n, ra.ptr)
a.synchronize()
yb, rb, cb, xb = b.alloc(n * pkg.FRAME_BYTES), b.alloc(n * 1024), b.alloc(n * pkg.CARD_BYTES), b.alloc(n * pkg.EXPIRY_DTYPE.itemsize)
b.synth_frames(0xCA4D10, n, n, yb.ptr)
b.pipeline_expiry(yb.ptr, n, rb.ptr, xb.ptr, cb.ptr)
b.synchronize()
stop = threading.Event()


def hog():
    while not stop.is_set():
        b.scan_expiry(cb.ptr, n, rb.ptr, xb.ptr)
        b.synchronize()


t = threading.Thread(target=hog)
t.start()
ref = None
moved = 0
for p in range(passes):
    a.transform(ya.ptr, n, ra.ptr, ca.ptr)
    a.synchronize()
    cards = ca.download(np.uint8).reshape(n, -1)
    sig = np.array([int(hashlib.md5(c.tobytes()).hexdigest()[:8], 16) for c in cards[::1]], np.int64) if n <= 4096 else cards[:, ::97].astype(np.int64).sum(1) * 31 + cards[:, 5::89].astype(np.int64).sum(1)
    if ref is None:
        ref = sig
    else:
        bad = np.nonzero(sig != ref)[0]
        if len(bad):
            moved += 1
            print("pass %d: %d cards differ from the first pass, frames %s" % (p, len(bad), bad[:20]))
stop.set()
t.join()
print("%s: %d passes of %d frames beside the other context's expiry CNN, %d passes with cards that moved" % (
    os.path.basename(os.environ.get("DMZ_HIP_LIB", "default library")), passes, n, moved))
