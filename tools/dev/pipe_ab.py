#!/usr/bin/env python3
"""Developer tool (GPU box): wall-clock ms per pipeline step and a digest of every output byte, for A/Bs of the queue
orchestration (DMZ_HIP_PIPE_CHUNKS, DMZ_HIP_WARP_LDS_PAD, DMZ_HIP_DETECT_LDS_CU are read at context creation / first launch).
usage: python tools/dev/pipe_ab.py [batch] [reps]"""
import hashlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as entry

pkg = entry.load_package()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ctx = pkg.Context(0)
y = ctx.alloc(B * pkg.FRAME_BYTES)
res = ctx.alloc(B * 1024)
cards = ctx.alloc(B * pkg.CARD_BYTES)
exp = ctx.alloc(B * pkg.EXPIRY_DTYPE.itemsize)
ctx.synth_frames(0xCA4D10, 0, B, y.ptr)
for _ in range(2):
    ctx.pipeline_expiry(y.ptr, B, res.ptr, exp.ptr, cards.ptr)
ctx.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    ctx.pipeline_expiry(y.ptr, B, res.ptr, exp.ptr, cards.ptr)
ctx.synchronize()
dt = (time.perf_counter() - t0) / reps
h = hashlib.sha256()
h.update(res.download(np.uint8).tobytes())
h.update(exp.download(np.uint8).tobytes())
hc = hashlib.sha256(cards.download(np.uint8)[: 4096 * pkg.CARD_BYTES].tobytes()).hexdigest()[:12]
tag = " ".join("%s=%s" % (k[8:], v) for k, v in sorted(os.environ.items()) if k.startswith("DMZ_HIP_") and k != "DMZ_HIP_LIB")
print("%-60s B=%d %.3f ms/step %.0f frames/s  records %s cards %s" % (tag or "default", B, dt * 1e3, B / dt, h.hexdigest()[:12], hc))
