#!/usr/bin/env python3
"""Developer probe (GPU box): detect on card-less noise frames (bench.py --corpus mixed, kind 0-3).
usage: tools/dev/detect_noise.py [batch]     (a library built with -DDMZ_DT_TIMING prints the phase timeline)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as entry
pkg = entry.load_package()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ctx = pkg.Context(0)
rng = np.random.default_rng(1)
frames = rng.integers(18, 58, (B, 480, 640), dtype=np.uint8)
y = ctx.alloc(frames.nbytes).upload(frames)
res = ctx.alloc(B * 1024)
ctx.detect(y.ptr, B, res.ptr)
ctx.set_profiling(True)
ctx.stage_times()
for _ in range(2):
    ctx.detect(y.ptr, B, res.ptr)
t = ctx.stage_times()
print("noise frames B=%d:" % B, {k: round(v[0] / 2, 3) for k, v in t.items() if v[1]})
r = res.download(pkg.RESULT_DTYPE)
print("found per box:", r["found"].mean(axis=0))
