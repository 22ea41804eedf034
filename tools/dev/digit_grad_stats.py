#!/usr/bin/env python3
"""Developer probe (GPU box): distribution of the 5-tap cross gradient inside the digit ROIs of the benchmark corpus -- how often the
lanes of k_digit_patches that share a histogram copy (same row, four columns apart) hold the same value."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as entry
pkg = entry.load_package()
B = 256
ctx = pkg.Context(0)
y = ctx.alloc(B * pkg.FRAME_BYTES); res = ctx.alloc(B * 1024); cards = ctx.alloc(B * pkg.CARD_BYTES); exp = ctx.alloc(B * pkg.EXPIRY_DTYPE.itemsize)
ctx.synth_frames(0xCA4D10, 0, B, y.ptr)
ctx.pipeline_expiry(y.ptr, B, res.ptr, exp.ptr, cards.ptr); ctx.synchronize()
r = res.download(pkg.RESULT_DTYPE, B)
c = cards.download(np.uint8).reshape(B, 270, 428).astype(np.int32)
hist = np.zeros(256, np.int64); same4 = tot4 = 0; digits = 0
for f in range(B):
    if not (r["flags"][f] & pkg.FLAG_VSEG_OK): continue
    y0 = int(r["vseg_y_offset"][f])
    for d in range(int(r["n_offsets"][f])):
        x0 = int(r["offsets"][f][d])
        roi = c[f, y0:y0 + 27, x0:x0 + 19]
        p = np.pad(roi, 1, mode="edge")
        taps = np.stack([p[1:-1, 1:-1], p[:-2, 1:-1], p[2:, 1:-1], p[1:-1, :-2], p[1:-1, 2:]])
        g = taps.max(0) - taps.min(0)
        hist += np.bincount(g.ravel(), minlength=256)
        same4 += int((g[:, 4:] == g[:, :-4]).sum()); tot4 += g[:, 4:].size; digits += 1
tot = hist.sum()
print("digits %d; gradient == 0: %.1f %%; <= 2: %.1f %%; <= 8: %.1f %%; the ten most frequent values hold %.1f %%; equal to the pixel four columns to the left: %.1f %%"
      % (digits, 100 * hist[0] / tot, 100 * hist[:3].sum() / tot, 100 * hist[:9].sum() / tot, 100 * np.sort(hist)[-10:].sum() / tot, 100 * same4 / tot4))
print("most frequent:", [(int(v), round(100 * hist[v] / tot, 1)) for v in np.argsort(hist)[::-1][:8]])
