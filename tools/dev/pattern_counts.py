import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import __graft_entry__ as entry
pkg = entry.load_package()
B = 16384
ctx = pkg.Context(0)
y = ctx.alloc(B * pkg.FRAME_BYTES); res = ctx.alloc(B * 1024)
ctx.synth_frames(0xCA4D10, 0, B, y.ptr)
r = np.zeros(B, pkg.RESULT_DTYPE)
ctx.pipeline(y.ptr, B, r)
print("pattern_type histogram:", np.bincount(r["pattern_type"], minlength=3), "n_offsets:", np.bincount(r["n_offsets"], minlength=17)[[0,15,16]])
