import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import __graft_entry__ as entry
pkg = entry.load_package()
B = 8192
ctx = pkg.Context(0)
y = ctx.alloc(B * pkg.FRAME_BYTES); res = ctx.alloc(B * 1024); exp = ctx.alloc(B * pkg.EXPIRY_DTYPE.itemsize)
ctx.synth_frames(0xCA4D10, 0, B, y.ptr)
ctx.pipeline_expiry(y.ptr, B, res.ptr, exp.ptr); ctx.synchronize()
e = exp.download(pkg.EXPIRY_DTYPE, B)
print("n_stripes histogram:", np.bincount(e["n_stripes"], minlength=4))
print("n_groups histogram:", np.bincount(e["n_groups"], minlength=9))
print("n_found histogram:", np.bincount(np.minimum(e["n_found"], 12), minlength=13))
