import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as entry
pkg = entry.load_package()
ctx = pkg.Context(0)
n = 16384
y = ctx.alloc(n * pkg.FRAME_BYTES); res = ctx.alloc(n * 1024); exp = ctx.alloc(n * pkg.EXPIRY_DTYPE.itemsize)
ctx.synth_frames(0xCA4D10, 0, n, y.ptr)
for mode in (0, 1, 2):
    ctx.set_expiry_conv(mode)
    outs = []
    for rep in range(3):
        ctx.pipeline_expiry(y.ptr, n, res.ptr, exp.ptr); ctx.synchronize()
        outs.append(exp.download(pkg.EXPIRY_DTYPE, n).copy())
    a, b, c = outs
    for name in ("n_groups", "n_found", "n_stripes", "stripe_base_row", "stripe_sum", "categorised"):
        print(mode, name, int((a[name] != b[name]).sum()), int((a[name] != c[name]).sum()))
    ga, gb = a["groups"], b["groups"]
    for name in ("top", "left", "char_top", "char_left", "scores"):
        d = (ga[name] != gb[name])
        print(mode, "groups." + name, int(d.sum()), "frames", int(d.reshape(n, -1).any(1).sum()))
    d = (ga["scores"] != gb["scores"]).reshape(n, -1).any(1)
    if d.any():
        f = int(np.nonzero(d)[0][0])
        print("first frame", f, "n_groups", a["n_groups"][f], "maxdiff", np.abs(ga["scores"][f] - gb["scores"][f]).max())
        print(ga["scores"][f][:a["n_groups"][f]].reshape(-1, 10)[:4]); print(gb["scores"][f][:a["n_groups"][f]].reshape(-1, 10)[:4])
