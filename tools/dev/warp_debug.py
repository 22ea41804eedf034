import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as entry
pkg = entry.load_package(); orc = entry.load_oracle(); o = orc.Oracle()
ctx = pkg.Context(0)
rng = np.random.default_rng(0)
yy, xx = np.mgrid[0:480, 0:640]
src = ((xx + 2 * yy) % 256).astype(np.uint8)
for name, q in (("crop", [100, 90, 527, 90, 100, 359, 527, 359]), ("half", [100.5, 90.25, 527.5, 90.25, 100.5, 359.25, 527.5, 359.25]),
                ("persp", [106, 105, 533, 108, 103, 374, 536, 371])):
    m = o.calc_persp_transform(np.array(q, np.float32), np.array([0, 0, 427, 0, 0, 269, 427, 269], np.float32))
    want = o.warp_perspective(src, m)
    got = np.zeros((1, 270, 428), np.uint8)
    ctx.warp_perspective(src[None].copy(), 1, m[None].copy(), got)
    got = got[0]
    bad = got != want
    print(name, "diff", int(bad.sum()), "rows with diff", np.nonzero(bad.any(1))[0][:12], "cols", np.nonzero(bad.any(0))[0][:12])
    if bad.any():
        ys, xs = np.nonzero(bad)
        for k in range(min(6, len(ys))):
            print("   ", ys[k], xs[k], "got", got[ys[k], xs[k]], "want", want[ys[k], xs[k]], "src", src[q[1].__int__() + ys[k] if name == "crop" else 0, 0])
        print(got[:3, :8], want[:3, :8])
