import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as entry
pkg = entry.load_package()
ctx = pkg.Context(0)
rng = np.random.default_rng(1)
x = (rng.integers(0, 256, (1024, 176)) / np.float32(255)).astype(np.float32)
x = np.concatenate([x] * 4)  # rows 1024.. repeat rows 0..1023: every copy must give the same bits
for mode in (0, 1, 2):
    ctx.set_expiry_conv(mode)
    a = ctx.apply_expiry_model(x).reshape(4, 1024, 10)
    print("mode", mode, "rows differing from the first copy:", [int((a[k] != a[0]).any(1).sum()) for k in (1, 2, 3)],
          "max", float(np.abs(a - a[0]).max()))
