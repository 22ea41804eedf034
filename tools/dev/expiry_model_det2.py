import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as entry
pkg = entry.load_package(); orc = entry.load_oracle(); o = orc.Oracle()
ctx = pkg.Context(0)
rng = np.random.default_rng(1)
x = (rng.integers(0, 256, (4096, 176)) / np.float32(255)).astype(np.float32)
ctx.set_expiry_conv(0); ref = ctx.apply_expiry_model(x)
ctx.set_expiry_conv(1)
for n in (1024, 1028, 1100, 2048, 4096):
    a = ctx.apply_expiry_model(x[:n])
    bad = np.nonzero(np.abs(a - ref[:n]).max(1) > 1e-4)[0]
    print(n, "bad rows", len(bad), bad[:12], "blocks", sorted(set((bad // 4).tolist()))[:12])
    if len(bad):
        r = bad[0]; print("   row", r, a[r][:5], ref[r][:5])
