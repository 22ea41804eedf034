import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import __graft_entry__ as entry
pkg = entry.load_package()
from oracle import orc
oracle = orc.Oracle()
ctx = pkg.Context(0)
rng = np.random.default_rng(12)
x = (rng.integers(0, 256, (513, 176)) / np.float32(255)).astype(np.float32)
x[:64] = rng.random((64, 176), dtype=np.float32)
want = np.stack([oracle.applyc_expiry(v)[0] for v in x])
out = {}
for name, mode in (("f32", 0), ("bf16x3", 1), ("bf16", 2), ("f16x3", 3)):
    ctx.set_expiry_conv(mode)
    out[name] = ctx.apply_expiry_model(x)
    print("%-7s max|score-oracle| %.3g  vs f32 %.3g  labels %.5f" % (name, np.abs(out[name] - want).max(), np.abs(out[name] - out["f32"]).max(), (out[name].argmax(1) == want.argmax(1)).mean()))
# timing
B = 16384
y = ctx.alloc(B * pkg.FRAME_BYTES); res = ctx.alloc(B * 1024); exp = ctx.alloc(B * pkg.EXPIRY_DTYPE.itemsize)
ctx.synth_frames(0xCA4D10, 0, B, y.ptr)
for name, mode in (("bf16x3", 1), ("f16x3", 3), ("bf16x3", 1), ("f16x3", 3)):
    ctx.set_expiry_conv(mode)
    ctx.pipeline_expiry(y.ptr, B, res.ptr, exp.ptr); ctx.synchronize()
    ctx.set_profiling(True); ctx.stage_times()
    for _ in range(3):
        ctx.pipeline_expiry(y.ptr, B, res.ptr, exp.ptr)
    ctx.synchronize()
    t = ctx.stage_times()
    print(name, {k: round(v[0] / 3, 3) for k, v in t.items() if "expiry" in k})
    ctx.set_profiling(False)
