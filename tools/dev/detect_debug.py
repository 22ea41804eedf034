import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as entry
pkg = entry.load_package(); orc = entry.load_oracle(); o = orc.Oracle()
ctx = pkg.Context(0)
SEED = 0xCA4D10
rng = np.random.default_rng(99)
n = 40
ys = []
for i in range(n):
    y, _ = o.synth_frame(SEED, 700 + i)
    y = y.copy()
    wipes = [(slice(80, 130), slice(None)), (slice(350, 400), slice(None)), (slice(None), slice(80, 135)), (slice(None), slice(505, 560))]
    for e, (rs, cs) in enumerate(wipes):
        r = rng.random()
        if r < 0.35:
            y[rs, cs] = 60
    ys.append(y)
ys = np.ascontiguousarray(np.stack(ys))
res = np.zeros(n, pkg.RESULT_DTYPE)
ctx.detect(ys, n, res)
bad = 0
for i in range(n):
    w = o.detect_edges(ys[i])
    if not (np.array_equal(res[i]["found"], w["found"]) and np.array_equal(res[i]["rho"].view(np.uint32), w["rho"].view(np.uint32)) and np.array_equal(res[i]["theta"].view(np.uint32), w["theta"].view(np.uint32))):
        bad += 1
        print(i, "got", res[i]["found"], res[i]["rho"], res[i]["theta"], "want", w["found"], w["rho"], w["theta"])
print("bad", bad)
