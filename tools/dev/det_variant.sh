#!/bin/bash
# GPU box: run-to-run determinism of the expiry records with a library variant (default mode only)
cd "$(dirname "$0")/../.."
DMZ_HIP_LIB=$PWD/variants/$1.so python3 - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import __graft_entry__ as entry
pkg = entry.load_package()
ctx = pkg.Context(0)
n = 16384
y = ctx.alloc(n * pkg.FRAME_BYTES); res = ctx.alloc(n * 1024); exp = ctx.alloc(n * pkg.EXPIRY_DTYPE.itemsize)
ctx.synth_frames(0xCA4D10, 0, n, y.ptr)
outs = []
for rep in range(4):
    ctx.pipeline_expiry(y.ptr, n, res.ptr, exp.ptr); ctx.synchronize()
    outs.append(exp.download(pkg.EXPIRY_DTYPE, n).copy())
for k in range(1, 4):
    d = (outs[0]["groups"]["scores"] != outs[k]["groups"]["scores"]).reshape(n, -1).any(1)
    print("rep", k, "records differing (all bytes):", int((outs[0].view(np.uint8).reshape(n, -1) != outs[k].view(np.uint8).reshape(n, -1)).any(1).sum()))
    print("rep", k, "frames with different scores:", int(d.sum()), "max diff", float(np.abs(outs[0]["groups"]["scores"] - outs[k]["groups"]["scores"]).max()))
PY
