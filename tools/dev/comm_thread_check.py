"""Developer check (GPU box): dmz_hip_comm_init from a helper thread (as bench.py does under its deadline), then a world-size-1 gather."""
import os, sys, threading
sys.path.insert(0, os.getcwd())
import numpy as np
import __graft_entry__ as entry
pkg = entry.load_package()
ctx = pkg.Context(0)
uid = pkg.comm_unique_id()
err = []
def _init():
    try:
        ctx.comm_init(1, 0, uid)
    except pkg.DmzHipError as e:
        err.append(str(e))
th = threading.Thread(target=_init, daemon=True); th.start(); th.join(120)
print("alive", th.is_alive(), "err", err)
n = 256
res = ctx.alloc(n * 1024); dst = ctx.alloc(n * 1024)
res.upload(np.arange(n * 1024, dtype=np.uint8) if False else (np.arange(n * 1024) % 251).astype(np.uint8))
ctx.gather_records(res.ptr, 1024, n, 0, dst.ptr, slot=0)
ctx.gather_wait(-1, host_sync=True)
print("gather equal:", np.array_equal(res.download(np.uint8), dst.download(np.uint8)))
ctx.comm_destroy(); ctx.close()
