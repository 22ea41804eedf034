#!/usr/bin/env python3
"""Developer tool (GPU box): run the pipeline `reps` times over one resident batch and report every record field and card that
differs from the first run (a race, an uninitialised read or a transient fault shows up as a frame whose outputs move).  Every
CU's LDS is overwritten with a NaN / non-NaN pattern in front of each run.
usage: python tools/dev/determinism.py [batch] [reps] [seed]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as entry

pkg = entry.load_package()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
seed = int(sys.argv[3], 0) if len(sys.argv) > 3 else 0xCA4D10
ctx = pkg.Context(0)
y = ctx.alloc(B * pkg.FRAME_BYTES)
res = ctx.alloc(B * 1024)
cards = ctx.alloc(B * pkg.CARD_BYTES)
ref_cards = ctx.alloc(B * pkg.CARD_BYTES)
exp = ctx.alloc(B * pkg.EXPIRY_DTYPE.itemsize)
ctx.synth_frames(seed, 0, B, y.ptr)
first = None
tag = " ".join("%s=%s" % (k[8:], v) for k, v in sorted(os.environ.items()) if k.startswith("DMZ_HIP_") and k != "DMZ_HIP_LIB")


def card_of(buf, f):
    out = np.empty(pkg.CARD_BYTES, np.uint8)
    ctx._check(ctx.lib.dmz_hip_memcpy_d2h(ctx.h, out.ctypes.data, buf.ptr + f * pkg.CARD_BYTES, out.nbytes))
    return out.reshape(270, 428)


events = 0
for r in range(reps):
    ctx.debug_fill_lds(0xFFFFFFFF if r % 2 else 0x7F7F7F7F)
    ctx.pipeline_expiry(y.ptr, B, res.ptr, exp.ptr, (ref_cards if r == 0 else cards).ptr)
    ctx.synchronize()
    got = (res.download(pkg.RESULT_DTYPE, B), exp.download(pkg.EXPIRY_DTYPE, B))
    if first is None:
        first = got
        continue
    for name, a, b in (("result", first[0], got[0]), ("expiry", first[1], got[1])):
        bad = np.nonzero((a.view(np.uint8).reshape(B, -1) != b.view(np.uint8).reshape(B, -1)).any(axis=1))[0]
        if len(bad):
            events += 1
            print("[%s] run %d: %d %s records differ, frames %s" % (tag, r, len(bad), name, bad[:40]))
            for f in bad[:2]:
                for fld in a.dtype.names:
                    if not np.array_equal(a[f][fld], b[f][fld]):
                        print("    frame %d field %s: %s -> %s" % (f, fld, str(a[f][fld]).replace("\n", " ")[:60], str(b[f][fld]).replace("\n", " ")[:60]))
            if name == "result":
                for f in list(bad[:3]) + [int(bad[-1])]:
                    c0, c1 = card_of(ref_cards, int(f)), card_of(cards, int(f))
                    d = c0 != c1
                    print("    frame %d: card bytes that differ from the first run's: %d (rows %s)" % (f, int(d.sum()), np.nonzero(d.any(1))[0][:6]))
print("[%s] %d runs of %d frames compared, %d with differences" % (tag, reps, B, events))
# a library built with -DDMZ_DEV_SELFCHECK (tools/dev/homography_fault.sh selfcheck) counts double evaluations that disagreed
import ctypes
for name, what in (("dmz_dbg_selfcheck_geom", "k_geometry corner sets"), ("dmz_dbg_selfcheck_warp", "k_warp_windows strip windows")):
    if hasattr(ctx.lib, name):
        out = (ctypes.c_ulonglong * 2)()
        getattr(ctx.lib, name)(out)
        print("[%s] %s evaluated twice: %d, differed: %d" % (tag, what, out[0], out[1]))

# a library built with -DDMZ_DEV_HTRACE (tools/dev/homography_fault.sh trace): k_homography evaluated twice with a hash of its
# state after every Householder step; for every pair that disagreed, the first step whose hashes differ
if hasattr(ctx.lib, "dmz_dbg_htrace"):
    buf = (ctypes.c_uint * (64 * 40))()
    nev = ctx.lib.dmz_dbg_htrace(buf)
    ev = np.frombuffer(buf, np.uint32).reshape(64, 40)
    names = ["pivot / coefficient of Householder step %d" % k for k in range(8)] + ["Q^T b", "back substitution"]
    print("[%s] k_homography pairs that disagreed: %d" % (tag, nev))
    for e in ev[:min(nev, 64)]:
        t1, t2 = e[1:11], e[11:21]
        d = np.nonzero(t1 != t2)[0]
        m1, m2 = e[21:30].view(np.float32), e[30:39].view(np.float32)
        print("    frame %d lane %d: first difference at %s (steps that differ: %s); matrix elements that differ: %s"
              % (e[0], e[39] & 63, names[d[0]] if len(d) else "none of the traced states (the result only)", d.tolist(),
                 np.nonzero(m1.view(np.uint32) != m2.view(np.uint32))[0].tolist()))

# -DDMZ_DEV_HCANARY (tools/dev/homography_fault.sh canary): registers k_homography never uses held a pattern from its first to its
# last instruction; every register found changed at the end, with the lanes that changed
if hasattr(ctx.lib, "dmz_dbg_hcanary"):
    buf = (ctypes.c_uint * (256 * 4))()
    nev = ctx.lib.dmz_dbg_hcanary(buf)
    ev = np.frombuffer(buf, np.uint32).reshape(256, 4)
    print("[%s] k_homography canary registers found changed: %d" % (tag, nev))
    for e in ev[:min(nev, 256)]:
        mask = int(e[2]) | (int(e[3]) << 32)
        print("    wave of frames %d..: v%d changed in lanes %s" % (e[0], e[1], [l for l in range(64) if (mask >> l) & 1]))
