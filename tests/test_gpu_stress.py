"""Stress sized to a rare event.  Round 5 saw `k_homography` -- a register-to-register float computation
-- come out wrong for one quarter-wave about once per 65 536 frames when another queue's kernels ran beside it
(DESIGN_LOG.md, "transient fault"); the kernel has evaluated until two consecutive results agree ever since, and a fault that
never settles is flagged (DMZ_HIP_FLAG_FAULT) instead of used.  Nothing in the suite would have noticed a recurrence: this test
rectifies 65 536 frames again and again on one context while a second context on the same GPU keeps the expiry CNN running, and
compares every pass with the first -- every record byte, and the focus / brightness scores of every card as its checksum (a wrong
homography moves ~90 000 of a card's 115 560 bytes).  `tools/dev/homography_fault.sh` is the developer form with the self-check
compiled out under the one queue pattern that showed the fault."""
import os
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SEED = 0xCA4D10


def test_transform_beside_another_contexts_expiry_cnn(pkg):
    n = int(os.environ.get("DMZ_STRESS_FRAMES", "65536"))
    passes = int(os.environ.get("DMZ_STRESS_PASSES", "24"))
    nb = min(n, 16384)
    a, b = pkg.Context(0), pkg.Context(0)
    try:
        ya, ra, ca = a.alloc(n * pkg.FRAME_BYTES), a.alloc(n * 1024), a.alloc(n * pkg.CARD_BYTES)
        fa, ba = a.alloc(n * 4), a.alloc(n * 4)
        a.synth_frames(SEED, 0, n, ya.ptr)
        a.detect(ya.ptr, n, ra.ptr)
        a.synchronize()
        yb, rb, cb = b.alloc(nb * pkg.FRAME_BYTES), b.alloc(nb * 1024), b.alloc(nb * pkg.CARD_BYTES)
        xb = b.alloc(nb * pkg.EXPIRY_DTYPE.itemsize)
        b.synth_frames(SEED, n, nb, yb.ptr)
        b.pipeline_expiry(yb.ptr, nb, rb.ptr, xb.ptr, cb.ptr)
        b.synchronize()
        stop = threading.Event()
        hog_error = []

        def hog():
            try:
                while not stop.is_set():
                    b.scan_expiry(cb.ptr, nb, rb.ptr, xb.ptr)
                    b.synchronize()
            except Exception as e:  # noqa: BLE001 (reported by the main thread)
                hog_error.append(e)

        t = threading.Thread(target=hog)
        t.start()
        first = None
        moved = []
        try:
            for p in range(passes):
                a.transform(ya.ptr, n, ra.ptr, ca.ptr)
                a.scores(ca.ptr, n, fa.ptr, ba.ptr, width=pkg.CARD_W, height=pkg.CARD_H, use_full_image=True)
                a.synchronize()
                got = (ra.download(np.uint8).reshape(n, 1024).copy(), fa.download(np.uint32).copy(), ba.download(np.uint32).copy())
                if first is None:
                    first = got
                    continue
                bad = np.nonzero((got[0] != first[0]).any(axis=1) | (got[1] != first[1]) | (got[2] != first[2]))[0]
                if len(bad):
                    moved.append((p, bad[:8].tolist()))
        finally:
            stop.set()
            t.join()
        assert not hog_error, hog_error
        res = first[0].view(pkg.RESULT_DTYPE).reshape(-1)
        assert ((res["flags"] & pkg.FLAG_WARPED) != 0).mean() > 0.99 and not (res["flags"] & pkg.FLAG_FAULT).any()
        assert not moved, "cards or records moved between passes: %s" % moved[:4]
    finally:
        a.close()
        b.close()
