"""BASELINE.json's full size (configs[3]: 65 536 frames on one GPU) through size-independent properties:
determinism, independence of a frame's records from batch size and position, stage-chain consistency,
and a random sample of the batch against the oracle."""
import hashlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SEED, N = 0xCA4D10, 65536


def test_full_batch_properties(ctx, pkg, oracle):
    y = ctx.alloc(N * pkg.FRAME_BYTES)
    cards = ctx.alloc(N * pkg.CARD_BYTES)
    res = ctx.alloc(N * 1024)
    exp = ctx.alloc(N * pkg.EXPIRY_DTYPE.itemsize)
    ctx.synth_frames(SEED, 0, N, y.ptr)
    ctx.pipeline_expiry(y.ptr, N, res.ptr, exp.ptr, cards.ptr)
    ctx.synchronize()
    r1 = res.download(np.uint8)
    x1 = exp.download(np.uint8)
    h_res, h_exp = hashlib.sha256(r1).hexdigest(), hashlib.sha256(x1).hexdigest()

    # 1. determinism: the same batch again gives the same bytes (LDS atomics are integer-only)
    ctx.pipeline_expiry(y.ptr, N, res.ptr, exp.ptr, cards.ptr)
    ctx.synchronize()
    assert hashlib.sha256(res.download(np.uint8)).hexdigest() == h_res
    assert hashlib.sha256(exp.download(np.uint8)).hexdigest() == h_exp

    # 2. a frame's records do not depend on the batch it is scanned in: chunks of odd sizes at odd offsets
    rs, xs = pkg.RESULT_DTYPE.itemsize, pkg.EXPIRY_DTYPE.itemsize
    small_res = ctx.alloc(8200 * rs)
    small_exp = ctx.alloc(8200 * xs)
    rng = np.random.default_rng(5)
    for lo, m in ((0, 8192), (8191, 8200), (30001, 777), (65535, 1), (57344, 8192), (40000, 4099)):
        ctx.pipeline_expiry(y.ptr + lo * pkg.FRAME_BYTES, m, small_res.ptr, small_exp.ptr)
        ctx.synchronize()
        assert np.array_equal(small_res.download(np.uint8, m * rs), r1[lo * rs:(lo + m) * rs]), (lo, m)
        assert np.array_equal(small_exp.download(np.uint8, m * xs), x1[lo * xs:(lo + m) * xs]), (lo, m)

    # 3. chain consistency: the stage entry points on the pipeline's own cards reproduce its scan fields
    rec = r1.view(pkg.RESULT_DTYPE)
    lo, m = 12345, 4096
    part = rec[lo:lo + m].copy()
    again = part.copy()
    for f in ("vseg_score", "vseg_y_offset", "pattern_type", "n_offsets", "offsets", "hseg_score", "number_width",
              "pattern_offset", "number_score", "digits", "scores"):
        again[f] = 0
    again["flags"] &= pkg.FLAG_WARPED
    ctx.scan_cards(cards.ptr + lo * pkg.CARD_BYTES, m, again, only_warped=True)
    assert again.tobytes() == part.tobytes()
    exp_again = np.zeros(m, pkg.EXPIRY_DTYPE)
    ctx.scan_expiry(cards.ptr + lo * pkg.CARD_BYTES, m, part, exp_again)
    assert exp_again.tobytes() == x1[lo * xs:(lo + m) * xs].tobytes()

    # 4. a random sample of the batch against the oracle
    frames = None
    for i in rng.choice(N, 48, replace=False):
        i = int(i)
        frame, _ = oracle.synth_frame(SEED, i)
        want, wcard = oracle.scan_frame(frame)
        g = rec[i]
        assert np.array_equal(g["found"], want["found"]) and np.array_equal(g["corners"].view(np.uint32), want["corners"].view(np.uint32)), i
        assert g["flags"] == want["flags"] and g["vseg_y_offset"] == want["vseg_y_offset"], i
        assert np.array_equal(g["offsets"], want["offsets"]) and np.array_equal(g["digits"], want["digits"]), i
        assert np.abs(g["scores"] - want["scores"]).max() <= 1e-4, i
        we = oracle.scan_card_expiry(wcard, want)
        ge = x1.view(pkg.EXPIRY_DTYPE)[i]
        assert ge["n_found"] == we["n_found"] and ge["n_stripes"] == we["n_stripes"], i
        k = int(we["n_groups"])
        assert np.array_equal(ge["groups"]["char_left"][:k], we["groups"]["char_left"][:k]), i

    # 5. gate statistics of the corpus stay where DESIGN.md reports them
    usable = float(((rec["flags"] & pkg.FLAG_USABLE) != 0).mean())
    assert 0.6 < usable < 0.75 and bool((rec["found_all"] != 0).all())
    for b in (y, cards, res, exp, small_res, small_exp):
        b.free()
