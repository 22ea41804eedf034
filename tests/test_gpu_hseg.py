"""GPU parity of the digit x-offset search alone (dmz_hip_best_n_hseg_batch = best_n_hseg, n_hseg.cpp:88-151) on strips built to
stress its filtered form: the device decides a pass from table scores when the best candidate leads by more than the rounding
noise of the reference's ordered float sum, and falls back to the ordered sums otherwise (csrc/hseg.hip).  Flat and periodic strips
make many candidates (nearly) equal; every output must still carry the oracle's bits."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SEED = 90210


def _strips(oracle, rng, n_each):
    """(card, y_offset, pattern) triples; only the 27 rows at y_offset matter"""
    out = []

    def card_with(strip, y):
        card = rng.integers(0, 256, (270, 428), dtype=np.uint8)
        card[y:y + 27] = strip
        return card

    for i in range(n_each):  # synthetic cards at and around their own number line, both patterns
        card, _ = oracle.synth_card(SEED, i)
        _, y, p, _, _ = oracle.best_n_vseg(card)
        for dy in (0, -2, 3):
            yy = int(np.clip(y + dy, 0, 243))
            out.append((card, yy, 1 + (i + dy) % 2))
    for i in range(n_each):  # flat strips: every column sum is equal, every in-bounds candidate scores the same
        y = int(rng.integers(0, 244))
        out.append((card_with(np.full((27, 428), rng.integers(0, 256), np.uint8), y), y, 1 + i % 2))
    for i in range(n_each):  # vertical bars whose period is a digit width: candidates one period apart tie
        y = int(rng.integers(0, 244))
        period = rng.uniform(15.5, 20.5)
        x = np.arange(428)
        bars = ((np.floor((x + rng.uniform(0, period)) / period * 2) % 2) * rng.integers(40, 255)).astype(np.uint8)
        strip = np.repeat(bars[None, :], 27, axis=0)
        if i % 3 == 0:
            strip = (strip // 2 + rng.integers(0, 3, strip.shape)).astype(np.uint8)
        out.append((card_with(strip, y), y, 1 + i % 2))
    for i in range(n_each):  # noise, and noise in a few levels
        y = int(rng.integers(0, 244))
        strip = rng.integers(0, 256, (27, 428), dtype=np.uint8)
        if i % 2:
            strip = (strip // 64 * 64).astype(np.uint8)
        out.append((card_with(strip, y), y, 1 + i % 2))
    for i in range(n_each):  # a number-like strip only in part of the width: offsets at both ends of the range
        y = int(rng.integers(0, 244))
        strip = np.full((27, 428), 30, np.uint8)
        lo = int(rng.integers(0, 200))
        hi = int(rng.integers(lo + 40, 428))
        strip[:, lo:hi] = rng.integers(0, 256, (27, hi - lo), dtype=np.uint8)
        out.append((card_with(strip, y), y, 1 + i % 2))
    return out


def test_best_n_hseg_on_stress_strips(ctx, pkg, oracle):
    rng = np.random.default_rng(SEED)
    cases = _strips(oracle, rng, 24)
    n = len(cases)
    cards = np.stack([c[0] for c in cases])
    res = np.zeros(n, pkg.RESULT_DTYPE)
    res["flags"] = pkg.FLAG_VSEG_OK
    res["vseg_y_offset"] = [c[1] for c in cases]
    res["pattern_type"] = [c[2] for c in cases]
    res["vseg_score"] = 0.5
    before = res.copy()
    ctx.best_n_hseg(cards, n, res)
    for i, (card, y, p) in enumerate(cases):
        n_off, offsets, score, width, po = oracle.best_n_hseg(card[y:y + 27], p)
        g = res[i]
        assert g["n_offsets"] == n_off, i
        assert np.array_equal(g["offsets"], offsets), (i, g["offsets"], offsets)
        assert g["hseg_score"].view(np.uint32) == np.float32(score).view(np.uint32), (i, g["hseg_score"], score)
        assert g["number_width"].view(np.uint32) == np.float32(width).view(np.uint32), (i, g["number_width"], width)
        assert g["pattern_offset"] == po, i
        # nothing else in the record moves
        for name in ("flags", "vseg_y_offset", "pattern_type", "vseg_score"):
            assert g[name] == before[i][name], (i, name)


def test_best_n_hseg_skips_records_without_a_segmentation_and_rejects_bad_ones(ctx, pkg, oracle):
    card, _ = oracle.synth_card(SEED, 0)
    cards = np.stack([card, card])
    res = np.zeros(2, pkg.RESULT_DTYPE)
    res["flags"][1] = pkg.FLAG_VSEG_OK
    res["vseg_y_offset"][1] = 100
    res["pattern_type"][1] = 1
    ctx.best_n_hseg(cards, 2, res)
    assert res["n_offsets"][0] == 0 and res["hseg_score"][0] == 0 and res["n_offsets"][1] == 16
    bad = res.copy()
    bad["vseg_y_offset"][1] = 250
    with pytest.raises(Exception):
        ctx.best_n_hseg(cards, 2, bad)
    bad = res.copy()
    bad["pattern_type"][1] = 0
    with pytest.raises(Exception):
        ctx.best_n_hseg(cards, 2, bad)


def test_best_n_hseg_on_device_resident_buffers(ctx, pkg, oracle):
    n = 24
    dcards = ctx.alloc(n * pkg.CARD_BYTES)
    ctx.synth_cards(SEED, 0, n, dcards.ptr)
    cards = dcards.download(np.uint8).reshape(n, 270, 428)
    res = np.zeros(n, pkg.RESULT_DTYPE)
    ys = []
    for i in range(n):
        _, y, p, _, _ = oracle.best_n_vseg(cards[i])
        ys.append((int(np.clip(y, 0, 243)), p if p else 1))
    res["flags"] = pkg.FLAG_VSEG_OK
    res["vseg_y_offset"] = [a for a, _ in ys]
    res["pattern_type"] = [b for _, b in ys]
    dres = ctx.alloc(n * res.dtype.itemsize)
    dres.upload(res.view(np.uint8))
    ctx.best_n_hseg(dcards.ptr, n, dres.ptr)
    ctx.synchronize()
    got = dres.download(pkg.RESULT_DTYPE, n)
    for i in range(n):
        n_off, offsets, score, width, po = oracle.best_n_hseg(cards[i][ys[i][0]:ys[i][0] + 27], ys[i][1])
        assert got[i]["n_offsets"] == n_off and np.array_equal(got[i]["offsets"], offsets), i
        assert got[i]["hseg_score"].view(np.uint32) == np.float32(score).view(np.uint32), i
        assert got[i]["number_width"].view(np.uint32) == np.float32(width).view(np.uint32) and got[i]["pattern_offset"] == po, i


def test_best_n_hseg_does_not_depend_on_lds_leftovers(ctx, pkg, oracle):
    """The masked tail terms of a candidate's score are fma(|g|, 0, s): exact only for finite g, and the tail reads run up to
    the WAVE's longest segment -- past column 427 into g's zero padding.  Words behind that padding would be leftovers of
    whatever workgroup used the CU's LDS before: here every CU's LDS is filled with 0xFFFFFFFF (a NaN) and then with 0 in
    front of the search; the records must not move, for both patterns, and equal the oracle's on a sample."""
    n = 8192
    dcards = ctx.alloc(n * pkg.CARD_BYTES)
    ctx.synth_cards(SEED + 5, 0, n, dcards.ptr)
    rng = np.random.default_rng(SEED + 5)
    res = np.zeros(n, pkg.RESULT_DTYPE)
    res["flags"] = pkg.FLAG_VSEG_OK
    res["vseg_y_offset"] = rng.integers(0, 244, n)
    res["pattern_type"] = 1 + (np.arange(n) % 2)
    dres = ctx.alloc(n * res.dtype.itemsize)
    got = []
    for word in (0xFFFFFFFF, 0x00000000, 0x7F800000):  # NaN, zero, +Inf
        dres.upload(res.view(np.uint8))
        ctx.debug_fill_lds(word)
        ctx.best_n_hseg(dcards.ptr, n, dres.ptr)
        ctx.synchronize()
        got.append(dres.download(pkg.RESULT_DTYPE, n))
    for other in got[1:]:
        assert np.array_equal(got[0].view(np.uint8), other.view(np.uint8))
    cards = dcards.download(np.uint8).reshape(n, 270, 428)
    for i in range(0, n, 128):
        y, p = int(res["vseg_y_offset"][i]), int(res["pattern_type"][i])
        n_off, offsets, score, width, po = oracle.best_n_hseg(cards[i][y:y + 27], p)
        g = got[0][i]
        assert g["n_offsets"] == n_off and np.array_equal(g["offsets"], offsets), i
        assert g["hseg_score"].view(np.uint32) == np.float32(score).view(np.uint32), i
        assert g["number_width"].view(np.uint32) == np.float32(width).view(np.uint32) and g["pattern_offset"] == po, i
