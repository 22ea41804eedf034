"""GPU parity on the 15-digit (4-6-5) number layout: n_vseg.cpp:26-30's second pattern, the second instantiation of the
hseg score (hseg.hip), a 15-digit categorise (digits.hip).  The synthetic corpus draws 16-digit cards only, and random
fuzz frames never reach VSEG_OK with that pattern, so the cards are built here: the digit boxes of a synthetic card,
found by the oracle, are pasted back in the 4-6-5 arrangement at the card's own pitch."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SEED = 0xA3E7


def amex_like_card(oracle, idx):
    card, _ = oracle.synth_card(SEED, idx)
    r = oracle.scan_card_image(card, warped=False)
    if not (r["flags"] & 4) or r["pattern_type"] != 1:
        return None
    y = int(r["vseg_y_offset"])
    offs = [int(x) for x in r["offsets"][:16]]
    pitch = float(r["number_width"])
    out = card.copy()
    band = slice(max(0, y - 2), min(270, y + 29))
    bg = np.median(card[max(0, y - 14):max(1, y - 4)], axis=0).astype(np.uint8)
    out[band] = bg[None, :]
    slots = [0, 1, 2, 3, 5, 6, 7, 8, 9, 10, 12, 13, 14, 15, 16]
    for j, sl in enumerate(slots):
        x = int(round(offs[0] + sl * pitch))
        if x + 19 > 428:
            return None
        out[band, x:x + 19] = card[band, offs[j]:offs[j] + 19]
    return out


def test_fifteen_digit_layout_against_oracle(ctx, pkg, oracle):
    cards = [c for c in (amex_like_card(oracle, i) for i in range(160)) if c is not None]
    arr = np.ascontiguousarray(np.stack(cards))
    n = len(cards)
    res = np.zeros(n, pkg.RESULT_DTYPE)
    exp = np.zeros(n, pkg.EXPIRY_DTYPE)
    ctx.scan_cards(arr, n, res)
    ctx.scan_expiry(arr, n, res, exp)
    fifteen = usable = 0
    for i in range(n):
        w = oracle.scan_card_image(arr[i], warped=False)
        g = res[i]
        assert (g["flags"] & 7) == (w["flags"] & 7), i
        assert g["vseg_y_offset"] == w["vseg_y_offset"] and g["pattern_type"] == w["pattern_type"], i
        assert abs(float(g["vseg_score"]) - float(w["vseg_score"])) <= 1e-4, i
        assert g["n_offsets"] == w["n_offsets"] and np.array_equal(g["offsets"], w["offsets"]), i
        assert g["pattern_offset"] == w["pattern_offset"], i
        assert g["hseg_score"].view(np.uint32) == w["hseg_score"].view(np.uint32), i
        assert g["number_width"].view(np.uint32) == w["number_width"].view(np.uint32), i
        assert np.abs(g["scores"] - w["scores"]).max() <= 1e-4, i
        if not np.array_equal(g["digits"], w["digits"]):  # only a near-tie of two votes may differ
            for d in np.nonzero(g["digits"] != w["digits"])[0]:
                top2 = np.sort(w["scores"][d])[-2:]
                assert float(top2[1] - top2[0]) <= 2e-4, (i, d)
        assert abs(float(g["number_score"]) - float(w["number_score"])) <= 1e-4, i
        we = oracle.scan_card_expiry(arr[i], w)
        assert exp[i]["n_stripes"] == we["n_stripes"] and exp[i]["n_found"] == we["n_found"], i
        if w["flags"] & 4 and w["pattern_type"] == 2:
            fifteen += 1
            assert w["n_offsets"] == 15
            usable += int(bool(w["flags"] & 1))
    print("cards with the 15-digit pattern: %d of %d (usable: %d)" % (fifteen, n, usable))
    assert fifteen >= 20 and usable >= 4
