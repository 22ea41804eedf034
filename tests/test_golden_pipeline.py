"""Committed golden vectors (tests/golden/pipeline_golden.npz, made by
tests/golden/make_pipeline_golden.py): the oracle must still reproduce them (CPU), and the
HIP path must reproduce them through the C-ABI (GPU)."""
import hashlib
import os

import numpy as np
import pytest

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "pipeline_golden.npz"))
SEED = int(G["seed"])


def _records(dtype):
    return G["records"].view(dtype).reshape(-1)


def test_oracle_reproduces_golden(oracle, orc):
    want = _records(orc.RESULT_DTYPE)
    for i in range(len(want)):
        y, d = oracle.synth_frame(SEED, i)
        assert np.array_equal(d, G["true_digits"][i])
        got, card = oracle.scan_frame(y)
        assert got.tobytes() == want[i].tobytes(), i
        assert oracle.scan_card_expiry(card, got).tobytes() == G["expiry"].view(orc.EXPIRY_DTYPE).reshape(-1)[i].tobytes(), i
        assert hashlib.sha256(card.tobytes()).hexdigest() == str(G["card_sha256"][i])
    assert np.array_equal(oracle.scan_frame(oracle.synth_frame(SEED, 0)[0])[1], G["card0"])


@pytest.mark.gpu
def test_hip_pipeline_reproduces_golden(ctx, pkg):
    want = _records(pkg.RESULT_DTYPE)
    n = len(want)
    y = ctx.alloc(n * pkg.FRAME_BYTES)
    res = ctx.alloc(n * 1024)
    cards = ctx.alloc(n * pkg.CARD_BYTES)
    exp = ctx.alloc(n * pkg.EXPIRY_DTYPE.itemsize)
    ctx.synth_frames(SEED, 0, n, y.ptr)
    ctx.pipeline_expiry(y.ptr, n, res.ptr, exp.ptr, cards.ptr)
    ctx.synchronize()
    got = res.download(pkg.RESULT_DTYPE, n)
    gexp = exp.download(pkg.EXPIRY_DTYPE, n)
    wexp = G["expiry"].view(pkg.EXPIRY_DTYPE).reshape(-1)
    gcards = cards.download(np.uint8).reshape(n, 270, 428)
    for i in range(n):
        assert hashlib.sha256(gcards[i].tobytes()).hexdigest() == str(G["card_sha256"][i]), i
        for f in ("found", "found_all", "flags", "vseg_y_offset", "pattern_type", "n_offsets", "offsets",
                  "pattern_offset", "digits"):
            assert np.array_equal(got[i][f], want[i][f]), (i, f)
        for f in ("rho", "theta", "corners", "hseg_score", "number_width"):
            assert np.array_equal(np.asarray(got[i][f]).view(np.uint32), np.asarray(want[i][f]).view(np.uint32)), (i, f)
        assert abs(float(got[i]["vseg_score"]) - float(want[i]["vseg_score"])) <= 1e-4
        assert np.abs(got[i]["scores"] - want[i]["scores"]).max() <= 1e-4
        # expiry: everything but the float scores is bit-exact
        a, b = gexp[i:i + 1].copy(), wexp[i:i + 1].copy()
        assert np.abs(a["groups"]["scores"] - b["groups"]["scores"]).max() <= 1e-4, i
        a["groups"]["scores"] = 0
        b["groups"]["scores"] = 0
        assert a.tobytes() == b.tobytes(), i
    for b in (y, res, cards, exp):
        b.free()
