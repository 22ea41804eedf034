"""GPU parity: the HIP path (through the C-ABI) against the CPU oracle on the same
seeded synthetic frames."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SEED = 0xCA4D10


def _oracle_batch(oracle, frames):
    res, cards = [], []
    for f in frames:
        r, c = oracle.scan_frame(f)
        res.append(r)
        cards.append(c)
    return res, np.stack(cards)


def test_synth_frames_match_oracle_generator(ctx, pkg, oracle):
    n = 3
    y = ctx.alloc(n * pkg.FRAME_BYTES)
    ctx.synth_frames(SEED, 5, n, y.ptr)
    ctx.synchronize()
    got = y.download(np.uint8).reshape(n, 480, 640)
    for i in range(n):
        want, _ = oracle.synth_frame(SEED, 5 + i)
        assert np.array_equal(got[i], want)
    y.free()


def test_full_pipeline_matches_oracle(ctx, pkg, oracle):
    n = 48
    y = ctx.alloc(n * pkg.FRAME_BYTES)
    res = ctx.alloc(n * 1024)
    cards = ctx.alloc(n * pkg.CARD_BYTES)
    ctx.synth_frames(SEED, 0, n, y.ptr)
    ctx.pipeline(y.ptr, n, res.ptr, cards.ptr)
    ctx.synchronize()
    got = res.download(pkg.RESULT_DTYPE, n)
    gcards = cards.download(np.uint8).reshape(n, 270, 428)
    frames = y.download(np.uint8).reshape(n, 480, 640)
    want, wcards = _oracle_batch(oracle, frames)
    near_ties = 0
    for i in range(n):
        g, w = got[i], want[i]
        # detection: integer / index results and the float bits of lines and corners
        assert np.array_equal(g["found"], w["found"]), i
        assert g["found_all"] == w["found_all"], i
        m = w["found"] != 0
        assert np.array_equal(g["rho"][m].view(np.uint32), w["rho"][m].view(np.uint32)), i
        assert np.array_equal(g["theta"][m].view(np.uint32), w["theta"][m].view(np.uint32)), i
        assert np.array_equal(g["corners"].view(np.uint32), w["corners"].view(np.uint32)), i
        # rectified card: byte exact
        assert np.array_equal(gcards[i], wcards[i]), "card %d: %d bytes differ" % (i, (gcards[i] != wcards[i]).sum())
        # scan: indices exact, float scores within 1e-4
        if g["vseg_y_offset"] != w["vseg_y_offset"] or g["pattern_type"] != w["pattern_type"]:
            # only acceptable as a float near-tie of the ORACLE'S two window sums; the later stages are then checked
            # against the oracle re-run at the device's segmentation
            from test_gpu_parity_large import prove_vseg_near_tie
            assert prove_vseg_near_tie(oracle, wcards[i], int(g["vseg_y_offset"]), int(g["pattern_type"]),
                                       int(w["vseg_y_offset"]), int(w["pattern_type"])), i
            near_ties += 1
            w = oracle.scan_card_image_at(wcards[i], int(g["vseg_y_offset"]), int(g["pattern_type"]),
                                          float(g["vseg_score"]), base=w)
        assert g["flags"] == w["flags"], (i, g["flags"], w["flags"])
        assert abs(float(g["vseg_score"]) - float(w["vseg_score"])) <= 1e-4
        assert g["n_offsets"] == w["n_offsets"]
        assert np.array_equal(g["offsets"], w["offsets"]), i
        assert g["pattern_offset"] == w["pattern_offset"]
        assert g["hseg_score"].view(np.uint32) == w["hseg_score"].view(np.uint32), i
        assert g["number_width"].view(np.uint32) == w["number_width"].view(np.uint32), i
        assert np.abs(g["scores"] - w["scores"]).max() <= 1e-4, i
        assert abs(float(g["number_score"]) - float(w["number_score"])) <= 1e-3
        assert np.array_equal(g["digits"], w["digits"]), i
    assert near_ties <= 1
    for b in (y, res, cards):
        b.free()
