"""Parity statistics on a larger seeded corpus (4096 frames by default): every integer/index output
must match the oracle; float scores within 1e-4.  An index may differ only where the reference's own
decision hangs on a float comparison closer than the score tolerance, and every such event must be
PROVEN from the oracle's values (no blanket allowance):
  * a digit label: the oracle's top two vote scores of that digit are within 2e-4 (|delta| <= 1e-4 on
    every class and a different arg-max imply a top-two gap <= 2e-4);
  * the usable flag: the oracle's number_score is within 1e-3 of the gate value 3 (160 scores summed);
  * y_offset / pattern: the oracle's OWN window sums of the two (offset, pattern) choices are within 1e-4 -- in its final
    scores, or in its coarse pass with the fine pass re-run from the other coarse choice (prove_vseg_near_tie).
    A proven tie does not end the frame's check: the oracle's later stages (hseg, digit models, expiry) are re-run at the
    device's segmentation and compared like everything else, so a tie cannot hide a second difference.
  * the vseg gate (score > 15): the oracle's own score is within 1e-4 of 15 and of the device's score; the later stages are
    re-run with the device's score at the gate, as for a segmentation tie.
Anything else counts as `unexplained` and fails the test.
The oracle runs on a thread pool (its C code holds no shared mutable state; ctypes releases the GIL)."""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SEED = int(os.environ.get("DMZ_PARITY_SEED", "31337"))
# DMZ_PARITY_FLAVOUR=1: the sweep with the device AND the oracle in the stock x86-64 (SSE2) order of the homography
FLAVOUR = int(os.environ.get("DMZ_PARITY_FLAVOUR", "0"))


def prove_vseg_near_tie(oracle, card, gy, gp, wy, wp):
    """Is the device's (y_offset, pattern) = (gy, gp) a float near-tie of the oracle's (wy, wp)?  Proven from the ORACLE'S
    own scores only, in one of the two ways n_vseg.cpp:94-168 can come out differently under |delta| <= 1e-4 scores:
      (a) fine tie: the oracle's final window sums at the two choices are within 1e-4;
      (b) coarse tie: the coarse pass (every fourth row) has another window sum within 1e-4 of its maximum, and the fine
          pass run from THAT coarse choice (other rows evaluated, n_vseg.cpp:140-152) ends at the device's choice -- exactly or
          by a fine tie as in (a)."""
    def wsum(vis, amx, yy, p):
        return float((vis if p == 1 else amx)[yy:yy + 27].astype(np.float64).sum()) if p else 0.0

    _, _, _, vis, amx = oracle.best_n_vseg(card)
    if abs(wsum(vis, amx, gy, gp) - wsum(vis, amx, wy, wp)) < 1e-4:
        return True
    cv, ca = np.zeros(270, np.float32), np.zeros(270, np.float32)
    for yy in range(0, 270, 4):
        p3 = oracle.applym_vseg(oracle.vseg_row_features(card[yy, 10:418]))
        cv[yy], ca[yy] = p3[1], p3[2]
    sums = [(wsum(cv, ca, yy, p), yy, p) for yy in range(0, 244) for p in (1, 2)]
    top = max(t[0] for t in sums)
    for sc, yc, pc in sums:
        if top - sc >= 1e-4:
            continue
        fv, fa = cv.copy(), ca.copy()
        for yy in range(max(0, yc - 8), min(270, yc + 27 + 8)):
            if fv[yy] == 0 and fa[yy] == 0:
                p3 = oracle.applym_vseg(oracle.vseg_row_features(card[yy, 10:418]))
                fv[yy], fa[yy] = p3[1], p3[2]
        _, y2, p2 = oracle.best_segmentation_for_vseg_scores(fv, fa)
        if (y2, p2) == (gy, gp) or abs(wsum(fv, fa, gy, gp) - wsum(fv, fa, y2, p2)) < 1e-4:
            return True
    return False


def new_stats():
    """the counters _compare_frame fills"""
    return dict(card_bytes_diff=0, det_diff=0, ties=0, idx_diff=0, max_score_err=0.0, max_vseg_err=0.0,
                label_diff=0, flag_diff=0, unexplained=0, expiry_frames=0, expiry_groups=0, expiry_seg_diff=0,
                expiry_slash_flips=0, max_expiry_err=0.0)


def compare_sample_with_oracle(ctx, pkg, oracle, indices, frame_of, rec, expv, cards_buf, scan=None):
    """the frames `indices` of a batch whose records (rec, expv) are on the host and whose cards are resident in cards_buf
    (None: no card bytes), through the same frame-by-frame comparison as the sweeps: detect bits (rho / theta / corners),
    card bytes, segmentation indices, hseg_score bits, scores, labels, expiry.  frame_of(i) -> the frame (or, with
    scan = "cards", the pre-warped crop) the oracle scans.  Returns the counters."""
    stats = new_stats()
    for i in indices:
        i = int(i)
        if scan == "cards":
            wcard = frame_of(i)
            w = oracle.scan_card_image(wcard, warped=False)
            g = rec[i].copy()
            for f in ("found", "found_all", "rho", "theta", "corners"):  # a crop has no detection record
                g[f] = w[f]
        else:
            w, wcard = oracle.scan_frame(frame_of(i))
            g = rec[i]
        we = oracle.scan_card_expiry(wcard, w)
        if cards_buf is not None:
            gcard = np.empty(pkg.CARD_BYTES, np.uint8)
            ctx._check(ctx.lib.dmz_hip_memcpy_d2h(ctx.h, gcard.ctypes.data, cards_buf.ptr + i * pkg.CARD_BYTES, gcard.nbytes))
            gcard = gcard.reshape(270, 428)
        else:
            gcard = wcard
        _compare_frame(pkg, oracle, stats, g, expv[i], gcard, w, wcard, we, where=i)
    return stats


def compare_with_oracle(ctx, pkg, oracle, y, n):
    """Full pipeline on the n frames resident in `y` against the oracle, frame by frame; returns the counters."""
    res = ctx.alloc(n * 1024)
    cards = ctx.alloc(n * pkg.CARD_BYTES)
    exp = ctx.alloc(n * pkg.EXPIRY_DTYPE.itemsize)
    ctx.pipeline_expiry(y.ptr, n, res.ptr, exp.ptr, cards.ptr)
    ctx.synchronize()
    got = res.download(pkg.RESULT_DTYPE, n)
    gexp = exp.download(pkg.EXPIRY_DTYPE, n)
    stats = dict(found_all=int(got["found_all"].astype(bool).sum()),
                 vseg_ok=int(((got["flags"] & pkg.FLAG_VSEG_OK) != 0).sum()) if hasattr(pkg, "FLAG_VSEG_OK") else -1,
                 # frames the 15-digit pattern won (n_vseg.cpp:26-30): the second instantiation of the hseg score, a 15-digit categorise
                 amex_like=int((((got["flags"] & pkg.FLAG_VSEG_OK) != 0) & (got["pattern_type"] == 2)).sum()),
                 **new_stats())
    nthreads = max(1, min(64, (os.cpu_count() or 2) // 2))

    def slice_to_host(buf, first, count, item_bytes):  # frames / cards of one chunk (a sweep's whole batch is tens of GB)
        out = np.empty(count * item_bytes, np.uint8)
        ctx._check(ctx.lib.dmz_hip_memcpy_d2h(ctx.h, out.ctypes.data, buf.ptr + first * item_bytes, out.nbytes))
        return out

    chunk = 4096
    with ThreadPoolExecutor(nthreads) as pool:
        for c0 in range(0, n, chunk):
            cn = min(chunk, n - c0)
            frames = slice_to_host(y, c0, cn, pkg.FRAME_BYTES).reshape(cn, 480, 640)
            gcards = slice_to_host(cards, c0, cn, pkg.CARD_BYTES).reshape(cn, 270, 428)

            def oracle_frame(j):
                w, wcard = oracle.scan_frame(frames[j])
                return w, wcard, oracle.scan_card_expiry(wcard, w)

            oracle_out = list(pool.map(oracle_frame, range(cn), chunksize=8))
            for j in range(cn):
                _compare_frame(pkg, oracle, stats, got[c0 + j], gexp[c0 + j], gcards[j], *oracle_out[j], where=c0 + j)
    for b in (res, cards, exp):
        b.free()
    return stats


def _compare_frame(pkg, oracle, stats, g, ge, gcard, w, wcard, we, where=None):
    """one frame's device records against the oracle's; counts into stats (`where`: the frame's index, for the report of a
    difference)"""
    before = {k: stats[k] for k in ("det_diff", "idx_diff", "unexplained", "expiry_seg_diff", "expiry_slash_flips")}
    _compare_frame_body(pkg, oracle, stats, g, ge, gcard, w, wcard, we)
    for k, v in before.items():
        if stats[k] != v and len(stats.setdefault("_events", [])) < 12:
            stats["_events"].append((k, where))
            print("\n%s at frame %s: device y_offset %d pattern %d offsets %s pattern_offset %d hseg_score %r number_width %r | oracle "
                  "y_offset %d pattern %d offsets %s pattern_offset %d hseg_score %r number_width %r"
                  % (k, where, g["vseg_y_offset"], g["pattern_type"], g["offsets"].tolist(), g["pattern_offset"],
                     float(g["hseg_score"]), float(g["number_width"]), w["vseg_y_offset"], w["pattern_type"], w["offsets"].tolist(),
                     w["pattern_offset"], float(w["hseg_score"]), float(w["number_width"])))


def _compare_frame_body(pkg, oracle, stats, g, ge, gcard, w, wcard, we):
    for _ in (0,):  # (a one-pass loop: `continue` ends the frame's check as in the original flat loop)
        fmask = w["found"] != 0  # rho / theta bits of every edge that was found, also on frames with fewer than four
        if not (np.array_equal(g["found"], w["found"]) and g["found_all"] == w["found_all"]
                and np.array_equal(g["rho"].view(np.uint32)[fmask], w["rho"].view(np.uint32)[fmask])
                and np.array_equal(g["theta"].view(np.uint32)[fmask], w["theta"].view(np.uint32)[fmask])
                and np.array_equal(g["corners"].view(np.uint32), w["corners"].view(np.uint32))):
            stats["det_diff"] += 1
            continue
        stats["card_bytes_diff"] += int((gcard != wcard).sum())
        if g["vseg_y_offset"] != w["vseg_y_offset"] or g["pattern_type"] != w["pattern_type"]:
            near = prove_vseg_near_tie(oracle, wcard, int(g["vseg_y_offset"]), int(g["pattern_type"]),
                                       int(w["vseg_y_offset"]), int(w["pattern_type"]))
            if not near:
                stats["unexplained"] += 1
                continue
            stats["ties"] += 1
            # downstream of the tie: the oracle's later stages at the device's segmentation
            w = oracle.scan_card_image_at(wcard, int(g["vseg_y_offset"]), int(g["pattern_type"]), float(g["vseg_score"]),
                                          base=w)
            we = oracle.scan_card_expiry(wcard, w)
        stats["max_vseg_err"] = max(stats["max_vseg_err"], abs(float(g["vseg_score"]) - float(w["vseg_score"])))
        if (g["flags"] ^ w["flags"]) & pkg.FLAG_VSEG_OK:
            # the vseg gate compares the score with 15 (frame.cpp:38, kMinVSegScore): only a float near-tie may flip it -- the
            # oracle's OWN score within the score tolerance of the gate value (round 6: first seen on 1 of 524 288 frames,
            # 14.999999 against 15.000001).  Proven, counted with the segmentation ties, and not the end of the frame's check:
            # the oracle's later stages are re-run with the device's score at the gate.
            if not (abs(float(w["vseg_score"]) - 15.0) < 1e-4 and abs(float(g["vseg_score"]) - float(w["vseg_score"])) <= 1e-4):
                stats["unexplained"] += 1
                continue
            stats["ties"] += 1
            w = oracle.scan_card_image_at(wcard, int(g["vseg_y_offset"]), int(g["pattern_type"]), float(g["vseg_score"]), base=w)
            we = oracle.scan_card_expiry(wcard, w)
        if not (np.array_equal(g["offsets"], w["offsets"]) and g["pattern_offset"] == w["pattern_offset"]
                and g["hseg_score"].view(np.uint32) == w["hseg_score"].view(np.uint32)):
            stats["idx_diff"] += 1
            continue
        stats["max_score_err"] = max(stats["max_score_err"], float(np.abs(g["scores"] - w["scores"]).max()))
        for d in np.nonzero(g["digits"] != w["digits"])[0]:
            top2 = np.sort(w["scores"][d])[-2:]
            stats["label_diff" if float(top2[1] - top2[0]) <= 2e-4 else "unexplained"] += 1
        if g["flags"] != w["flags"]:
            # the usable gate compares number_score with 3: only a float near-tie may flip it
            near = (g["flags"] ^ w["flags"]) == pkg.FLAG_USABLE and abs(float(w["number_score"]) - 3.0) < 1e-3
            stats["flag_diff" if near else "unexplained"] += 1
            if not near:
                continue
            # a proven tie of the usable gate does not end the frame's check: the oracle's expiry stage at the device's flags
            # (scan.cpp:57-64: the expiry digits are categorised for usable frames only)
            w = w.copy()
            w["flags"] = g["flags"]
            we = oracle.scan_card_expiry(wcard, w)
        # ---- expiry: stripes, groups and rects exact; scores 1e-4 ----
        ns = int(we["n_stripes"])
        if not (ge["n_stripes"] == ns and np.array_equal(ge["stripe_base_row"][:ns], we["stripe_base_row"][:ns])
                and np.array_equal(ge["stripe_sum"][:ns], we["stripe_sum"][:ns])
                and ge["categorised"] == we["categorised"]):
            stats["expiry_seg_diff"] += 1
            continue
        if ge["n_found"] != we["n_found"]:
            stats["expiry_slash_flips"] += 1  # only a slash probability within float noise of 0.7 may do this
            continue
        k = int(we["n_groups"])
        stats["expiry_frames"] += int(k > 0)
        stats["expiry_groups"] += k
        a, b = ge["groups"][:k], we["groups"][:k]
        same = all(np.array_equal(a[f], b[f]) for f in ("top", "left", "width", "height", "char_top", "char_left",
                                                        "stripe_base_row"))
        if not same:
            stats["expiry_seg_diff"] += 1
            continue
        if k:
            stats["max_expiry_err"] = max(stats["max_expiry_err"], float(np.abs(a["scores"] - b["scores"]).max()))


def assert_parity(stats, n, min_expiry_frames=0, max_ties=2):
    """the bars of DESIGN.md section 4 on one run's counters"""
    assert stats["det_diff"] == 0 and stats["card_bytes_diff"] == 0 and stats["idx_diff"] == 0, stats
    assert stats["max_score_err"] <= 1e-4 and stats["max_vseg_err"] <= 1e-4, stats
    # near-tie events are proven one by one in _compare_frame; anything unproven fails
    assert stats["unexplained"] == 0, stats
    # ... and proven near-ties of two votes / of the usable gate are rare events (measured: <= 1 per 65 536 corpus frames);
    # a regression that produced many "provable" ones must not pass
    assert stats["label_diff"] <= max(2, n // 2048) and stats["flag_diff"] <= max(2, n // 2048), stats
    # ... and so are proven near-ties of two segmentations: none in 3 x 131 072 corpus frames, one per ~450 fuzz frames
    # (garbage cards have flat vseg scores; the fuzz test passes its own bound)
    # (round 6: the proven near-ties of the vseg GATE count here too -- measured 1 - 2 per 524 288 corpus frames together -- so
    # the bound grows with the sweep: 2 up to 262 144 frames, n / 131 072 beyond)
    assert stats["ties"] <= max(max_ties, n // 131072), stats
    assert stats["expiry_seg_diff"] == 0 and stats["expiry_slash_flips"] == 0 and stats["max_expiry_err"] <= 1e-4, stats
    assert stats["expiry_frames"] >= min_expiry_frames, stats


def _with_flavour(ctx, oracle, fn):
    if not FLAVOUR:
        return fn()
    ctx.set_reference_flavour(FLAVOUR)
    oracle.set_reference_flavour(FLAVOUR)
    try:
        return fn()
    finally:
        ctx.set_reference_flavour(0)
        oracle.set_reference_flavour(0)


def test_1024_frames_against_oracle(ctx, pkg, oracle):
    n = int(os.environ.get("DMZ_PARITY_FRAMES", "4096"))  # raise for a one-off sweep (the oracle does ~215 frames/s per core)
    y = ctx.alloc(n * pkg.FRAME_BYTES)
    ctx.synth_frames(SEED, 1000, n, y.ptr)
    stats = _with_flavour(ctx, oracle, lambda: compare_with_oracle(ctx, pkg, oracle, y, n))
    y.free()
    print("parity stats%s over %d frames: %s" % (" (SSE2 flavour)" if FLAVOUR else "", n, stats))
    assert_parity(stats, n, min_expiry_frames=n // 4)


def _fuzz_frame(rng, kind, oracle):
    """Frames the synthetic corpus never produces: no clean card, or a card in the wrong place."""
    yy, xx = np.mgrid[0:480, 0:640]
    if kind >= 7:    # a card frame posterised to a few grey levels: flat regions and plateaus -> runs of EQUAL Scharr window
        # sums and stripe sums, i.e. the candidate order of the expiry segmentation (std::sort's tie order) decides picks
        f = oracle.synth_frame(SEED + 98, int(rng.integers(0, 1 << 20)))[0].astype(np.int64)
        step = int(rng.choice([8, 16, 32, 64]))
        f = (f // step) * step
        if kind == 8:  # ... and a periodic column pattern over the lower half of the card (period 3: all 9-px windows tie)
            per = int(rng.choice([2, 3, 4, 6]))
            f[300:372, 110:530] += np.where((xx[300:372, 110:530] % per) == 0, int(rng.integers(8, 40)), 0)
        return np.clip(f, 0, 255).astype(np.uint8)
    if kind >= 5:    # a synthetic card frame shifted off the guide frame, dimmed, with heavy noise
        f = oracle.synth_frame(SEED + 99, int(rng.integers(0, 1 << 20)))[0].astype(np.int64)
        f = np.roll(f, (int(rng.integers(-14, 15)), int(rng.integers(-14, 15))), axis=(0, 1))
        if kind == 6:
            f = f * rng.uniform(0.35, 0.9) + rng.integers(0, int(rng.integers(2, 60)), (480, 640))
        return np.clip(f, 0, 255).astype(np.uint8)
    if kind == 0:    # uniform noise
        f = rng.integers(0, 256, (480, 640))
    elif kind == 1:  # low-contrast noise on a ramp
        f = (xx * rng.uniform(0, 0.3) + yy * rng.uniform(0, 0.3) + rng.integers(0, 40, (480, 640))) % 256
    elif kind == 2:  # a bright rotated quadrilateral anywhere, noisy background
        f = rng.integers(0, 60, (480, 640))
        a = rng.uniform(-0.5, 0.5)
        cx, cy = rng.uniform(200, 440), rng.uniform(150, 330)
        u = (xx - cx) * np.cos(a) + (yy - cy) * np.sin(a)
        v = -(xx - cx) * np.sin(a) + (yy - cy) * np.cos(a)
        inside = (np.abs(u) < rng.uniform(120, 260)) & (np.abs(v) < rng.uniform(80, 170))
        f = np.where(inside, 150 + rng.integers(0, 100, (480, 640)), f)
    elif kind == 3:  # bars and a grid: many strong collinear edges, saturated Sobel responses
        p = int(rng.integers(6, 40))
        f = np.where(((xx // p) + (yy // int(rng.integers(6, 40)))) % 2 == 0, 255, 0)
    else:            # the guide-frame rectangle drawn as thin lines, plus clutter lines
        f = rng.integers(0, 30, (480, 640))
        for _ in range(int(rng.integers(3, 12))):
            if rng.integers(0, 2):
                f[int(rng.integers(0, 480)), :] = 255
            else:
                f[:, int(rng.integers(0, 640))] = 255
        f[105:375, 106] = 255; f[105:375, 533] = 255; f[105, 106:534] = 255; f[374, 106:534] = 255
    return f.astype(np.uint8)


def test_fuzz_frames_against_oracle(ctx, pkg, oracle):
    """Noise, ramps, stray quadrilaterals, bar patterns and line clutter through the whole pipeline:
    detect with thousands of NMS survivors, edges that do not meet, quads that leave the frame, garbage
    for the scan stages -- every record must still equal the oracle's."""
    n = int(os.environ.get("DMZ_FUZZ_FRAMES", "270"))
    rng = np.random.default_rng(int(os.environ.get("DMZ_PARITY_SEED", "31337")) + 7)
    frames = np.stack([_fuzz_frame(rng, i % 9, oracle) for i in range(n)])
    y = ctx.alloc(frames.nbytes).upload(frames)
    stats = _with_flavour(ctx, oracle, lambda: compare_with_oracle(ctx, pkg, oracle, y, n))
    y.free()
    print("fuzz parity stats%s over %d frames: %s" % (" (SSE2 flavour)" if FLAVOUR else "", n, stats))
    # garbage cards have flat vseg scores: near-ties (each one proven within 1e-4) are more frequent here -- measured 74 - 75
    # per 33 600 frames (profiles/r4_*sweep*.log); the bound is about twice that rate
    assert_parity(stats, n, max_ties=max(2, n // 256))
