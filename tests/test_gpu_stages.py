"""GPU parity per stage, through the C-ABI: model KATs, homography, warp with the oracle's
matrix, scan of pre-warped cards, edge cases (blank / upside-down / undetectable frames,
host-pointer staging, truncated corners)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
KATS = np.load(os.path.join(os.path.dirname(__file__), "golden", "model_kats.npz"))
SEED = 4242


def test_model_kats_on_device(ctx):
    out = ctx.apply_vseg_model(KATS["vseg_in"])[0]
    assert np.abs(out - KATS["vseg_out"]).max() <= 1e-5
    for idx, name in enumerate(("5c241121", "01266c1b", "b00bf70c")):
        out = ctx.apply_digit_model(idx, KATS["digit_%s_in" % name])[0]
        assert np.abs(out - KATS["digit_%s_out" % name]).max() <= 1e-5, name


def test_models_match_oracle_on_random_batches(ctx, oracle):
    rng = np.random.default_rng(0)
    x = rng.uniform(0, 1, (64, 204)).astype(np.float32)
    got = ctx.apply_vseg_model(x)
    want = np.stack([oracle.applym_vseg(r) for r in x])
    assert np.abs(got - want).max() <= 1e-5
    d = rng.uniform(0, 1, (32, 513)).astype(np.float32)
    for m in range(3):
        got = ctx.apply_digit_model(m, d)
        want = np.stack([oracle.applyc_digit(m, r) for r in d])
        assert np.abs(got - want).max() <= 1e-5


def test_homography_bit_exact(ctx, oracle):
    rng = np.random.default_rng(1)
    dst = np.array([0, 0, 427, 0, 0, 269, 427, 269], np.float32)
    base = np.array([106, 105, 533, 105, 106, 374, 533, 374], np.float32)
    for _ in range(200):
        src = base + rng.uniform(-12, 12, 8).astype(np.float32)
        a, b = ctx.calc_persp_transform(src, dst), oracle.calc_persp_transform(src, dst)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_homography_in_the_sse2_order_of_a_stock_x86_build(ctx, pkg, oracle):
    """DMZ_HIP_OPT_EIGEN_SSE2 / dmz_hip_set_reference_flavour(1): the device's Householder QR in the summation order of the
    reference as a stock x86-64 build compiles it (the oracle's second order is pinned on that build,
    tests/test_oracle_vs_ref.py) -- single homographies through dmz_hip_calc_persp_transform, and the whole chain: the
    pipeline's cards equal the oracle's warp with the SSE2-order matrix, and differ from the default-order cards."""
    rng = np.random.default_rng(41)
    dst = np.array([0, 0, 427, 0, 0, 269, 427, 269], np.float32)
    base = np.array([106, 105, 533, 105, 106, 374, 533, 374], np.float32)
    try:
        ctx.set_reference_flavour(1)
        for k in range(300):
            src = base + rng.uniform(-12, 12, 8).astype(np.float32)
            got = ctx.calc_persp_transform(src, dst)
            want = oracle.calc_persp_transform(src, dst, sse=True)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (k, got, want)
        n = 48
        y = ctx.alloc(n * pkg.FRAME_BYTES)
        res, cards = ctx.alloc(n * 1024), ctx.alloc(n * pkg.CARD_BYTES)
        ctx.synth_frames(SEED + 3, 0, n, y.ptr)
        ctx.pipeline(y.ptr, n, res.ptr, cards.ptr)
        ctx.synchronize()
        sse_cards = cards.download(np.uint8).reshape(n, 270, 428)
        got = res.download(pkg.RESULT_DTYPE, n)
        # the per-call override (DMZ_HIP_OPT_EIGEN_SCALAR): one call in the scalar order while the context default is SSE2
        ctx.pipeline(y.ptr, n, res.ptr, cards.ptr, options=pkg.OPT_EIGEN_SCALAR)
        ctx.synchronize()
        scalar_call_cards = cards.download(np.uint8).reshape(n, 270, 428)
    finally:
        ctx.set_reference_flavour(0)
    ctx.pipeline(y.ptr, n, res.ptr, cards.ptr)
    ctx.synchronize()
    default_cards = cards.download(np.uint8).reshape(n, 270, 428)
    assert np.array_equal(scalar_call_cards, default_cards)
    ctx.pipeline(y.ptr, n, res.ptr, cards.ptr, options=pkg.OPT_EIGEN_SSE2)  # and the other way round
    ctx.synchronize()
    assert np.array_equal(cards.download(np.uint8).reshape(n, 270, 428), sse_cards)
    assert not (res.download(pkg.RESULT_DTYPE, n)["flags"] & pkg.FLAG_FAULT).any()
    frames = y.download(np.uint8).reshape(n, 480, 640)
    moved = 0
    for i in range(n):
        c = got[i]["corners"].astype(np.float32)
        src = np.array([c[0], c[1], c[4], c[5], c[2], c[3], c[6], c[7]], np.float32)  # tl, tr, bl, br (dmz.cpp:446-471)
        assert np.array_equal(sse_cards[i], oracle.warp_perspective(frames[i], oracle.calc_persp_transform(src, dst, sse=True))), i
        assert np.array_equal(default_cards[i], oracle.warp_perspective(frames[i], oracle.calc_persp_transform(src, dst))), i
        moved += int((sse_cards[i] != default_cards[i]).any())
    assert moved > n // 4
    for b in (y, res, cards):
        b.free()


def test_warp_with_given_matrices_byte_exact(ctx, pkg, oracle):
    """SURVEY Appendix B layering (ii): feed the oracle's float[9] to the HIP warp."""
    rng = np.random.default_rng(2)
    n = 6
    frames = np.stack([oracle.synth_frame(SEED, i)[0] for i in range(n)])
    dst = np.array([0, 0, 427, 0, 0, 269, 427, 269], np.float32)
    base = np.array([106, 105, 533, 105, 106, 374, 533, 374], np.float32)
    mats = []
    for i in range(n):
        jitter = rng.uniform(-30, 30, 8) if i >= 4 else rng.uniform(-6, 6, 8)  # i>=4: partly outside
        if i == 5:
            jitter += np.array([-150, -130] * 4)  # card hangs over the frame's top-left corner
        mats.append(oracle.calc_persp_transform(base + jitter.astype(np.float32), dst))
    mats = np.stack(mats)
    cards = np.zeros((n, 270, 428), np.uint8)
    ctx.warp_perspective(frames, n, mats, cards)  # host pointers: staged through HBM
    for i in range(n):
        want = oracle.warp_perspective(frames[i], mats[i])
        assert np.array_equal(cards[i], want), (i, int((cards[i] != want).sum()))


def test_warp_generic_path_matrices_byte_exact(ctx, pkg, oracle):
    """Quads whose strips do not fit the staged window (rotation, zoom, a side collapsed towards a
    point, a self-crossing quad whose W changes sign inside the card) take k_warp's generic path;
    tiny far-away quads leave the card mostly outside the frame (BORDER_CONSTANT 0)."""
    frame = oracle.synth_frame(SEED, 3)[0]
    dst = np.array([0, 0, 427, 0, 0, 269, 427, 269], np.float32)
    cx, cy = 320.0, 240.0

    def quad(angle_deg, sx, sy, skew=(0, 0, 0, 0, 0, 0, 0, 0)):
        a = np.deg2rad(angle_deg)
        pts = []
        for ux, uy in ((-1, -1), (1, -1), (-1, 1), (1, 1)):  # tl, tr, bl, br
            x, y = ux * sx, uy * sy
            pts += [cx + x * np.cos(a) - y * np.sin(a), cy + x * np.sin(a) + y * np.cos(a)]
        return (np.array(pts) + np.array(skew)).astype(np.float32)

    quads = [
        quad(30, 213, 134), quad(90, 213, 134), quad(-135, 150, 100), quad(7, 213, 134),
        quad(0, 330, 250),                       # zoom out: 1.55 source px per card px
        quad(0, 600, 400),                       # card far larger than the frame
        quad(0, 20, 12),                         # strong zoom in
        quad(0, 213, 134, (0, 0, -400, 0, 0, 0, 0, 0)),      # top edge collapsed towards its left end
        quad(0, 213, 134, (430, 0, -430, 0, 0, 0, 0, 0)),    # tl/tr swapped: self-crossing quad
        quad(15, 213, 134, (-300, -260, -300, -260, -300, -260, -300, -260)),  # rotated and mostly outside
    ]
    mats = np.stack([oracle.calc_persp_transform(q, dst) for q in quads])
    n = len(quads)
    frames = np.repeat(frame[None], n, axis=0)
    cards = np.full((n, 270, 428), 0xA5, np.uint8)
    ctx.warp_perspective(frames, n, mats, cards)
    for i in range(n):
        want = oracle.warp_perspective(frame, mats[i])
        assert np.array_equal(cards[i], want), (i, int((cards[i] != want).sum()))


def test_warp_perspective_strength_sweep_byte_exact(ctx, pkg, oracle):
    """k_warp's cheap coordinate forms are chosen per strip from rho = |M7| / |W|, the relative change of the projective
    denominator per card row (warp.hip, k_warp_windows): 1 / W linear in the row below 2^-24, the coordinates affine in the
    extrapolated reciprocal up to 2^-11, the exact sequence beyond.  Trapezoids whose top edge is inset by 10^-5 .. 60 px
    walk rho through all of it (and across both thresholds), with shear / rotation / sub-pixel offsets on top, over noise
    frames (a coordinate that is off by 1/32 px changes the byte)."""
    rng = np.random.default_rng(77)
    dst = np.array([0, 0, 427, 0, 0, 269, 427, 269], np.float32)
    insets = np.concatenate([[0.0], np.logspace(-5, np.log10(60.0), 47)])
    quads = []
    for k, e in enumerate(insets):
        x0, y0 = 106 + rng.uniform(-20, 20), 105 + rng.uniform(-20, 20)
        w, h = 427 * rng.uniform(0.8, 1.2), 269 * rng.uniform(0.8, 1.2)
        q = np.array([x0 + e, y0, x0 + w - e, y0, x0, y0 + h, x0 + w, y0 + h])  # tl, tr, bl, br
        if k % 3 == 1:   # shear
            q[[0, 2]] += rng.uniform(-25, 25)
        if k % 3 == 2:   # small rotation about the quad's centre
            a = np.deg2rad(rng.uniform(-6, 6))
            c = q.reshape(4, 2) - [x0 + w / 2, y0 + h / 2]
            q = (c @ np.array([[np.cos(a), np.sin(a)], [-np.sin(a), np.cos(a)]]) + [x0 + w / 2, y0 + h / 2]).reshape(8)
        if k % 4 == 3:   # the perspective across the card instead of down it
            q = np.array([x0, y0 + e, x0 + w, y0, x0, y0 + h - e, x0 + w, y0 + h])
        quads.append(q.astype(np.float32))
    n = len(quads)
    mats = np.stack([oracle.calc_persp_transform(q, dst) for q in quads])
    frames = rng.integers(0, 256, (n, 480, 640), dtype=np.uint8)
    cards = np.full((n, 270, 428), 0xA5, np.uint8)
    ctx.warp_perspective(frames, n, mats, cards)
    for i in range(n):
        want = oracle.warp_perspective(frames[i], mats[i])
        assert np.array_equal(cards[i], want), (i, float(insets[i]), int((cards[i] != want).sum()))
    # The same sweep on a WIDE plane: source x of ~7 600 .. 8 100 px puts |fX| at 2^18 units of 1/32 px, where
    # the affine form's error budget no longer holds near rho = 2^-11 -- k_warp_windows checks the budget per strip and such
    # strips take the exact loop; the bytes must not care which form ran.
    W, H = 8192, 400
    shift = np.array([7600, -40] * 4, np.float32)
    mats_w = np.stack([oracle.calc_persp_transform(q + shift, dst) for q in quads])
    plane = rng.integers(0, 256, (n, H, W), dtype=np.uint8)
    cards_w = np.full((n, 270, 428), 0x5A, np.uint8)
    ctx.warp_perspective(plane, n, mats_w, cards_w, width=W, height=H)
    for i in range(n):
        want = oracle.warp_perspective(plane[i], mats_w[i])
        assert np.array_equal(cards_w[i], want), ("wide", i, float(insets[i]), int((cards_w[i] != want).sum()))


def test_scan_prewarped_cards(ctx, pkg, oracle):
    """BASELINE configs[2]: vseg/hseg/categorise on pre-warped 428x270 crops."""
    n = 40
    dcards = ctx.alloc(n * pkg.CARD_BYTES)
    res = ctx.alloc(n * 1024)
    res.upload(np.zeros(n * 1024, np.uint8))
    ctx.synth_cards(SEED, 0, n, dcards.ptr)
    ctx.scan_cards(dcards.ptr, n, res.ptr)
    ctx.synchronize()
    got = res.download(pkg.RESULT_DTYPE, n)
    cards = dcards.download(np.uint8).reshape(n, 270, 428)
    for i in range(n):
        ref_card, _ = oracle.synth_card(SEED, i)
        assert np.array_equal(cards[i], ref_card)
        w = oracle.scan_card_image(cards[i], warped=False)
        g = got[i]
        assert g["vseg_y_offset"] == w["vseg_y_offset"] and g["pattern_type"] == w["pattern_type"], i
        assert (g["flags"] & 7) == (w["flags"] & 7), i
        assert np.array_equal(g["offsets"], w["offsets"]) and g["n_offsets"] == w["n_offsets"]
        assert g["hseg_score"].view(np.uint32) == w["hseg_score"].view(np.uint32)
        assert np.abs(g["scores"] - w["scores"]).max() <= 1e-4
        assert np.array_equal(g["digits"], w["digits"])
    dcards.free()
    res.free()


def test_edge_case_frames(ctx, pkg, oracle):
    """blank, pure noise, a frame whose card is upside down, a card-less frame with one strong
    edge, and a normal frame; host pointers in and out (staging path)."""
    rng = np.random.default_rng(3)
    normal, _ = oracle.synth_frame(SEED, 11)
    blank = np.full((480, 640), 128, np.uint8)
    noise = rng.integers(0, 256, (480, 640)).astype(np.uint8)
    flipped = normal[::-1, ::-1].copy()
    one_edge = np.full((480, 640), 50, np.uint8)
    one_edge[:105] = 200
    black = np.zeros((480, 640), np.uint8)
    white = np.full((480, 640), 255, np.uint8)
    frames = np.stack([blank, noise, flipped, one_edge, normal, black, white])
    n = len(frames)
    res = np.zeros(n, pkg.RESULT_DTYPE)
    cards = np.zeros((n, 270, 428), np.uint8)
    ctx.pipeline(frames, n, res, cards)
    for i in range(n):
        w, wcard = oracle.scan_frame(frames[i])
        g = res[i]
        assert np.array_equal(g["found"], w["found"]), i
        assert g["found_all"] == w["found_all"], i
        assert np.array_equal(cards[i], wcard), i
        assert g["flags"] == w["flags"], (i, g["flags"], w["flags"])
        if w["found_all"]:
            assert np.array_equal(g["corners"].view(np.uint32), w["corners"].view(np.uint32))
            assert g["vseg_y_offset"] == w["vseg_y_offset"]
            assert np.abs(g["scores"] - w["scores"]).max() <= 1e-4
    assert res[0]["found_all"] == 0 and res[5]["found_all"] == 0 and res[6]["found_all"] == 0
    assert res[2]["flags"] & pkg.FLAG_UPSIDE_DOWN
    assert res[3]["found"].tolist() == [1, 0, 0, 0]
    assert res[4]["flags"] & pkg.FLAG_VSEG_OK


def test_truncated_corners_option(ctx, pkg, oracle):
    """cython_dmz/dmz.pyx:267-270 casts the corner points to int before dmz_transform_card."""
    n = 4
    frames = np.stack([oracle.synth_frame(SEED, 20 + i)[0] for i in range(n)])
    res = np.zeros(n, pkg.RESULT_DTYPE)
    cards = np.zeros((n, 270, 428), np.uint8)
    ctx.pipeline(frames, n, res, cards, options=pkg.OPT_TRUNCATE_CORNERS)
    for i in range(n):
        w, wcard = oracle.scan_frame(frames[i], truncate=True)
        assert np.array_equal(cards[i], wcard), i
        assert res[i]["vseg_y_offset"] == w["vseg_y_offset"]


def test_detect_then_transform_then_scan_equals_pipeline(ctx, pkg, oracle):
    """The three batched entry points chained by the caller == dmz_hip_pipeline_batch."""
    n = 5
    frames = np.stack([oracle.synth_frame(SEED, 30 + i)[0] for i in range(n)])
    r1 = np.zeros(n, pkg.RESULT_DTYPE)
    c1 = np.zeros((n, 270, 428), np.uint8)
    ctx.pipeline(frames, n, r1, c1)
    r2 = np.zeros(n, pkg.RESULT_DTYPE)
    c2 = np.zeros((n, 270, 428), np.uint8)
    ctx.detect(frames, n, r2)
    ctx.transform(frames, n, r2, c2)
    ctx.scan_cards(c2, n, r2, only_warped=True)
    assert np.array_equal(c1, c2)
    assert r1.tobytes() == r2.tobytes()


def test_chroma_fallback_planes(ctx, pkg, oracle):
    """dmz.cpp:346-369: an edge missing on Y is searched on Cb then Cr (rho doubled)."""
    y, _ = oracle.synth_frame(SEED, 40)
    y_no_top = y.copy()
    y_no_top[80:130] = y_no_top[140:141]  # wipe the top edge from the luma plane
    cb = np.ascontiguousarray(y[::2, ::2])  # half-size plane that still has all four edges
    cr = np.full((240, 320), 128, np.uint8)
    res = np.zeros(1, pkg.RESULT_DTYPE)
    ctx.detect(y_no_top[None], 1, res, cb=cb[None], cr=cr[None])
    w = oracle.detect_edges(y_no_top, cb=cb, cr=cr)
    assert np.array_equal(res[0]["found"], w["found"])
    m = w["found"] != 0
    assert np.array_equal(res[0]["rho"][m].view(np.uint32), w["rho"][m].view(np.uint32))
    assert np.array_equal(res[0]["theta"][m].view(np.uint32), w["theta"][m].view(np.uint32))
    assert np.array_equal(res[0]["corners"].view(np.uint32), w["corners"].view(np.uint32))
    only_y = oracle.detect_edges(y_no_top)
    assert only_y["found"][0] == 0 and w["found"][0] == 1  # the fallback is what found the top edge


def test_bad_arguments_are_errors_not_crashes(ctx, pkg):
    res = np.zeros(1, pkg.RESULT_DTYPE)
    with pytest.raises(pkg.DmzHipError):
        ctx.pipeline(np.zeros((1, 480, 640), np.uint8), 0, res)
    with pytest.raises(pkg.DmzHipError):  # 4000x3000: boxes exceed the LDS-resident detect kernel
        ctx.detect(np.zeros((1, 3000, 4000), np.uint8), 1, res, width=4000, height=3000)


def test_detect_other_frame_geometries(ctx, pkg, oracle):
    """dmz.cpp:279-341 works on the central 4:3 region of any frame size; 1280x720 gives
    583x43 / 58x361 boxes (bigger workgroups, different LDS layout), portrait swaps insets."""
    y, _ = oracle.synth_frame(SEED, 50)
    big = np.full((720, 1280), 60, np.uint8)
    big[:, 160:1120] = np.repeat(np.repeat(y, 3, axis=0), 3, axis=1)[::2, ::2][:720, :960]
    small = np.ascontiguousarray(y[::2, ::2])  # 320x240 (the chroma-plane geometry)
    portrait = np.ascontiguousarray(y.T)       # 480x640, orientation 1
    for frame, orientation in ((big, 3), (small, 3), (portrait, 1), (y, 4)):
        h, w = frame.shape
        res = np.zeros(1, pkg.RESULT_DTYPE)
        ctx.detect(frame[None], 1, res, width=w, height=h, orientation=orientation)
        want = oracle.detect_edges(frame, orientation=orientation)
        assert np.array_equal(res[0]["found"], want["found"]), (w, h, orientation)
        m = want["found"] != 0
        assert np.array_equal(res[0]["rho"][m].view(np.uint32), want["rho"][m].view(np.uint32)), (w, h)
        assert np.array_equal(res[0]["theta"][m].view(np.uint32), want["theta"][m].view(np.uint32))
        assert np.array_equal(res[0]["corners"].view(np.uint32), want["corners"].view(np.uint32))
        assert res[0]["found_all"] == want["found_all"]


def test_hough_lines_at_the_limits_of_every_box(ctx, pkg, oracle):
    """detect.hip keeps vote counters only for the rho bins a pixel of the box can reach (its corners bound them) and adds two
    copies of them in the arg-max.  Straight edges along the rims and through the middle of each detection box, tilted to both
    ends of the +-5 degree fan and beyond it, at the standard geometry and at 1280x720 / 320x240: found flags, the winning
    (rho, theta) and the corners equal the oracle's."""
    rng = np.random.default_rng(2024)
    for (h, w, orientation) in ((480, 640, 3), (720, 1280, 3), (240, 320, 3)):
        boxes = oracle.detection_boxes(w, h, orientation)  # four (x, y, w, h): the wide ones look for horizontal lines
        yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
        frames = []
        for tilt in (-6.0, -5.0, -4.5, 0.0, 4.5, 5.0, 6.0):
            t = np.tan(np.radians(tilt))
            for frac in (0.0, 0.1, 0.5, 0.9, 1.0):
                # one nearly horizontal edge through the top and the bottom box, one nearly vertical through the left and the right
                # one: at the box's first / last rows (columns) and in between, pivoting on the box's centre
                f = np.full((h, w), 40.0)
                for (bx, by, bw, bh) in boxes:
                    cx, cy = bx + bw / 2.0, by + bh / 2.0
                    if bw > bh:
                        side = (yy - (by + frac * (bh - 1)) - t * (xx - cx)) > 0
                        band = np.abs(yy - cy) < 3 * bh
                    else:
                        side = (xx - (bx + frac * (bw - 1)) - t * (yy - cy)) > 0
                        band = np.abs(xx - cx) < 3 * bw
                    f = np.where(band & side, f + 60.0, f)
                frames.append((f + rng.integers(0, 6, (h, w))).clip(0, 255).astype(np.uint8))
        frames = np.stack(frames)
        n = len(frames)
        res = np.zeros(n, pkg.RESULT_DTYPE)
        ctx.detect(frames, n, res, width=w, height=h, orientation=orientation)
        nfound = 0
        for i in range(n):
            want = oracle.detect_edges(frames[i], orientation=orientation)
            assert np.array_equal(res[i]["found"], want["found"]), (w, h, i)
            m = want["found"] != 0
            nfound += int(m.sum())
            assert np.array_equal(res[i]["rho"][m].view(np.uint32), want["rho"][m].view(np.uint32)), (w, h, i)
            assert np.array_equal(res[i]["theta"][m].view(np.uint32), want["theta"][m].view(np.uint32)), (w, h, i)
            assert np.array_equal(res[i]["corners"].view(np.uint32), want["corners"].view(np.uint32)), (w, h, i)
        assert nfound >= n, (w, h, nfound)  # (the frames do put lines into the boxes)


def test_dense_candidate_boxes_in_every_kernel_form(ctx, pkg, oracle):
    """Card-less texture leaves hundreds of weak Canny candidates per wave: the hysteresis then floods on bitmaps instead of
    walking candidate lists (detect.hip).  Noise, a noisy ramp and a fine checker at the standard geometry (single-walk
    kernels, parked and register-resident gradients) and at 1280x720 / 320x240 / portrait (generic two-walk kernels, other
    wave counts): found flags, lines and corners equal the oracle's."""
    rng = np.random.default_rng(404)
    for (h, w, orientation) in ((480, 640, 3), (720, 1280, 3), (240, 320, 3), (640, 480, 1), (480, 640, 4)):
        yy, xx = np.mgrid[0:h, 0:w]
        frames = [rng.integers(18, 58, (h, w), dtype=np.uint8),
                  ((xx * 0.2 + yy * 0.1 + rng.integers(0, 50, (h, w))) % 256).astype(np.uint8),
                  (np.where(((xx // 3) + (yy // 3)) % 2 == 0, 200, 40) + rng.integers(0, 30, (h, w))).astype(np.uint8)]
        for k, frame in enumerate(frames):
            res = np.zeros(1, pkg.RESULT_DTYPE)
            ctx.detect(frame[None], 1, res, width=w, height=h, orientation=orientation)
            want = oracle.detect_edges(frame, orientation=orientation)
            assert np.array_equal(res[0]["found"], want["found"]), (w, h, orientation, k)
            m = want["found"] != 0
            assert np.array_equal(res[0]["rho"][m].view(np.uint32), want["rho"][m].view(np.uint32)), (w, h, k)
            assert np.array_equal(res[0]["theta"][m].view(np.uint32), want["theta"][m].view(np.uint32)), (w, h, k)
            assert np.array_equal(res[0]["corners"].view(np.uint32), want["corners"].view(np.uint32)), (w, h, k)
            assert res[0]["found_all"] == want["found_all"]


def test_padded_and_unaligned_strides(ctx, pkg, oracle):
    """frames embedded in larger buffers: row stride > width (aligned and odd), frame stride with slack, an
    unaligned base address -- detect and transform must not depend on the packing"""
    n = 5
    frames = np.stack([oracle.synth_frame(SEED, 300 + i)[0] for i in range(n)])
    want = [oracle.scan_frame(f) for f in frames]
    for row_stride, slack, base_off in ((704, 0, 0), (643, 1931, 0), (640, 64, 3), (641, 7, 1)):
        frame_stride = row_stride * 480 + slack
        buf = np.full(base_off + n * frame_stride + 64, 0x5A, np.uint8)
        for i in range(n):
            view = buf[base_off + i * frame_stride: base_off + i * frame_stride + row_stride * 480]
            view = view.reshape(480, row_stride)
            view[:, :640] = frames[i]
        d = ctx.alloc(buf.nbytes)
        d.upload(buf)
        res = np.zeros(n, pkg.RESULT_DTYPE)
        cards = np.zeros((n, 270, 428), np.uint8)
        ctx.detect(d.ptr + base_off, n, res, frame_stride=frame_stride, row_stride=row_stride)
        ctx.transform(d.ptr + base_off, n, res, cards, frame_stride=frame_stride, row_stride=row_stride)
        for i in range(n):
            w, wcard = want[i]
            assert np.array_equal(res[i]["found"], w["found"]), (row_stride, i)
            assert np.array_equal(res[i]["corners"].view(np.uint32), w["corners"].view(np.uint32)), (row_stride, i)
            if w["found_all"]:
                assert np.array_equal(cards[i], wcard), (row_stride, base_off, i)
        d.free()


def test_card_stride_with_slack(ctx, pkg, oracle):
    """scan + expiry on cards that sit 116 KiB apart instead of tightly packed"""
    n = 6
    stride = pkg.CARD_BYTES + 3224  # multiple of 4
    cards = np.stack([oracle.synth_card(SEED, 40 + i)[0] for i in range(n)])
    buf = np.full(n * stride, 0xA5, np.uint8)
    for i in range(n):
        buf[i * stride: i * stride + pkg.CARD_BYTES] = cards[i].reshape(-1)
    res = np.zeros(n, pkg.RESULT_DTYPE)
    exp = np.zeros(n, pkg.EXPIRY_DTYPE)
    ctx._check(ctx.lib.dmz_hip_scan_cards_batch(ctx.h, buf.ctypes.data, stride, n, 0, res.ctypes.data))
    ctx._check(ctx.lib.dmz_hip_scan_expiry_batch(ctx.h, buf.ctypes.data, stride, n, res.ctypes.data, exp.ctypes.data))
    res2 = np.zeros(n, pkg.RESULT_DTYPE)
    exp2 = np.zeros(n, pkg.EXPIRY_DTYPE)
    ctx.scan_cards(cards, n, res2)
    ctx.scan_expiry(cards, n, res2, exp2)
    assert res.tobytes() == res2.tobytes() and exp.tobytes() == exp2.tobytes()
    with pytest.raises(pkg.DmzHipError):
        ctx._check(ctx.lib.dmz_hip_scan_cards_batch(ctx.h, buf.ctypes.data, stride + 2, n, 0, res.ctypes.data))


def test_contexts_are_independent(pkg, oracle):
    """context life cycle: repeated create/destroy, and two contexts driven from two host threads at once
    (one context per thread, like dmz_context) produce the single-context result"""
    import threading
    for _ in range(8):
        c = pkg.Context(0)
        c.close()
    n = 64
    ref_ctx = pkg.Context(0)
    y = ref_ctx.alloc(n * pkg.FRAME_BYTES)
    ref_ctx.synth_frames(SEED, 900, n, y.ptr)
    ref_ctx.synchronize()
    want_res = np.zeros(n, pkg.RESULT_DTYPE)
    want_exp = np.zeros(n, pkg.EXPIRY_DTYPE)
    ref_ctx.pipeline_expiry(y.ptr, n, want_res, want_exp)
    out = {}

    def worker(k):
        c = pkg.Context(0)
        for rep in range(4):
            res = np.zeros(n, pkg.RESULT_DTYPE)
            exp = np.zeros(n, pkg.EXPIRY_DTYPE)
            c.pipeline_expiry(y.ptr, n, res, exp)
            out[(k, rep)] = (res.tobytes(), exp.tobytes())
        c.close()

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for key, (r, e) in out.items():
        assert r == want_res.tobytes() and e == want_exp.tobytes(), key
    y.free()
    ref_ctx.close()


def test_records_do_not_depend_on_previous_buffer_contents(ctx, pkg):
    """every byte of the result and expiry records is defined by the call: poisoned output buffers give the
    same bytes as zeroed ones (device and host destinations)"""
    n = 96
    y = ctx.alloc(n * pkg.FRAME_BYTES)
    ctx.synth_frames(SEED, 4000, n, y.ptr)
    # two frames without a card: their rectified "cards" must come back as zeros, not as buffer leftovers
    blank = np.full(pkg.FRAME_BYTES, 90, np.uint8)
    for i in (5, 77):
        ctx._check(ctx.lib.dmz_hip_memcpy_h2d(ctx.h, y.ptr + i * pkg.FRAME_BYTES, blank.ctypes.data, blank.nbytes))
    outs = []
    for fill in (0x00, 0xA5, 0xFF):
        res = ctx.alloc(n * 1024)
        exp = ctx.alloc(n * pkg.EXPIRY_DTYPE.itemsize)
        cards = ctx.alloc(n * pkg.CARD_BYTES)
        res.upload(np.full(n * 1024, fill, np.uint8))
        exp.upload(np.full(n * pkg.EXPIRY_DTYPE.itemsize, fill, np.uint8))
        cards.upload(np.full(n * pkg.CARD_BYTES, fill, np.uint8))
        ctx.pipeline_expiry(y.ptr, n, res.ptr, exp.ptr, cards.ptr)
        ctx.synchronize()
        c = cards.download(np.uint8).reshape(n, -1)
        assert not c[5].any() and not c[77].any()
        outs.append((res.download(np.uint8).tobytes(), exp.download(np.uint8).tobytes(), c.tobytes()))
        cards.free()
        res.free()
        exp.free()
        hres = np.full(n, 0, pkg.RESULT_DTYPE)
        hexp = np.zeros(n, pkg.EXPIRY_DTYPE)
        hres.view(np.uint8)[:] = fill
        hexp.view(np.uint8)[:] = fill
        hcards = np.full((n, pkg.CARD_BYTES), fill, np.uint8)
        ctx.pipeline_expiry(y.ptr, n, hres, hexp, hcards)
        outs.append((hres.tobytes(), hexp.tobytes(), hcards.tobytes()))
    assert all(o == outs[0] for o in outs)
    y.free()


def test_chroma_fallback_batch(ctx, pkg, oracle):
    """a batch where random edges are wiped from Y (and some from Cb too): each of the four edges of each
    frame must come from the first plane that shows it -- Y, then Cb, then Cr -- exactly as in the oracle"""
    import os
    n = int(os.environ.get("DMZ_CHROMA_FRAMES", "40"))
    rng = np.random.default_rng(99)
    ys, cbs, crs = [], [], []
    for i in range(n):
        y, _ = oracle.synth_frame(SEED, 700 + i)
        cb = np.ascontiguousarray(y[::2, ::2])
        cr = np.ascontiguousarray(y[1::2, 1::2])
        y = y.copy()
        wipes = [(slice(80, 130), slice(None)), (slice(350, 400), slice(None)),
                 (slice(None), slice(80, 135)), (slice(None), slice(505, 560))]
        for e, (rs, cs) in enumerate(wipes):
            r = rng.random()
            if r < 0.35:  # not on Y
                y[rs, cs] = 60
                if r < 0.15:  # not on Cb either
                    cb[(slice(rs.start // 2, rs.stop // 2) if rs.start is not None else rs,
                        slice(cs.start // 2, cs.stop // 2) if cs.start is not None else cs)] = 60
                    if r < 0.05:  # nowhere
                        cr[(slice(rs.start // 2, rs.stop // 2) if rs.start is not None else rs,
                            slice(cs.start // 2, cs.stop // 2) if cs.start is not None else cs)] = 60
        ys.append(y)
        cbs.append(cb)
        crs.append(cr)
    ys, cbs, crs = (np.ascontiguousarray(np.stack(v)) for v in (ys, cbs, crs))
    res = np.zeros(n, pkg.RESULT_DTYPE)
    ctx.detect(ys, n, res, cb=cbs, cr=crs)
    from_chroma = missing = 0
    for i in range(n):
        w = oracle.detect_edges(ys[i], cb=cbs[i], cr=crs[i])
        assert np.array_equal(res[i]["found"], w["found"]), i
        m = w["found"] != 0
        assert np.array_equal(res[i]["rho"][m].view(np.uint32), w["rho"][m].view(np.uint32)), i
        assert np.array_equal(res[i]["theta"][m].view(np.uint32), w["theta"][m].view(np.uint32)), i
        assert res[i]["found_all"] == w["found_all"], i
        if w["found_all"]:
            assert np.array_equal(res[i]["corners"].view(np.uint32), w["corners"].view(np.uint32)), i
        only_y = oracle.detect_edges(ys[i])
        from_chroma += int((w["found"] != only_y["found"]).sum())
        missing += int((w["found"] == 0).sum())
    assert from_chroma >= n // 4  # the batch does exercise the fallback
    print("edges recovered from chroma: %d, edges found nowhere: %d" % (from_chroma, missing))
