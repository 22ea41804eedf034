"""GPU parity of the camera-side plumbing against the oracle: byte-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_deinterleave_c2_and_rgba(ctx, pkg, oracle):
    rng = np.random.default_rng(8)
    for h, w in ((240, 320), (3, 5), (1, 4), (7, 33)):
        inter = rng.integers(0, 256, (h, w, 2)).astype(np.uint8)
        c1 = np.zeros((h, w), np.uint8)
        c2 = np.zeros((h, w), np.uint8)
        ctx.deinterleave_c2(inter, h * w, c1, c2)
        w1, w2 = oracle.split_u8(inter)
        assert np.array_equal(c1, w1) and np.array_equal(c2, w2), (h, w)
    # device-resident batch of 64 half-size CbCr planes
    n = 64 * 240 * 320
    inter = rng.integers(0, 256, (n, 2)).astype(np.uint8)
    d_in, d1, d2 = ctx.alloc(n * 2), ctx.alloc(n), ctx.alloc(n)
    d_in.upload(inter)
    ctx.deinterleave_c2(d_in.ptr, n, d1.ptr, d2.ptr)
    ctx.synchronize()
    assert np.array_equal(d1.download(np.uint8), inter[:, 0]) and np.array_equal(d2.download(np.uint8), inter[:, 1])
    for b in (d_in, d1, d2):
        b.free()
    for size in (4, 16, 4 * 999):
        rgba = rng.integers(0, 256, size * 4).astype(np.uint8)
        r = np.zeros(size, np.uint8)
        ctx.deinterleave_rgba_to_r(rgba, r, size)
        assert np.array_equal(r, oracle.deinterleave_rgba_to_r(rgba))
    with pytest.raises(pkg.DmzHipError):
        ctx.deinterleave_rgba_to_r(rgba, r, 6)


def test_ycbcr_to_rgb(ctx, pkg, oracle):
    rng = np.random.default_rng(9)
    # every (Cb, Cr) pair with a sweep of Y: the whole input space of the chroma terms
    cb, cr = np.meshgrid(np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8))
    for yv in (0, 1, 16, 100, 128, 235, 254, 255):
        y = np.full((256, 256), yv, np.uint8)
        for ch in (3, 4):
            out = np.zeros((256, 256, ch), np.uint8)
            ctx.ycbcr_to_rgb(y, np.ascontiguousarray(cb), np.ascontiguousarray(cr), 256 * 256, out, channels=ch)
            assert np.array_equal(out, oracle.ycbcr_to_rgb(y, cb, cr, ch)), (yv, ch)
    # sizes that are not a multiple of 4 pixels
    for npx in (1, 2, 3, 5, 115560 + 3):
        y, cb1, cr1 = (rng.integers(0, 256, npx).astype(np.uint8) for _ in range(3))
        out = np.zeros((npx, 3), np.uint8)
        ctx.ycbcr_to_rgb(y, cb1, cr1, npx, out)
        assert np.array_equal(out, oracle.ycbcr_to_rgb(y.reshape(1, -1), cb1.reshape(1, -1), cr1.reshape(1, -1))[0]), npx
    with pytest.raises(pkg.DmzHipError):
        ctx.ycbcr_to_rgb(y, cb1, cr1, npx, out, channels=2)


def test_colour_card_chain(ctx, pkg, oracle):
    """NV21-style frames: Y + interleaved half-size CbCr -> planes -> detect (chroma fallback available)
    -> rectify Y, Cb, Cr (chroma with `upsample`) -> RGB card; against the oracle's chain."""
    n = 6
    rng = np.random.default_rng(10)
    ys = np.stack([oracle.synth_frame(77, i)[0] for i in range(n)])
    # chroma: smooth gradients + noise at half resolution
    gy, gx = np.mgrid[0:240, 0:320]
    cbp = np.stack([(128 + 40 * np.sin(gx / 37.0 + i) + rng.integers(-3, 4, (240, 320))).astype(np.uint8) for i in range(n)])
    crp = np.stack([(120 + 50 * np.cos(gy / 29.0 - i) + rng.integers(-3, 4, (240, 320))).astype(np.uint8) for i in range(n)])
    inter = np.ascontiguousarray(np.stack([cbp, crp], axis=-1))
    cb = np.zeros_like(cbp)
    cr = np.zeros_like(crp)
    ctx.deinterleave_c2(inter, n * 240 * 320, cb, cr)
    assert np.array_equal(cb, cbp) and np.array_equal(cr, crp)
    res = np.zeros(n, pkg.RESULT_DTYPE)
    ctx.detect(ys, n, res, cb=cb, cr=cr)
    ycard = np.zeros((n, 270, 428), np.uint8)
    cbcard = np.zeros((n, 270, 428), np.uint8)
    crcard = np.zeros((n, 270, 428), np.uint8)
    ctx.transform(ys, n, res, ycard)
    ctx.transform(cb, n, res, cbcard, width=320, height=240, options=pkg.OPT_UPSAMPLE)
    ctx.transform(cr, n, res, crcard, width=320, height=240, options=pkg.OPT_UPSAMPLE)
    rgb = np.zeros((n, 270, 428, 3), np.uint8)
    ctx.ycbcr_to_rgb(ycard, cbcard, crcard, n * 270 * 428, rgb)
    for i in range(n):
        w = oracle.detect_edges(ys[i], 3, cbp[i], crp[i])
        assert np.array_equal(res[i]["corners"].view(np.uint32), w["corners"].view(np.uint32)), i
        if not w["found_all"]:
            continue
        wy = oracle.transform_card(ys[i], w["corners"])
        wcb = oracle.transform_card(cbp[i], w["corners"], upsample=True)
        wcr = oracle.transform_card(crp[i], w["corners"], upsample=True)
        assert np.array_equal(ycard[i], wy) and np.array_equal(cbcard[i], wcb) and np.array_equal(crcard[i], wcr), i
        assert np.array_equal(rgb[i], oracle.ycbcr_to_rgb(wy, wcb, wcr)), i


def test_focus_and_brightness_scores(ctx, pkg, oracle):
    """dmz_focus_score / dmz_brightness_score: float bits equal the oracle's (exact integer sums, then the
    same double operations)"""
    rng = np.random.default_rng(12)
    n = 12
    frames = np.stack([oracle.synth_frame(5, i)[0] for i in range(n - 3)] +
                      [np.zeros((480, 640), np.uint8), np.full((480, 640), 255, np.uint8),
                       rng.integers(0, 256, (480, 640)).astype(np.uint8)])
    for full in (False, True):
        focus = np.zeros(n, np.float32)
        bright = np.zeros(n, np.float32)
        ctx.scores(frames, n, focus, bright, use_full_image=full)
        for i in range(n):
            assert focus[i].view(np.uint32) == oracle.focus_score(frames[i], full).view(np.uint32), (i, full)
            assert bright[i].view(np.uint32) == oracle.brightness_score(frames[i], full).view(np.uint32), (i, full)
    # other image sizes scale the ROI (dmz.cpp:152-160); device-resident input, one output only
    for (w, h) in ((1280, 720), (320, 240), (480, 640)):
        img = rng.integers(0, 256, (3, h, w)).astype(np.uint8)
        d = ctx.alloc(img.nbytes)
        d.upload(img)
        focus = np.zeros(3, np.float32)
        ctx.scores(d.ptr, 3, focus, None, width=w, height=h)
        for i in range(3):
            assert focus[i].view(np.uint32) == oracle.focus_score(img[i]).view(np.uint32), (w, h, i)
        d.free()


def test_blur_cards(ctx, pkg, oracle):
    """dmz_blur_card: 25 x 25 median of the leading digit boxes, in place, in digit order; byte-exact"""
    rng = np.random.default_rng(13)
    n = 5
    rgb = rng.integers(0, 256, (n, 270, 428, 3)).astype(np.uint8)
    rgb[1] = (np.indices((270, 428)).sum(0)[..., None] % 251).astype(np.uint8)  # smooth ramps
    sess = np.zeros(n, pkg.SESSION_DTYPE)
    cases = [(16, 17.9, 150, 4), (15, 19.2, 241, 0), (16, 18.0, 0, 12), (16, 17.0, 130, 16), (16, 20.5, 200, 4)]
    for i, (no, nw, yo, _) in enumerate(cases):
        sess[i]["n_offsets"] = no
        sess[i]["offsets"][:no] = 30 + 19 * np.arange(no) + i
        sess[i]["number_width"] = np.float32(nw)
        sess[i]["vseg_y_offset"] = yo
    sess[4]["offsets"][0] = 0      # box clipped at the left edge
    sess[4]["offsets"][15] = 420   # and at the right edge (only reached with unblur 0)
    for unblur in (4, 0, -1):
        got = rgb.copy()
        ctx.blur_cards(got, n, sess, unblur)
        for i in range(n):
            want = oracle.blur_card(rgb[i], sess[i]["offsets"], sess[i]["n_offsets"], sess[i]["number_width"],
                                    sess[i]["vseg_y_offset"], unblur)
            assert np.array_equal(got[i], want), (i, unblur)
    # RGBA
    rgba = rng.integers(0, 256, (1, 270, 428, 4)).astype(np.uint8)
    got = rgba.copy()
    ctx.blur_cards(got, 1, sess[:1], 4, channels=4)
    assert np.array_equal(got[0], oracle.blur_card(rgba[0], sess[0]["offsets"], 16, sess[0]["number_width"], 150, 4))
