"""Hand-checked tiny cases for the oracle stages whose arithmetic lives in un-vendored
OpenCV (parity unpinned by the reference: SURVEY Appendix A) and for the in-tree stages
that need an OpenCV library symbol to run (Canny, Hough, equalise)."""
import numpy as np
import pytest


def test_detection_boxes_known_answers(oracle):
    # SURVEY Appendix E (dmz.cpp:279-341 evaluated for the canonical geometry)
    b = oracle.detection_boxes(640, 480, 3)
    assert b.tolist() == [[125, 91, 389, 28], [125, 360, 389, 28], [87, 119, 38, 241], [514, 119, 38, 241]]
    c = oracle.detection_boxes(320, 240, 3)
    assert c.tolist() == [[63, 46, 193, 14], [63, 179, 193, 14], [43, 60, 20, 119], [256, 60, 20, 119]]


def test_sobel7_constant_and_ramp(oracle):
    flat = np.full((12, 20), 77, np.uint8)
    assert not oracle.sobel7(flat, True).any() and not oracle.sobel7(flat, False).any()
    # horizontal ramp of slope 1: dx = sum_j deriv[j]*(j-3) * sum(smooth) = 32*64 away from the borders
    ramp = np.tile(np.arange(40, dtype=np.uint8), (12, 1))
    dx = oracle.sobel7(ramp, True)
    assert (dx[:, 3:-3] == 32 * 64).all()
    assert not oracle.sobel7(ramp, False).any()
    # replicate border: column 0 sees pixels {0,0,0,0,1,2,3} -> deriv = 5*1+4*2+1*3 = 16
    assert (dx[:, 0] == 16 * 64).all()
    # vertical orientation is the transpose
    assert np.array_equal(oracle.sobel7(ramp.T.copy(), False), dx.T)


def test_sobel7_saturates_to_int16(oracle):
    img = np.zeros((10, 16), np.uint8)
    img[:, 8:] = 255
    dx = oracle.sobel7(img, True)
    assert dx.max() == 32767  # 255*10*64 = 163200 saturates
    img2 = img[:, ::-1].copy()
    assert oracle.sobel7(img2, True).min() == -32768


def test_canny_and_hough_on_a_clean_vertical_step(oracle):
    h, w = 241, 38
    img = np.full((h, w), 60, np.uint8)
    img[:, 20:] = 180
    dx, dy = oracle.sobel7(img, True), oracle.sobel7(img, False)
    edges, low, high = oracle.adaptive_canny7(dx, dy)
    assert high == 3 * low or high == 3 * low + 1 or high == 3 * low + 2
    cols = np.unique(np.nonzero(edges)[1])
    assert set(edges.ravel().tolist()) <= {0, 255}
    assert len(cols) >= 1 and cols.min() >= 17 and cols.max() <= 22
    found, rho, theta, n, r, mx = oracle.hough(edges, dx, dy, vertical=True)
    assert found and mx > 40
    # vertical line x ~ 20: theta ~ pi, rho ~ -x
    assert abs(theta - np.pi) < 0.02 and abs(abs(rho) - 20) <= 2
    # a horizontal-line search on the same box must reject the pixels by gradient direction
    found_h, *_ = oracle.hough(edges, dx, dy, vertical=False)
    assert not found_h


def test_blank_and_flat_inputs_find_nothing(oracle):
    flat = np.full((480, 640), 128, np.uint8)
    r = oracle.detect_edges(flat)
    assert r["found"].tolist() == [0, 0, 0, 0] and r["found_all"] == 0
    # flat chroma planes never contribute (dmz.cpp:346-369 fallback)
    y, _ = oracle.synth_frame(1, 0)
    c = np.full((240, 320), 128, np.uint8)
    a, b = oracle.detect_edges(y), oracle.detect_edges(y, cb=c, cr=c)
    assert np.array_equal(a["found"], b["found"]) and np.array_equal(a["corners"], b["corners"])


def test_warp_identity_and_translation(oracle):
    rng = np.random.default_rng(0)
    src = rng.integers(0, 256, (480, 640)).astype(np.uint8)
    # corners exactly a 428x270 axis-aligned rectangle at (100, 90): warp == crop
    corners = np.array([100, 90, 100, 359, 527, 90, 527, 359], np.float32)  # tl, bl, tr, br
    card = oracle.transform_card(src, corners)
    assert np.array_equal(card, src[90:360, 100:528])
    # half-pixel shift in x: bilinear average of horizontal neighbours, (a+b+1)>>1 rounding
    corners2 = corners.copy()
    corners2[0::2] += 0.5
    card2 = oracle.transform_card(src, corners2)
    a = src[90:360, 100:528].astype(np.int32)
    b = src[90:360, 101:529].astype(np.int32)
    assert np.abs(card2.astype(np.int32) - ((a + b + 1) >> 1)).max() <= 1
    # outside the source: zero fill (CV_WARP_FILL_OUTLIERS, fillval 0)
    corners3 = np.array([-300, -300, -300, -31, 127, -300, 127, -31], np.float32)
    assert not oracle.transform_card(src, corners3)[:200, :200].any()


def test_row_preprocessing(oracle):
    row = np.zeros(408, np.uint8)
    row[100:110] = 200
    f = oracle.vseg_row_features(row)
    assert f.shape == (204,) and f.min() == 0.0 and f.max() == 1.0
    # gradient is non-zero only around the two steps
    nz = np.nonzero(f)[0]
    assert set(nz.tolist()) <= {49, 50, 54, 55}
    # all-equal row normalises to all zeros (convert.cpp comment / cvNormalize with zero range)
    assert not oracle.vseg_row_features(np.full(408, 9, np.uint8)).any()


def test_equalize_and_number_scores_shapes(oracle):
    card, digits = oracle.synth_card(3, 0)
    res = oracle.scan_card_image(card)
    assert res["flags"] & 4 and res["pattern_type"] == 1 and res["n_offsets"] == 16
    assert 121 <= res["vseg_y_offset"] <= 243
    s = res["scores"]
    assert s.shape == (16, 10) and (s >= 0).all() and (s <= 1.0001).all()
    # offsets are increasing and spaced like the generator's 18.3 px pitch
    d = np.diff(res["offsets"].astype(np.int32))
    assert (d >= 17).all() and (d <= 40).all()


def test_upside_down_and_empty_cards(oracle):
    card, _ = oracle.synth_card(3, 1)
    flipped = card[::-1, ::-1].copy()
    r = oracle.scan_card_image(flipped)
    assert r["flags"] & 2 and not (r["flags"] & 4)  # upside down, nothing else computed
    assert r["n_offsets"] == 0 and not r["scores"].any()
    blank = np.full((270, 428), 170, np.uint8)
    r2 = oracle.scan_card_image(blank)
    assert not (r2["flags"] & 4) and r2["vseg_score"] <= 15


def test_luhn(oracle):
    assert oracle.passes_luhn([4, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1])
    assert not oracle.passes_luhn([4, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2])
    for i in range(8):
        _, d = oracle.synth_frame(9, i)
        assert oracle.passes_luhn(d)


def test_plumbing_hand_checked(oracle):
    """cvSplit order; BT.601 fixed point of convert.cpp:448-490 on hand-computed pixels"""
    inter = np.arange(2 * 3 * 2, dtype=np.uint8).reshape(2, 3, 2)
    c1, c2 = oracle.split_u8(inter)
    assert c1.tolist() == [[0, 2, 4], [6, 8, 10]] and c2.tolist() == [[1, 3, 5], [7, 9, 11]]
    y = np.array([[100, 16, 235, 255, 0]], np.uint8)
    cb = np.array([[128, 128, 255, 0, 200]], np.uint8)
    cr = np.array([[128, 255, 128, 0, 30]], np.uint8)
    rgb = oracle.ycbcr_to_rgb(y, cb, cr)
    # grey stays grey
    assert rgb[0, 0].tolist() == [100, 100, 100]
    # Cr = +127: R = 16 + round(127 * 22987 / 16384) = 16 + 178, G = 16 + floor((127 * -11698 + 8192) / 16384) = 16 - 91 -> 0
    assert rgb[0, 1].tolist() == [194, 0, 16]
    # Cb = +127: B = 235 + 225 -> 255, G = 235 + floor((127 * -5636 + 8192) / 16384) = 235 - 44
    assert rgb[0, 2].tolist() == [235, 191, 255]
    # Cb = Cr = -128: R = 255 - 180 = 75, G = 255 + floor((721408 + 1497344 + 8192) / 16384) -> 255, B = 255 - 227 = 28
    assert rgb[0, 3].tolist() == [75, 255, 28]
    rgba = oracle.ycbcr_to_rgb(y, cb, cr, channels=4)
    assert np.array_equal(rgba[..., :3], rgb) and (rgba[..., 3] == 255).all()


def test_blur_card_median_against_pillow(oracle):
    """orc_blur_card's box = exact 25 x 25 per-channel median with replicated borders at the ROI edge:
    corroborated by an independent implementation (Pillow's RankFilter pads by edge replication)."""
    PIL = pytest.importorskip("PIL")
    from PIL import Image, ImageFilter
    rng = np.random.default_rng(21)
    img = rng.integers(0, 256, (270, 428, 3)).astype(np.uint8)
    # one box covering x 0..427, y 0..57 (the first digit's box is two number heights tall)
    out = oracle.blur_card(img, [1], 1, 426.0, 1, 0)
    for c in range(3):
        want = np.asarray(Image.fromarray(img[:58, :, c]).filter(ImageFilter.MedianFilter(25)))
        assert np.array_equal(out[:58, :, c], want), c
    assert np.array_equal(out[58:], img[58:])


def test_sobel7_and_morph_gradient_against_scipy(oracle):
    """independent corroboration of two restated OpenCV semantics: the 7-tap
    separable Sobel with BORDER_REPLICATE and the 3x3-cross morphological gradient"""
    ndi = pytest.importorskip("scipy.ndimage")
    rng = np.random.default_rng(22)
    img = rng.integers(0, 256, (46, 97)).astype(np.uint8)
    deriv = np.array([-1, -4, -5, 0, 5, 4, 1], np.int64)
    smooth = np.array([1, 6, 15, 20, 15, 6, 1], np.int64)
    a = img.astype(np.int64)
    dx = ndi.correlate1d(ndi.correlate1d(a, deriv, axis=1, mode="nearest"), smooth, axis=0, mode="nearest")
    dy = ndi.correlate1d(ndi.correlate1d(a, smooth, axis=1, mode="nearest"), deriv, axis=0, mode="nearest")
    assert np.array_equal(oracle.sobel7(img, True), np.clip(dx, -32768, 32767).astype(np.int16))
    assert np.array_equal(oracle.sobel7(img, False), np.clip(dy, -32768, 32767).astype(np.int16))
    cross = np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]], bool)
    grad = ndi.grey_dilation(img, footprint=cross, mode="nearest").astype(np.int32) - \
        ndi.grey_erosion(img, footprint=cross, mode="nearest").astype(np.int32)
    got = np.zeros_like(img)
    import ctypes as C
    oracle.lib.orc_morph_grad3_2d_cross(img.ctypes.data_as(C.c_void_p), img.shape[1], img.shape[1], img.shape[0],
                                        got.ctypes.data_as(C.c_void_p), img.shape[1])
    assert np.array_equal(got, grad.astype(np.uint8))


def test_normalize_scale_float_division_is_exact():
    """expiry.hip computes cvNormalize's (float)(255.0 / (double)max) as a float division: the two agree for
    every possible maximum of an int16 image"""
    mx = np.arange(1, 32768)
    a = (255.0 / mx.astype(np.float64)).astype(np.float32)
    b = np.float32(255.0) / mx.astype(np.float32)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
