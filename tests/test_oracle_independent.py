"""Independent NumPy restatements of the OpenCV-2.4 arithmetic the oracle restates in C, written from
SURVEY.md Appendix A (A2, A5-A10) and from the reference's in-tree llcv_equalize_hist (cv/stats.cpp:116-159)
-- not from oracle/orc_cv.c -- and compared bit for bit with the oracle.  The OpenCV libraries are not
in the image, so this is corroboration by a second, differently structured implementation (vectorised
float64 / integer NumPy instead of scalar C loops), not a pin on OpenCV itself; its value is that an
implementation slip in the C oracle on a GENERAL homography (the biggest byte mover of the path, and
the function the HIP warp kernel is compared with) would show up here."""
import ctypes as C

import numpy as np

INT_MIN, INT_MAX = -2147483648, 2147483647


def invert3x3(m32):
    """A10 (1): M -> double, 3 x 3 inverse as d = 1 / det, adjugate * d."""
    S = np.asarray(m32, np.float32).astype(np.float64).reshape(3, 3)
    det = (S[0, 0] * (S[1, 1] * S[2, 2] - S[1, 2] * S[2, 1]) - S[0, 1] * (S[1, 0] * S[2, 2] - S[1, 2] * S[2, 0])
           + S[0, 2] * (S[1, 0] * S[2, 1] - S[1, 1] * S[2, 0]))
    if det == 0.0:
        return np.zeros(9, np.float64)  # cv::invert reports failure and zero-fills the result
    d = 1.0 / det
    t = np.empty(9, np.float64)
    t[0] = (S[1, 1] * S[2, 2] - S[1, 2] * S[2, 1]) * d
    t[1] = (S[0, 2] * S[2, 1] - S[0, 1] * S[2, 2]) * d
    t[2] = (S[0, 1] * S[1, 2] - S[0, 2] * S[1, 1]) * d
    t[3] = (S[1, 2] * S[2, 0] - S[1, 0] * S[2, 2]) * d
    t[4] = (S[0, 0] * S[2, 2] - S[0, 2] * S[2, 0]) * d
    t[5] = (S[0, 2] * S[1, 0] - S[0, 0] * S[1, 2]) * d
    t[6] = (S[1, 0] * S[2, 1] - S[1, 1] * S[2, 0]) * d
    t[7] = (S[0, 1] * S[2, 0] - S[0, 0] * S[2, 1]) * d
    t[8] = (S[0, 0] * S[1, 1] - S[0, 1] * S[1, 0]) * d
    return t


def bilinear_table():
    """A10 (3): w[alpha][4] = sat16(round(32768 * {(1-fx)(1-fy), fx(1-fy), (1-fx)fy, fx fy})), each tuple
    corrected to sum to 32768 (the deficit goes to the tap the table construction leaves last)."""
    a = np.arange(1024)
    fx, fy = (a & 31) / 32.0, (a >> 5) / 32.0
    w = np.stack([(1 - fx) * (1 - fy), fx * (1 - fy), (1 - fx) * fy, fx * fy], 1)
    t = np.clip(np.rint(w * 32768.0), -32768, 32767).astype(np.int64)
    t[:, 3] += 32768 - t.sum(1)  # only alpha = 0 ({32767, 0, 0, 1}); it cannot change an 8-bit result
    return t


def warp_perspective_np(src, m32, dw=428, dh=270):
    """A10: cvWarpPerspective(src, dst, M, CV_INTER_LINEAR + CV_WARP_FILL_OUTLIERS, 0), uchar."""
    M = invert3x3(m32)  # singular: all zeros -> W = 0 -> every pixel maps to source (0, 0)
    sh, sw = src.shape
    xs = np.arange(dw)
    xb = ((xs // 64) * 64).astype(np.float64)[None, :]  # block origin x (bw = 64)
    x1 = (xs % 64).astype(np.float64)[None, :]
    yy = np.arange(dh).astype(np.float64)[:, None]      # y + y1
    X0 = M[0] * xb + M[1] * yy + M[2]
    Y0 = M[3] * xb + M[4] * yy + M[5]
    W0 = M[6] * xb + M[7] * yy + M[8]
    W = W0 + M[6] * x1
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        Wi = np.where(W != 0.0, 32.0 / W, 0.0)
        fX = np.maximum(INT_MIN, np.minimum(INT_MAX, (X0 + M[0] * x1) * Wi))
        fY = np.maximum(INT_MIN, np.minimum(INT_MAX, (Y0 + M[3] * x1) * Wi))
    X = np.rint(fX).astype(np.int64)  # cvRound: half to even
    Y = np.rint(fY).astype(np.int64)
    sx = np.clip(X >> 5, -32768, 32767)
    sy = np.clip(Y >> 5, -32768, 32767)
    alpha = (Y & 31) * 32 + (X & 31)
    tab = bilinear_table()[alpha]                      # [dh, dw, 4]
    padded = np.zeros((sh + 2, sw + 2), np.int64)       # BORDER_CONSTANT 0 around the image
    padded[1:-1, 1:-1] = src

    def tap(dy, dx):
        px, py = sx + dx, sy + dy
        inside = (px >= 0) & (px < sw) & (py >= 0) & (py < sh)
        return np.where(inside, padded[np.clip(py, -1, sh) + 1, np.clip(px, -1, sw) + 1], 0)

    acc = tap(0, 0) * tab[..., 0] + tap(0, 1) * tab[..., 1] + tap(1, 0) * tab[..., 2] + tap(1, 1) * tab[..., 3]
    return np.clip((acc + 16384) >> 15, 0, 255).astype(np.uint8)


def test_warp_perspective_byte_exact_on_random_homographies(oracle):
    rng = np.random.default_rng(20260)
    yy, xx = np.mgrid[0:480, 0:640]
    base = ((xx * 7 + yy * 13) % 251).astype(np.uint8)
    dst = np.array([0, 0, 427, 0, 0, 269, 427, 269], np.float32)  # card corners tl, tr, bl, br
    n_cases, n_border, n_generic = 0, 0, 0
    for case in range(260):
        src = (base ^ rng.integers(0, 256, (480, 640)).astype(np.uint8)) if case % 3 else base
        kind = case % 5
        if kind == 0:    # the guide frame with a few pixels of jitter (what the corpus produces)
            q = np.array([106, 105, 533, 105, 106, 374, 533, 374], np.float32) + rng.uniform(-8, 8, 8).astype(np.float32)
        elif kind == 1:  # strong perspective
            q = np.array([106, 105, 533, 105, 106, 374, 533, 374], np.float32) + rng.uniform(-60, 60, 8).astype(np.float32)
        elif kind == 2:  # quads that leave the frame on one or more sides
            q = rng.uniform(-200, 840, 8).astype(np.float32)
        elif kind == 3:  # rotated / zoomed
            a, s = rng.uniform(-3.2, 3.2), rng.uniform(0.3, 2.5)
            c = np.array([[-214, -135], [214, -135], [-214, 135], [214, 135]], np.float64) * s
            r = np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]])
            q = (c @ r.T + rng.uniform(100, 500, 2)).astype(np.float32).reshape(8)
        else:            # nearly degenerate / self-crossing
            q = rng.uniform(200, 400, 8).astype(np.float32)
            q[6:] = q[:2] + rng.uniform(-2, 2, 2).astype(np.float32)
        m = oracle.calc_persp_transform(q, dst)
        if not np.isfinite(m).all():
            continue
        want = warp_perspective_np(src, m)
        got = oracle.warp_perspective(src, m)
        assert np.array_equal(got, want), (case, kind, int((got != want).sum()))
        n_cases += 1
        n_border += int((want == 0).mean() > 0.05)
        n_generic += int(kind >= 2)
    assert n_cases >= 200 and n_border >= 40 and n_generic >= 100


def test_warp_perspective_extreme_matrices(oracle):
    """W crossing zero inside the card, huge coordinates (INT clamp / sat16), singular matrices."""
    rng = np.random.default_rng(5)
    src = rng.integers(0, 256, (480, 640)).astype(np.uint8)
    mats = [
        np.array([1, 0, 0, 0, 1, 0, 0.004, 0, 1], np.float32),        # horizon at x = -250 (outside)
        np.array([1, 0, 0, 0, 1, 0, -0.004, 0.002, 1], np.float32),   # horizon crosses the card
        np.array([1e-6, 0, 0, 0, 1e-6, 0, 0, 0, 1], np.float32),      # source coordinates ~1e8: clamps
        np.array([1, 0, 1e9, 0, 1, -1e9, 0, 0, 1], np.float32),
        np.array([0.5, 0.1, 30, -0.2, 0.7, 40, 1e-4, -2e-4, 1], np.float32),
        np.array([1, 2, 3, 2, 4, 6, 0, 0, 1], np.float32),            # singular
    ]
    for k, m in enumerate(mats):
        assert np.array_equal(oracle.warp_perspective(src, m), warp_perspective_np(src, m)), k


def test_vseg_row_features_bits(oracle):
    """A5 (one-row ROI: max3 - min3 with replicated ends), A6 ((a + b + 1) >> 1), A7, A8."""
    rng = np.random.default_rng(1)
    for case in range(300):
        row = rng.integers(0, 256, 408).astype(np.uint8)
        if case % 7 == 0:
            row[:] = row[0]
        if case % 5 == 0:
            row = (row // 64 * 64).astype(np.uint8)
        p = np.concatenate([row[:1], row, row[-1:]]).astype(np.int64)
        g = np.maximum(np.maximum(p[:-2], p[1:-1]), p[2:]) - np.minimum(np.minimum(p[:-2], p[1:-1]), p[2:])
        d = (g[0::2] + g[1::2] + 1) >> 1
        f = d.astype(np.float32) * np.float32(1.0 / 255.0)
        smin, smax = float(f.min()), float(f.max())
        scale = 1.0 / (smax - smin) if smax - smin > np.finfo(np.float64).eps else 0.0
        shift = 0.0 - smin * scale
        want = f * np.float32(scale) + np.float32(shift)
        got = oracle.vseg_row_features(row)
        assert np.array_equal(got.view(np.uint32), want.astype(np.float32).view(np.uint32)), case


def test_hseg_gradient_sums_bits(oracle):
    """A5 (3 x 3 cross: max5 - min5, replicate at the ROI edge), A9, A8 on the 428 x 27 number strip."""
    rng = np.random.default_rng(2)
    for case in range(60):
        strip = rng.integers(0, 256, (27, 428)).astype(np.uint8)
        if case % 4 == 0:
            strip[:, 100:300] = 17
        p = np.pad(strip.astype(np.int64), 1, mode="edge")
        c, n, s, w, e = p[1:-1, 1:-1], p[:-2, 1:-1], p[2:, 1:-1], p[1:-1, :-2], p[1:-1, 2:]
        g = np.maximum.reduce([c, n, s, w, e]) - np.minimum.reduce([c, n, s, w, e])
        f = g.sum(0).astype(np.float32)
        smin, smax = float(f.min()), float(f.max())
        scale = 1.0 / (smax - smin) if smax - smin > np.finfo(np.float64).eps else 0.0
        want = f * np.float32(scale) + np.float32(0.0 - smin * scale)
        got = oracle.hseg_grad_sums(strip)
        assert np.array_equal(np.asarray(got, np.float32).view(np.uint32), want.astype(np.float32).view(np.uint32)), case


def test_equalize_hist_bytes(oracle):
    """llcv_equalize_hist (cv/stats.cpp:116-159): histogram, lut[i] = sat8(cvRound(cumsum * (255.f / n))), lut[0] = 0."""
    rng = np.random.default_rng(3)
    u8p = C.POINTER(C.c_uint8)
    for case in range(200):
        h, w = (27, 19) if case % 2 else (16, 11)
        img = rng.integers(0, 256 if case % 3 else 40, (h, w)).astype(np.uint8)
        if case % 11 == 0:
            img[:] = 200
        hist = np.bincount(img.reshape(-1), minlength=256)
        scale = np.float32(255.0) / np.float32(w * h)
        prod = np.cumsum(hist).astype(np.float32) * scale       # int * float -> float
        lut = np.clip(np.rint(prod.astype(np.float64)), 0, 255).astype(np.uint8)
        lut[0] = 0
        want = lut[img]
        got = img.copy()
        oracle.lib.orc_equalize_hist(got.ctypes.data_as(u8p), w, w, h)
        assert np.array_equal(got, want), case
