"""ctypes binding of the CPU oracle (oracle/liboracle.so) and, when present, of
the partial reference build (oracle/_ref/libdmzref.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  Nothing under card.io-dmz_amd/ imports this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
WEIGHTS = os.path.join(ROOT, "card.io-dmz_amd", "weights", "dmz_models.bin")

# field-for-field mirror of orc_frame_result / dmz_hip_frame_result (1024 bytes)
RESULT_DTYPE = np.dtype([
    ("found", "<i4", (4,)), ("rho", "<f4", (4,)), ("theta", "<f4", (4,)),
    ("corners", "<f4", (8,)), ("found_all", "<i4"), ("flags", "<i4"),
    ("vseg_score", "<f4"), ("vseg_y_offset", "<i4"), ("pattern_type", "<i4"),
    ("n_offsets", "<i4"), ("offsets", "<u2", (16,)), ("hseg_score", "<f4"),
    ("number_width", "<f4"), ("pattern_offset", "<i4"), ("number_score", "<f4"),
    ("digits", "u1", (16,)), ("scores", "<f4", (16, 10)),
    ("expiry_month", "<i4"), ("expiry_year", "<i4"), ("reserved", "u1", (208,)),
])
assert RESULT_DTYPE.itemsize == 1024

FLAG_USABLE, FLAG_UPSIDE_DOWN, FLAG_VSEG_OK, FLAG_WARPED = 1, 2, 4, 8

# mirror of orc_expiry_group / orc_expiry_result == dmz_hip_expiry_group / dmz_hip_expiry_result
EXPIRY_MAX_GROUPS = 8
EXPIRY_GROUP_DTYPE = np.dtype([
    ("top", "<i2"), ("left", "<i2"), ("width", "<i2"), ("height", "<i2"),
    ("char_top", "<i2", (5,)), ("char_left", "<i2", (5,)),
    ("stripe_base_row", "<i2"), ("reserved", "<i2"), ("scores", "<f4", (4, 10)),
])
assert EXPIRY_GROUP_DTYPE.itemsize == 192
EXPIRY_DTYPE = np.dtype([
    ("n_groups", "<i4"), ("n_found", "<i4"), ("n_stripes", "<i4"),
    ("stripe_base_row", "<i4", (3,)), ("stripe_sum", "<i8", (3,)),
    ("categorised", "<i4"), ("reserved", "<i4"),
    ("groups", EXPIRY_GROUP_DTYPE, (EXPIRY_MAX_GROUPS,)),
])
assert EXPIRY_DTYPE.itemsize == 56 + 8 * 192

_u8p = C.POINTER(C.c_uint8)
_f32p = C.POINTER(C.c_float)
_i16p = C.POINTER(C.c_int16)
_i32p = C.POINTER(C.c_int)
_u16p = C.POINTER(C.c_uint16)


def _p(a, t):
    return a.ctypes.data_as(t)


def build(force=False):
    so = os.path.join(HERE, "liboracle.so")
    if force or not os.path.exists(so):
        subprocess.check_call(["make", "-C", HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def load_weights():
    raw = np.fromfile(WEIGHTS, dtype=np.uint8)
    assert bytes(raw[:8]) == b"DMZW0001"
    return np.frombuffer(raw[16:].tobytes(), dtype="<f4").copy()


class Oracle:
    def __init__(self):
        self.lib = C.CDLL(build())
        self.weights = load_weights()
        self.lib.orc_set_weights(_p(self.weights, _f32p))
        L = self.lib
        L.orc_synth_frame.argtypes = [C.c_uint64, C.c_uint64, _u8p, _u8p]
        L.orc_synth_card.argtypes = [C.c_uint64, C.c_uint64, _u8p, _u8p]
        L.orc_line_by_shifting_origin.argtypes = [C.c_float, C.c_float, C.c_int, C.c_int, _f32p, _f32p]
        L.orc_parametric_intersect.argtypes = [C.c_float] * 4 + [_f32p, _f32p]
        L.orc_best_n_hseg_constrained.argtypes = [_f32p, C.c_int, C.c_float, C.c_float, C.c_float,
                                                  C.c_int, C.c_int, C.c_int, _u16p, _f32p, _f32p, _i32p]

    # ---- synthetic inputs ----
    def synth_frame(self, seed, idx):
        y = np.empty((480, 640), np.uint8)
        d = np.empty(16, np.uint8)
        self.lib.orc_synth_frame(seed, idx, _p(y, _u8p), _p(d, _u8p))
        return y, d

    def synth_card(self, seed, idx):
        c = np.empty((270, 428), np.uint8)
        d = np.empty(16, np.uint8)
        self.lib.orc_synth_card(seed, idx, _p(c, _u8p), _p(d, _u8p))
        return c, d

    # ---- cv ----
    def detection_boxes(self, w, h, orientation):
        b = np.zeros((4, 4), np.int32)
        self.lib.orc_detection_boxes(w, h, orientation, _p(b, _i32p))
        return b

    def sobel7(self, roi, want_dx):
        roi = np.ascontiguousarray(roi, np.uint8)
        h, w = roi.shape
        out = np.empty((h, w), np.int16)
        self.lib.orc_sobel7(_p(roi, _u8p), w, w, h, int(want_dx), _p(out, _i16p))
        return out

    def adaptive_canny7(self, dx, dy):
        h, w = dx.shape
        out = np.empty((h, w), np.uint8)
        lo, hi = C.c_int(), C.c_int()
        self.lib.orc_adaptive_canny7(_p(np.ascontiguousarray(dx), _i16p), _p(np.ascontiguousarray(dy), _i16p),
                                     w, h, _p(out, _u8p), C.byref(lo), C.byref(hi))
        return out, lo.value, hi.value

    def hough(self, edges, dx, dy, vertical):
        h, w = edges.shape
        rho, theta = C.c_float(), C.c_float()
        n, r, mx = C.c_int(), C.c_int(), C.c_int()
        f = self.lib.orc_hough(_p(np.ascontiguousarray(edges), _u8p), _p(np.ascontiguousarray(dx), _i16p),
                               _p(np.ascontiguousarray(dy), _i16p), w, h, int(vertical),
                               C.byref(rho), C.byref(theta), C.byref(n), C.byref(r), C.byref(mx))
        return f, rho.value, theta.value, n.value, r.value, mx.value

    def line_by_shifting_origin(self, rho, theta, xo, yo):
        r, t = C.c_float(), C.c_float()
        self.lib.orc_line_by_shifting_origin(rho, theta, xo, yo, C.byref(r), C.byref(t))
        return r.value, t.value

    def parametric_intersect(self, r1, t1, r2, t2):
        x, y = C.c_float(), C.c_float()
        ok = self.lib.orc_parametric_intersect(r1, t1, r2, t2, C.byref(x), C.byref(y))
        return ok, x.value, y.value

    def detect_edges(self, y, orientation=3, cb=None, cr=None):
        res = np.zeros(1, RESULT_DTYPE)
        h, w = y.shape
        y = np.ascontiguousarray(y)
        if cb is not None:
            cb, cr = np.ascontiguousarray(cb), np.ascontiguousarray(cr)
            self.lib.orc_detect_edges(_p(y, _u8p), w, w, h, _p(cb, _u8p), _p(cr, _u8p), cb.shape[1],
                                      orientation, res.ctypes.data_as(C.c_void_p))
        else:
            self.lib.orc_detect_edges(_p(y, _u8p), w, w, h, None, None, 0, orientation,
                                      res.ctypes.data_as(C.c_void_p))
        return res[0]

    def set_reference_flavour(self, flavour):
        """0: the scalar-Eigen build (default); 1: a stock x86-64 build (SSE2 order of the homography): process-wide"""
        self.lib.orc_set_reference_flavour(int(flavour))

    def calc_persp_transform(self, src_pts, dst_pts, sse=False):
        s = np.ascontiguousarray(src_pts, np.float32).reshape(8)
        d = np.ascontiguousarray(dst_pts, np.float32).reshape(8)
        m = np.empty(9, np.float32)
        (self.lib.orc_calc_persp_transform_sse if sse else self.lib.orc_calc_persp_transform)(_p(s, _f32p), _p(d, _f32p), _p(m, _f32p))
        return m

    def warp_perspective(self, src, m, dw=428, dh=270):
        src = np.ascontiguousarray(src, np.uint8)
        m = np.ascontiguousarray(m, np.float32)
        dst = np.empty((dh, dw), np.uint8)
        self.lib.orc_warp_perspective(_p(src, _u8p), src.shape[1], src.shape[1], src.shape[0],
                                      _p(m, _f32p), _p(dst, _u8p), dw, dw, dh)
        return dst

    def split_u8(self, interleaved):
        a = np.ascontiguousarray(interleaved, np.uint8)  # (h, w, 2)
        h, w = a.shape[:2]
        c1, c2 = np.empty((h, w), np.uint8), np.empty((h, w), np.uint8)
        self.lib.orc_split_u8(_p(a, _u8p), 2 * w, w, h, _p(c1, _u8p), _p(c2, _u8p))
        return c1, c2

    def deinterleave_rgba_to_r(self, rgba):
        a = np.ascontiguousarray(rgba, np.uint8).reshape(-1)
        out = np.empty(a.size // 4, np.uint8)
        self.lib.orc_deinterleave_rgba_to_r(_p(a, _u8p), _p(out, _u8p), out.size)
        return out

    def ycbcr_to_rgb(self, y, cb, cr, channels=3):
        y, cb, cr = (np.ascontiguousarray(v, np.uint8) for v in (y, cb, cr))
        out = np.empty(y.shape + (channels,), np.uint8)
        self.lib.orc_ycbcr_to_rgb(_p(y, _u8p), _p(cb, _u8p), _p(cr, _u8p), y.shape[-1], int(y.size // y.shape[-1]),
                                  channels, _p(out, _u8p))
        return out

    def blur_card(self, rgb, offsets, n_offsets, number_width, y_offset, unblur_digits):
        out = np.ascontiguousarray(rgb, np.uint8).copy()
        offs = np.ascontiguousarray(offsets, np.uint16)
        self.lib.orc_blur_card.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_float,
                                           C.c_int, C.c_int]
        self.lib.orc_blur_card(out.ctypes.data, out.shape[1], out.shape[0], out.shape[2], offs.ctypes.data,
                               int(n_offsets), float(number_width), int(y_offset), int(unblur_digits))
        return out

    def scoring_roi(self, w, h, use_full):
        rc = np.zeros(4, np.int32)
        self.lib.orc_scoring_roi(w, h, int(use_full), _p(rc, _i32p))
        return rc

    def focus_score(self, img, use_full=False):
        img = np.ascontiguousarray(img, np.uint8)
        self.lib.orc_focus_score.restype = C.c_float
        return np.float32(self.lib.orc_focus_score(_p(img, _u8p), img.shape[1], img.shape[1], img.shape[0], int(use_full)))

    def brightness_score(self, img, use_full=False):
        img = np.ascontiguousarray(img, np.uint8)
        self.lib.orc_brightness_score.restype = C.c_float
        return np.float32(self.lib.orc_brightness_score(_p(img, _u8p), img.shape[1], img.shape[1], img.shape[0], int(use_full)))

    def transform_card(self, plane, corners, orientation=3, truncate=False, upsample=False):
        plane = np.ascontiguousarray(plane, np.uint8)
        c = np.ascontiguousarray(corners, np.float32).reshape(8)
        card = np.empty((270, 428), np.uint8)
        self.lib.orc_transform_card(_p(plane, _u8p), plane.shape[1], plane.shape[1], plane.shape[0],
                                    _p(c, _f32p), orientation, int(truncate) | (2 if upsample else 0), _p(card, _u8p))
        return card

    # ---- scan ----
    def applym_vseg(self, x):
        x = np.ascontiguousarray(x, np.float32)
        o = np.empty(3, np.float32)
        self.lib.orc_applym_vseg(_p(x, _f32p), _p(o, _f32p))
        return o

    def applyc_digit(self, model, x):
        x = np.ascontiguousarray(x, np.float32)
        o = np.empty(10, np.float32)
        self.lib.orc_applyc_digit(model, _p(x, _f32p), _p(o, _f32p))
        return o

    def applym_slash(self, x):
        x = np.ascontiguousarray(x, np.float32)
        o = np.empty(2, np.float32)
        self.lib.orc_applym_slash(_p(x, _f32p), _p(o, _f32p))
        return o

    def applyc_expiry(self, x):
        x = np.ascontiguousarray(x, np.float32)
        o = np.empty(10, np.float32)
        l1, l2, l3 = np.empty(3500, np.float32), np.empty(120, np.float32), np.empty(176, np.float32)
        self.lib.orc_applyc_expiry(_p(x, _f32p), _p(o, _f32p), _p(l1, _f32p), _p(l2, _f32p), _p(l3, _f32p))
        return o, l1, l2, l3

    def vseg_row_features(self, row408):
        row = np.ascontiguousarray(row408, np.uint8)
        f = np.empty(204, np.float32)
        self.lib.orc_vseg_row_features(_p(row, _u8p), _p(f, _f32p))
        return f

    def best_segmentation_for_vseg_scores(self, visa, amex):
        v, a = np.ascontiguousarray(visa, np.float32), np.ascontiguousarray(amex, np.float32)
        s, y, p = C.c_float(), C.c_int(), C.c_int()
        self.lib.orc_best_segmentation_for_vseg_scores(_p(v, _f32p), _p(a, _f32p), C.byref(s), C.byref(y), C.byref(p))
        return s.value, y.value, p.value

    def best_n_vseg(self, card):
        card = np.ascontiguousarray(card, np.uint8)
        s, y, p = C.c_float(), C.c_int(), C.c_int()
        v, a = np.empty(270, np.float32), np.empty(270, np.float32)
        self.lib.orc_best_n_vseg(_p(card, _u8p), card.shape[1], C.byref(s), C.byref(y), C.byref(p),
                                 _p(v, _f32p), _p(a, _f32p))
        return s.value, y.value, p.value, v, a

    def hseg_grad_sums(self, strip):
        strip = np.ascontiguousarray(strip, np.uint8)
        s = np.empty(428, np.float32)
        self.lib.orc_hseg_grad_sums(_p(strip, _u8p), strip.shape[1], _p(s, _f32p))
        return s

    def best_n_hseg_constrained(self, sums, pattern, w, o, offsets, score, nw, po):
        sums = np.ascontiguousarray(sums, np.float32)
        offs = np.array(offsets, np.uint16)
        sc, nwc, poc = C.c_float(score), C.c_float(nw), C.c_int(po)
        self.lib.orc_best_n_hseg_constrained(_p(sums, _f32p), pattern, w[0], w[1], w[2], o[0], o[1], o[2],
                                             _p(offs, _u16p), C.byref(sc), C.byref(nwc), C.byref(poc))
        return offs, sc.value, nwc.value, poc.value

    def best_n_hseg(self, strip, pattern):
        """n_hseg.cpp:88-151 on a 27-row strip: (n_offsets, offsets, score, number_width, pattern_offset)"""
        strip = np.ascontiguousarray(strip, np.uint8)
        res = np.zeros(1, RESULT_DTYPE)
        self.lib.orc_best_n_hseg(_p(strip, _u8p), strip.shape[1], int(pattern), res.ctypes.data_as(C.c_void_p))
        r = res[0]
        return int(r["n_offsets"]), r["offsets"].copy(), r["hseg_score"], r["number_width"], int(r["pattern_offset"])

    def number_scores(self, strip, offsets, n):
        strip = np.ascontiguousarray(strip, np.uint8)
        offs = np.ascontiguousarray(offsets, np.uint16)
        s = np.empty((16, 10), np.float32)
        self.lib.orc_number_scores(_p(strip, _u8p), strip.shape[1], _p(offs, _u16p), int(n), _p(s, _f32p))
        return s

    def scan_card_image(self, card, warped=True, collect_card_number=True):
        card = np.ascontiguousarray(card, np.uint8)
        res = np.zeros(1, RESULT_DTYPE)
        res["flags"] = FLAG_WARPED if warped else 0
        self.lib.orc_scan_card_image_ex(_p(card, _u8p), card.shape[1], int(collect_card_number),
                                        res.ctypes.data_as(C.c_void_p))
        return res[0]

    def scan_card_image_at(self, card, y_off, pattern, score, base=None, collect_card_number=True):
        """the stages after vseg at a given segmentation (parity tests: downstream of a proven vseg near-tie)"""
        card = np.ascontiguousarray(card, np.uint8)
        res = np.zeros(1, RESULT_DTYPE)
        if base is not None:
            res[0] = base
        else:
            res["flags"] = FLAG_WARPED
        self.lib.orc_scan_card_image_at(_p(card, _u8p), card.shape[1], int(y_off), int(pattern), C.c_float(float(score)),
                                        int(collect_card_number), res.ctypes.data_as(C.c_void_p))
        return res[0]

    def scan_frame(self, y, orientation=3, truncate=False, want_card=True):
        y = np.ascontiguousarray(y, np.uint8)
        res = np.zeros(1, RESULT_DTYPE)
        card = np.zeros((270, 428), np.uint8)
        self.lib.orc_scan_frame(_p(y, _u8p), y.shape[1], y.shape[1], y.shape[0], orientation, int(truncate),
                                _p(card, _u8p) if want_card else None, res.ctypes.data_as(C.c_void_p))
        return res[0], card

    def passes_luhn(self, digits):
        d = np.ascontiguousarray(digits, np.uint8)
        return bool(self.lib.orc_passes_luhn(_p(d, _u8p), len(d)))

    # ---- expiry path ----
    def scharr3_dx_abs(self, img):
        img = np.ascontiguousarray(img, np.uint8)
        out = np.zeros(img.shape, np.int16)
        self.lib.orc_scharr3_dx_abs(_p(img, _u8p), img.shape[1], img.shape[1], img.shape[0],
                                    _p(out, _i16p), img.shape[1])
        return out

    def best_expiry_seg(self, card, y_offset):
        card = np.ascontiguousarray(card, np.uint8)
        out = np.zeros(1, EXPIRY_DTYPE)
        self.lib.orc_best_expiry_seg(_p(card, _u8p), card.shape[1], int(y_offset), out.ctypes.data_as(C.c_void_p))
        return out[0]

    def sort_order_desc(self, keys):
        """order[k] = index of the k-th element after std::sort(.., sum descending) as libstdc++ leaves it"""
        k = np.ascontiguousarray(keys, np.int64)
        o = np.empty(len(k), np.int32)
        self.lib.orc_sort_order_desc(len(k), k.ctypes.data_as(C.POINTER(C.c_long)), o.ctypes.data_as(C.POINTER(C.c_int)))
        return o

    def sort_heap_sorts(self):
        return int(self.lib.orc_sort_heap_sorts())

    def best_expiry_seg_sort_lists(self, card, y_offset):
        """the key lists (window sums per stripe, stripe sums) orc_best_expiry_seg hands to the sort, in call order"""
        card = np.ascontiguousarray(card, np.uint8)
        buf = np.zeros(4 * 430 + 300, np.int64)
        out = np.zeros(1, EXPIRY_DTYPE)
        self.lib.orc_expiry_capture_sort_lists(buf.ctypes.data_as(C.c_void_p), len(buf))
        self.lib.orc_best_expiry_seg(_p(card, _u8p), card.shape[1], int(y_offset), out.ctypes.data_as(C.c_void_p))
        used = int(self.lib.orc_expiry_captured_len())
        self.lib.orc_expiry_capture_sort_lists(None, 0)
        lists, i = [], 0
        while i < used:
            n = int(buf[i])
            lists.append(buf[i + 1:i + 1 + n].copy())
            i += 1 + n
        return lists

    def prepare_image_for_cat(self, card, left, top):
        card = np.ascontiguousarray(card, np.uint8)
        x = np.empty(176, np.float32)
        self.lib.orc_prepare_image_for_cat(_p(card, _u8p), card.shape[1], int(left), int(top), _p(x, _f32p))
        return x

    def scan_card_expiry(self, card, res):
        card = np.ascontiguousarray(card, np.uint8)
        r = np.zeros(1, RESULT_DTYPE)
        r[0] = res
        out = np.zeros(1, EXPIRY_DTYPE)
        self.lib.orc_scan_card_expiry(_p(card, _u8p), card.shape[1], r.ctypes.data_as(C.c_void_p),
                                      out.ctypes.data_as(C.c_void_p))
        return out[0]


class Reference:
    """Partial build of the reference's own code (oracle/_ref/libdmzref.so); None if absent."""

    @staticmethod
    def available(flavour=""):
        return os.path.exists(os.path.join(HERE, "_ref", "libdmzref%s.so" % flavour))

    def __init__(self, flavour=""):
        """flavour "" = Eigen's scalar paths (what the oracle restates); "_vec" = a stock x86-64 build (SSE2 packet paths)"""
        self.lib = C.CDLL(os.path.join(HERE, "_ref", "libdmzref%s.so" % flavour))
        L = self.lib
        L.ref_line_by_shifting_origin.argtypes = [C.c_float, C.c_float, C.c_int, C.c_int, _f32p, _f32p]
        L.ref_parametric_intersect.argtypes = [C.c_float] * 4 + [_f32p, _f32p]
        L.ref_best_n_hseg_constrained.argtypes = [_f32p, C.c_int, C.c_float, C.c_float, C.c_float,
                                                  C.c_int, C.c_int, C.c_int, _u16p, _f32p, _f32p, _i32p]

    def pass_kats(self):
        return self.lib.ref_pass_kats()

    def sort_order(self, keys, stripes=False):
        """the reference's std::sort of its CharacterRect / StripeSum lists with its own comparators (expiry_seg.cpp:456, 842)"""
        k = np.ascontiguousarray(keys, np.int64)
        o = np.empty(len(k), np.int32)
        fn = self.lib.ref_sort_stripe_sums if stripes else self.lib.ref_sort_rect_sums
        fn(len(k), k.ctypes.data_as(C.POINTER(C.c_long)), o.ctypes.data_as(C.POINTER(C.c_int)))
        return o

    def _model(self, fn, x, n):
        x = np.ascontiguousarray(x, np.float32)
        o = np.empty(n, np.float32)
        fn(_p(x, _f32p), _p(o, _f32p))
        return o

    def applym_vseg(self, x):
        return self._model(self.lib.ref_applym_vseg, x, 3)

    def applym_slash(self, x):
        return self._model(self.lib.ref_applym_slash, x, 2)

    def applyc_expiry(self, x):
        return self._model(self.lib.ref_applyc_expiry, x, 10)

    def applyc_digit(self, model, x):
        x = np.ascontiguousarray(x, np.float32)
        o = np.empty(10, np.float32)
        self.lib.ref_applyc_digit(model, _p(x, _f32p), _p(o, _f32p))
        return o

    def line_by_shifting_origin(self, rho, theta, xo, yo):
        r, t = C.c_float(), C.c_float()
        self.lib.ref_line_by_shifting_origin(rho, theta, xo, yo, C.byref(r), C.byref(t))
        return r.value, t.value

    def parametric_intersect(self, r1, t1, r2, t2):
        x, y = C.c_float(), C.c_float()
        ok = self.lib.ref_parametric_intersect(r1, t1, r2, t2, C.byref(x), C.byref(y))
        return ok, x.value, y.value

    def calc_persp_transform(self, src_pts, dst_pts):
        s = np.ascontiguousarray(src_pts, np.float32).reshape(8)
        d = np.ascontiguousarray(dst_pts, np.float32).reshape(8)
        m = np.empty(9, np.float32)
        self.lib.ref_calc_persp_transform(_p(s, _f32p), _p(d, _f32p), _p(m, _f32p))
        return m

    def card_dest_points(self):
        d = np.empty(8, np.float32)
        self.lib.ref_card_dest_points(_p(d, _f32p))
        return d

    def best_segmentation_for_vseg_scores(self, visa, amex):
        v, a = np.ascontiguousarray(visa, np.float32).copy(), np.ascontiguousarray(amex, np.float32).copy()
        s, y, p = C.c_float(), C.c_int(), C.c_int()
        self.lib.ref_best_segmentation_for_vseg_scores(_p(v, _f32p), _p(a, _f32p), C.byref(s), C.byref(y), C.byref(p))
        return s.value, y.value, p.value

    def best_n_hseg_constrained(self, sums, pattern, w, o, offsets, score, nw, po):
        sums = np.ascontiguousarray(sums, np.float32).copy()
        offs = np.array(offsets, np.uint16)
        sc, nwc, poc = C.c_float(score), C.c_float(nw), C.c_int(po)
        self.lib.ref_best_n_hseg_constrained(_p(sums, _f32p), pattern, w[0], w[1], w[2], o[0], o[1], o[2],
                                             _p(offs, _u16p), C.byref(sc), C.byref(nwc), C.byref(poc))
        return offs, sc.value, nwc.value, poc.value

    def passes_luhn(self, digits):
        d = np.ascontiguousarray(digits, np.uint8).copy()
        return bool(self.lib.ref_passes_luhn(_p(d, _u8p), len(d)))


SESSION_DTYPE = np.dtype([
    ("complete", "<i4"), ("complete_frame", "<i4"), ("number_frame", "<i4"), ("n_numbers", "<i4"),
    ("predictions", "u1", (16,)), ("card_type", "<i4"), ("expiry_month", "<i4"), ("expiry_year", "<i4"),
    ("count15", "<i4"), ("count16", "<i4"), ("usable_frames", "<i4"), ("n_expiry_groups", "<i4"),
    ("vseg_y_offset", "<i4"), ("n_offsets", "<i4"), ("offsets", "<u2", (16,)), ("number_width", "<f4"), ("reserved", "<i4", (6,)),
])
assert SESSION_DTYPE.itemsize == 128


def _scan_session(self, frames, expiry, scan_expiry=True, frame_interval_ms=0, now_year=2026, now_month=10,
                  allow_past=False):
    frames = np.ascontiguousarray(frames)
    assert frames.dtype == RESULT_DTYPE
    out = np.zeros(1, SESSION_DTYPE)
    xp = None
    if expiry is not None:
        expiry = np.ascontiguousarray(expiry)
        assert expiry.dtype == EXPIRY_DTYPE and len(expiry) == len(frames)
        xp = expiry.ctypes.data_as(C.c_void_p)
    self.lib.orc_scan_session(frames.ctypes.data_as(C.c_void_p), xp, len(frames), int(scan_expiry),
                              int(frame_interval_ms), int(now_year), int(now_month), int(allow_past),
                              out.ctypes.data_as(C.c_void_p))
    return out[0]


def _ref_scan_session(self, frames, expiry, scan_expiry=True):
    frames = np.ascontiguousarray(frames)
    out = np.zeros(1, SESSION_DTYPE)
    xp = None
    if expiry is not None:
        expiry = np.ascontiguousarray(expiry)
        xp = expiry.ctypes.data_as(C.c_void_p)
    self.lib.ref_scan_session.restype = None
    self.lib.ref_scan_session(frames.ctypes.data_as(C.c_void_p), xp, len(frames), int(scan_expiry),
                              out.ctypes.data_as(C.c_void_p))
    return out[0]


def _gather_into_groups(lib, prefix, lefts, sums, top, height):
    lefts = np.ascontiguousarray(lefts, np.int32)
    sums = np.ascontiguousarray(sums, np.int64)
    n = len(lefts)
    gn, gl, gw = (np.zeros(max(n, 1), np.int32) for _ in range(3))
    rl = np.zeros(max(n, 1), np.int32)
    rs = np.zeros(max(n, 1), np.int64)
    i64p = C.POINTER(C.c_int64)
    fn = getattr(lib, prefix + "expiry_gather_into_groups")
    fn.restype = C.c_int
    ng = fn(C.c_int(n), _p(lefts, _i32p), _p(sums, i64p), C.c_int(top), C.c_int(height), _p(gn, _i32p),
            _p(gl, _i32p), _p(gw, _i32p), _p(rl, _i32p), _p(rs, i64p))
    k = int(gn[:ng].sum())
    return gn[:ng].copy(), gl[:ng].copy(), gw[:ng].copy(), rl[:k].copy(), rs[:k].copy()


def _regrid_group(lib, prefix, sobel, top, height, left, width, character_width=9):
    sobel = np.ascontiguousarray(sobel, np.int16)
    assert sobel.shape == (270, 428)
    l, w, cw, n = C.c_int(left), C.c_int(width), C.c_int(character_width), C.c_int(0)
    rl = np.zeros(128, np.int32)
    rs = np.zeros(128, np.int64)
    fn = getattr(lib, prefix + "expiry_regrid_group")
    fn.restype = None
    fn(_p(sobel, _i16p), C.c_int(top), C.c_int(height), C.byref(l), C.byref(w), C.byref(cw), C.byref(n),
       _p(rl, _i32p), _p(rs, C.POINTER(C.c_int64)))
    return l.value, w.value, cw.value, rl[:n.value].copy(), rs[:n.value].copy()


Oracle.scan_session = _scan_session
Reference.scan_session = _ref_scan_session
Oracle.expiry_gather_into_groups = lambda self, *a: _gather_into_groups(self.lib, "orc_", *a)
Oracle.expiry_regrid_group = lambda self, *a: _regrid_group(self.lib, "orc_", *a)
Reference.expiry_gather_into_groups = lambda self, *a: _gather_into_groups(self.lib, "ref_", *a)
Reference.expiry_regrid_group = lambda self, *a: _regrid_group(self.lib, "ref_", *a)
