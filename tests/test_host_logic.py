"""Host-side logic of the C++ mirror (card.io-dmz_amd/host/dmz_host.cpp) that needs no GPU:
Luhn and the issuer-prefix table against the reference's own compiled code."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "card.io-dmz_amd")


class CardInfo(C.Structure):
    _fields_ = [("card_type", C.c_uint8), ("number_length", C.c_int), ("prefix_length", C.c_int),
                ("min_prefix", C.c_long), ("max_prefix", C.c_long)]


def _host():
    so = os.path.join(PKG, "libdmz_host.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", PKG], stdout=subprocess.DEVNULL)
    lib = C.CDLL(so)
    luhn = getattr(lib, "_Z24dmz_passes_luhn_checksumPhh")
    luhn.restype = C.c_bool
    luhn.argtypes = [C.c_void_p, C.c_uint8]
    info = getattr(lib, "_Z35dmz_card_info_for_prefix_and_lengthPhhb")
    info.restype = CardInfo
    info.argtypes = [C.c_void_p, C.c_uint8, C.c_bool]
    return luhn, info


def test_luhn_and_card_info_match_reference(reference):
    luhn, info = _host()
    reference.lib.ref_card_type.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int)]
    rng = np.random.default_rng(11)
    prefixes = [[4], [3, 4], [3, 7], [5, 1], [5, 5], [6, 0, 1, 1], [3, 5, 2, 8], [2, 2, 2, 1], [2, 7, 2, 0],
                [6, 2], [3, 0, 0], [3, 6], [9], [1], [6, 4, 4], [6, 5], [5, 0], [8, 8]]
    for t in range(600):
        n = [14, 15, 16][t % 3]
        d = rng.integers(0, 10, n).astype(np.uint8)
        p = prefixes[t % len(prefixes)]
        d[: len(p)] = p
        assert bool(luhn(d.ctypes.data, n)) == reference.passes_luhn(d)
        for incomplete in (False, True):
            m = n if not incomplete else int(rng.integers(1, n + 1))
            nl = C.c_int()
            want_type = reference.lib.ref_card_type(d.ctypes.data, m, int(incomplete), C.byref(nl))
            got = info(d.ctypes.data, m, incomplete)
            assert (got.card_type, got.number_length) == (want_type, nl.value), (d[:m], incomplete)


def test_expiry_session_aggregation_matches_reference(reference):
    """expiry_aggregate_grouped_rects + get_stable_expiry_month_and_year (expiry_categorize.cpp:162-330) on
    replayed sessions: groups drift by a few pixels, appear/disappear, carry noisy one-hot-ish scores."""
    host = C.CDLL(os.path.join(PKG, "libdmz_host.so"))
    getattr(host, "_Z30dmz_hip_host_allow_past_expiryb")(C.c_bool(True))  # the reference build is the CYTHON flavour
    i16p, f32p, i32p = C.POINTER(C.c_int16), C.POINTER(C.c_float), C.POINTER(C.c_int)
    for fn in (host.dmz_hip_host_expiry_session_replay, reference.lib.ref_expiry_session_replay):
        fn.argtypes = [C.c_int, i32p, i16p, i16p, f32p, i32p, i32p, i32p]
    rng = np.random.default_rng(23)
    decided = 0
    for session in range(60):
        n_frames = int(rng.integers(3, 14))
        truth = rng.integers(0, 10, (3, 4))  # up to three persistent groups with their own digits
        truth[:, 0] = rng.integers(0, 2, 3)  # month tens digit 0/1 so that plausible dates occur
        truth[:, 2] = rng.integers(1, 4, 3)
        base_top = rng.integers(180, 250, 3)
        base_left = rng.integers(30, 300, 3)
        per_frame, tops, lefts, scores = [], [], [], []
        for f in range(n_frames):
            k = 0
            for g in range(3):
                if rng.random() < 0.7:
                    reps = 2 if rng.random() < 0.15 else 1  # sometimes the same group twice in one frame
                    for _ in range(reps):
                        tops.append(base_top[g] + rng.integers(-3, 4))
                        lefts.append(base_left[g] + rng.integers(-3, 4))
                        s = rng.random((4, 10)).astype(np.float32) * 0.08
                        for c in range(4):
                            s[c, truth[g, c]] += 0.9 if rng.random() < 0.85 else 0.2
                        scores.append(s / s.sum(1, keepdims=True))
                        k += 1
            per_frame.append(k)
        gpf = np.array(per_frame, np.int32)
        t = np.array(tops if tops else [0], np.int16)
        l = np.array(lefts if lefts else [0], np.int16)
        sc = np.ascontiguousarray(np.array(scores if scores else np.zeros((1, 4, 10)), np.float32))
        outs = []
        for fn in (host.dmz_hip_host_expiry_session_replay, reference.lib.ref_expiry_session_replay):
            m, y, na = (np.zeros(n_frames, np.int32) for _ in range(3))
            fn(n_frames, gpf.ctypes.data_as(i32p), t.ctypes.data_as(i16p), l.ctypes.data_as(i16p),
               sc.ctypes.data_as(f32p), m.ctypes.data_as(i32p), y.ctypes.data_as(i32p), na.ctypes.data_as(i32p))
            outs.append((m, y, na))
        for a, b in zip(outs[0], outs[1]):
            assert np.array_equal(a, b), session
        decided += int(outs[0][0][-1] > 0)
    assert decided >= 5  # the sessions do reach expiry decisions


class Point(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float)]


class Rect(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("w", C.c_float), ("h", C.c_float)]


def test_guide_frame_scale_point_and_orientation_match_reference(reference):
    """dmz_scale_point / dmz_guide_frame / dmz_opposite_orientation (dmz_olm.cpp:20-23,134-179) and the
    processor_support.h switches, against the reference's own compiled code, float bits equal."""
    _host()
    host = C.CDLL(os.path.join(PKG, "libdmz_host.so"))
    scale = getattr(host, "_Z15dmz_scale_point9dmz_point8dmz_rectS0_")
    scale.restype, scale.argtypes = Point, [Point, Rect, Rect]
    guide = getattr(host, "_Z15dmz_guide_framehff")
    guide.restype, guide.argtypes = Rect, [C.c_uint8, C.c_float, C.c_float]
    opposite = getattr(host, "_Z24dmz_opposite_orientationh")
    opposite.restype, opposite.argtypes = C.c_uint8, [C.c_uint8]
    f32p = C.POINTER(C.c_float)
    ref = reference.lib
    ref.ref_scale_point.argtypes = [f32p, f32p, f32p, f32p]
    ref.ref_guide_frame.argtypes = [C.c_int, C.c_float, C.c_float, f32p]
    rng = np.random.default_rng(5)
    for _ in range(2000):
        p = rng.uniform(-50, 700, 2).astype(np.float32)
        src = rng.uniform(1, 640, 4).astype(np.float32)
        dst = rng.uniform(1, 1920, 4).astype(np.float32)
        want = np.zeros(2, np.float32)
        ref.ref_scale_point(p.ctypes.data_as(f32p), src.ctypes.data_as(f32p), dst.ctypes.data_as(f32p),
                            want.ctypes.data_as(f32p))
        got = scale(Point(*p), Rect(*src), Rect(*dst))
        assert np.array_equal(np.array([got.x, got.y], np.float32).view(np.uint32), want.view(np.uint32))
    for orientation in range(0, 7):
        assert opposite(orientation) == ref.ref_opposite_orientation(orientation)
        for w, h in ((320, 480), (480, 320), (1080, 1920), (1279.5, 719.25), (1, 1), (0, 0)):
            want = np.zeros(4, np.float32)
            ref.ref_guide_frame(orientation, w, h, want.ctypes.data_as(f32p))
            got = guide(orientation, w, h)
            assert np.array_equal(np.array([got.x, got.y, got.w, got.h], np.float32).view(np.uint32),
                                  want.view(np.uint32)), (orientation, w, h)
    # the non-NEON flavour of processor_support.cpp:112-118; the warp switch reports the accelerator instead
    assert host._Z20dmz_has_neon_runtimev() == ref.ref_processor_support(0) == 0
    assert host._Z15dmz_use_vfp3_16v() == ref.ref_processor_support(1) == 0
    assert host._Z14dmz_has_opencvv() == 1
    has_gpu = host._Z19dmz_has_hip_runtimev()
    assert host._Z17dmz_use_gles_warpv() == int(bool(has_gpu))
    host._Z17dmz_set_gles_warpi(0)
    assert host._Z17dmz_use_gles_warpv() == 0
    host._Z17dmz_set_gles_warpi(1)
    assert host._Z17dmz_use_gles_warpv() == int(bool(has_gpu))


def test_scanner_state_keeps_the_reference_field_order(tmp_path):
    """ScannerState mirrors scan/scan.h:33-48 field for field (session_analytics included), the HIP context pointer
    comes last; reference-style Eigen accessors on the score types compile."""
    src = tmp_path / "layout.cpp"
    src.write_text('''
#include <stddef.h>
#include <stdio.h>
#include "dmz.h"
int main() {
  static_assert(offsetof(ScannerState, count15) < offsetof(ScannerState, count16), "");
  static_assert(offsetof(ScannerState, count16) < offsetof(ScannerState, aggregated15), "");
  static_assert(offsetof(ScannerState, aggregated15) < offsetof(ScannerState, aggregated16), "");
  static_assert(offsetof(ScannerState, aggregated16) < offsetof(ScannerState, session_analytics), "");
  static_assert(offsetof(ScannerState, session_analytics) < offsetof(ScannerState, successfulCardNumberResult), "");
  static_assert(offsetof(ScannerState, successfulCardNumberResult) < offsetof(ScannerState, mostRecentUsableHSeg), "");
  static_assert(offsetof(ScannerState, mostRecentUsableHSeg) < offsetof(ScannerState, mostRecentUsableVSeg), "");
  static_assert(offsetof(ScannerState, mostRecentUsableVSeg) < offsetof(ScannerState, timeOfCardNumberCompletionInMilliseconds), "");
  static_assert(offsetof(ScannerState, timeOfCardNumberCompletionInMilliseconds) < offsetof(ScannerState, scan_expiry), "");
  static_assert(offsetof(ScannerState, scan_expiry) < offsetof(ScannerState, expiry_month), "");
  static_assert(offsetof(ScannerState, expiry_month) < offsetof(ScannerState, expiry_year), "");
  static_assert(offsetof(ScannerState, expiry_year) < offsetof(ScannerState, expiry_groups), "");
  static_assert(offsetof(ScannerState, expiry_groups) < offsetof(ScannerState, name_groups), "");
  static_assert(offsetof(ScannerState, name_groups) < offsetof(ScannerState, dmz), "");
  static_assert(sizeof(NumberScores) == 16 * 10 * sizeof(float), "");
  NumberScores s; s.setZero(); s(3, 4) = 2.0f;
  NumberPredictions p; p(5, 0) = 7;
  ScannerState st; scanner_initialize(&st);
  printf("%g %ld %d %d %u\\n", s.sum(), p(5, 0), (int)s.rows(), (int)s.cols(), st.session_analytics.num_frames_scanned);
  dmz_rect g = dmz_guide_frame(FrameOrientationPortrait, 320, 480);
  dmz_point q = dmz_scale_point(dmz_create_point(1, 2), g, g);
  return (q.x == 1 && dmz_opposite_orientation(FrameOrientationLandscapeLeft) == FrameOrientationLandscapeRight) ? 0 : 1;
}
''')
    _host()
    exe = str(tmp_path / "layout")
    subprocess.check_call(["g++", "-std=c++17", "-Wno-invalid-offsetof", "-I", os.path.join(PKG, "host"), str(src), "-o", exe,
                           "-L", PKG, "-ldmz_host", "-ldmz_hip", "-Wl,-rpath," + PKG])
    out = subprocess.check_output([exe], text=True).split()
    assert out == ["2", "7", "16", "10", "0"]
