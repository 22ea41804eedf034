"""The C++ host mirror of the reference API (card.io-dmz_amd/host/dmz.h): an SDK-style
call sequence compiled with g++ and run on the GPU, checked against the oracle."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "card.io-dmz_amd")


@pytest.mark.gpu
def test_sdk_style_call_sequence(tmp_path, oracle, ctx):
    exe = str(tmp_path / "host_api_demo")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(PKG, "host"),
                           os.path.join(ROOT, "tests", "host_api_demo.cpp"), "-o", exe,
                           "-L", PKG, "-ldmz_host", "-ldmz_hip", "-Wl,-rpath," + PKG])
    n = 5
    frames = np.stack([oracle.synth_frame(99, 0)[0]] * n)  # the same card five times: a session
    raw = tmp_path / "frames.raw"
    frames.tofile(raw)
    out = subprocess.check_output([exe, str(raw), str(n)], text=True).strip().splitlines()
    want, wcard = oracle.scan_frame(frames[0])
    wexp = oracle.scan_card_expiry(wcard, want)
    weights = (np.arange(428 * 270) % 251 + 1).astype(np.uint64)
    cardsum = int((wcard.reshape(-1).astype(np.uint64) * weights).sum())
    for i in range(n):
        t = out[i].split()
        assert t[3] == str(int(want["found_all"]))
        assert np.float32(t[7]) == want["corners"][0] and np.float32(t[8]) == want["corners"][1]
        assert np.float32(t[10]) == want["corners"][6] and np.float32(t[11]) == want["corners"][7]
        assert int(t[13]) == cardsum
        assert int(t[15]) == int(bool(want["flags"] & 1))
        assert int(t[19]) == int(want["vseg_y_offset"])
        assert abs(float(t[21]) - float(want["vseg_score"])) < 1e-4
        assert t[23] == "".join(str(int(d)) for d in want["digits"][: int(want["n_offsets"])])
        # expiry groups of the frame (scan_expiry = true): count and character rects as the oracle finds them
        if i >= 3 and int(t[25]) == 0 and int(wexp["n_groups"]) > 0:
            # the session settled on an expiry after three sightings: scan.cpp:44 stops scanning for it
            assert int(out[n + 1].split()[-3]) > 0
            continue
        assert int(t[25]) == int(wexp["n_groups"])
        txt = out[i].split("expiry_groups")[1]
        for g in range(int(wexp["n_groups"])):
            grp = wexp["groups"][g]
            want_txt = "[%d %d %d %d : %s]" % (grp["top"], grp["left"], grp["width"], grp["height"],
                                               " ".join("%d,%d" % (l, tp) for l, tp in zip(grp["char_left"], grp["char_top"])))
            assert want_txt in txt, (want_txt, txt)
    sess = out[n].split()
    usable = int(bool(want["flags"] & 1))
    # count15 / count16 (scan.cpp:66-75): the synthetic card may be of either number pattern
    fifteen = int(want["n_offsets"]) == 15
    assert int(sess[2]) == (usable * n if fifteen else 0) and int(sess[4]) == (0 if fifteen else usable * n)
    ex = out[n + 1].split()
    assert ex[0] == "expiry" and int(ex[2]) == usable
    if usable and int(wexp["n_groups"]) > 0:
        # five identical frames: every group of the frame has been seen five times
        decided = int(ex[-3]) > 0
        assert int(ex[4]) >= 1 and all(int(v) == n or (decided and int(v) >= 3) for v in ex[6:6 + int(ex[4])])
    # cython_scan_card_image (frame.cpp:84-98): usable / vseg / hseg of the first frame's card
    cy = out[n + 2].split()
    assert cy[0] == "cython" and int(cy[2]) == usable and int(cy[4]) == int(want["vseg_y_offset"])
    assert int(cy[6]) == int(want["pattern_type"]) and int(cy[8]) == int(want["n_offsets"])
    assert [int(v) for v in cy[10:]] == [int(v) for v in want["offsets"][: int(want["n_offsets"])]]
    # the Cython flavour's expiry entry points (dmz.h:105-119) on the same card
    seg = out[n + 3]
    assert seg.split()[0] == "cyseg"
    seg_only = oracle.best_expiry_seg(wcard, int(want["vseg_y_offset"]))
    assert int(seg.split()[1]) == int(seg_only["n_groups"])
    for g in range(int(seg_only["n_groups"])):
        grp = seg_only["groups"][g]
        want_txt = "[%d %d %d %d : %s]" % (grp["top"], grp["left"], grp["width"], grp["height"],
                                           " ".join("%d,%d" % (l, tp) for l, tp in zip(grp["char_left"], grp["char_top"])))
        assert want_txt in seg, (want_txt, seg)
    cat = out[n + 4].split()
    assert cat[0] == "cycat" and int(cat[1]) == int(seg_only["n_groups"])
    forced = want.copy()
    forced["flags"] = 5  # usable + vseg ok: the digits of every group are categorised
    wcat = oracle.scan_card_expiry(wcard, forced)
    k = 2
    for g in range(int(wcat["n_groups"])):
        for row in range(4):
            best, score = cat[k].split(":")
            k += 1
            assert int(best) == int(wcat["groups"][g]["scores"][row].argmax())
            assert abs(float(score) - float(wcat["groups"][g]["scores"][row].max())) <= 1e-4
    sess = out[n + 5].split()
    assert sess[0] == "cysession"
    if int(wcat["n_groups"]) > 0:
        assert int(sess[2]) >= 1  # three sightings of the same groups: aggregated, not tripled
        assert int(sess[2]) <= int(wcat["n_groups"])
    sch = oracle.scharr3_dx_abs(wcard[180:270]).astype(np.int64).reshape(-1)
    cys = out[n + 6].split()
    assert cys[0] == "cyscharr" and int(cys[1]) == int(sch.sum())
    assert int(cys[2]) == int((sch * (np.arange(sch.size) % 997 + 1)).sum())
    m = oracle.calc_persp_transform([106, 105, 533, 105, 106, 374, 533, 374], [0, 0, 427, 0, 0, 269, 427, 269])
    p = out[n + 7].split()
    assert np.float32(p[1]) == m[0] and np.float32(p[2]) == m[2] and np.float32(p[3]) == m[5]


def test_host_library_exports_reference_names(pkg):
    """CPU: the mirror exports the reference's entry point names (C++ linkage, like the reference)."""
    pkg.build()
    so = os.path.join(PKG, "libdmz_host.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", PKG, "libdmz_host.so"], stdout=subprocess.DEVNULL)
    syms = subprocess.check_output(["nm", "-D", "--defined-only", "-C", so], text=True)
    for name in ("dmz_context_create()", "dmz_context_destroy(dmz_context*)", "dmz_detect_edges(",
                 "dmz_transform_card(", "dmz_found_all_edges(", "llcv_unwarp(", "llcv_calc_persp_transform(",
                 "scanner_initialize(", "scanner_reset(", "scanner_add_frame(", "scanner_add_frame_with_expiry(",
                 "scanner_result(", "scanner_destroy(", "mz_create()", "mz_destroy(", "mz_prepare_for_backgrounding(",
                 "dmz_passes_luhn_checksum(", "dmz_card_info_for_prefix_and_length(", "dmz_prepare_for_backgrounding(",
                 # dmz_olm.h:82,101,104, dmz.h:60, processor_support.h:59-66
                 "dmz_scale_point(", "dmz_guide_frame(", "dmz_opposite_orientation(", "dmz_has_opencv()",
                 "dmz_has_neon_runtime()", "dmz_use_vfp3_16()", "dmz_use_gles_warp()", "dmz_set_gles_warp(int)",
                 "dmz_deinterleave_uint8_c2(", "dmz_deinterleave_RGBA_to_R(", "dmz_YCbCr_to_RGB(", "dmz_focus_score(",
                 "dmz_brightness_score(", "dmz_blur_card(", "dmz_create_point(", "dmz_create_rect(", "dmz_rect_get_points(",
                 # scan/frame.h:30-46
                 "scan_card_image(", "cython_scan_card_image(",
                 # the Cython flavour: dmz.h:105-119, mz.h:37-52
                 "dmz_scharr3_dx_abs(", "dmz_best_expiry_seg(", "dmz_expiry_extract(", "dmz_expiry_extract_group(",
                 "py_mz_create_from_cv_image_data(", "py_mz_release_ipl_image(", "py_mz_get_cv_image_data(",
                 "py_mz_cvSetImageROI(", "py_mz_cvResetImageROI("):
        assert name in syms, name
