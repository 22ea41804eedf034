#!/usr/bin/env python3
"""One-time extraction of model DATA (weights + known-answer vectors) from the
reference's generated model files.

The reference ships its trained models as little-endian fp32 values spelled as
`static uint8_t data_<id>[N] = { 0x.., ... }; // <label>` initialisers
(/root/reference/models/generated/*.cpp, /root/reference/models/expiry/*.cpp;
blob map: SURVEY.md Appendix C).  This script reads only those byte arrays --
no code -- and writes

  card.io-dmz_amd/weights/dmz_models.bin   weights, layout = struct dmz_weights
                                           (card.io-dmz_amd/csrc/dmz_weights.h)
  tests/golden/model_kats.npz              the reference's embedded known-answer
                                           test inputs / expected outputs

Run in the build container only (needs /root/reference); both outputs are
committed, so nothing at test/bench time reads the reference.
"""
import os
import re
import struct
import sys

import numpy as np

REF = os.environ.get("DMZ_REFERENCE", "/root/reference")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

ARRAY_RE = re.compile(
    r"static\s+uint8_t\s+(data_\w+)\[(\d+)\][^=]*=\s*\{\s*//\s*([^\n]*)\n(.*?)\};",
    re.DOTALL)


def read_arrays(path):
    """-> list of (label, float32 ndarray) in file order."""
    text = open(path).read()
    out = []
    for m in ARRAY_RE.finditer(text):
        n = int(m.group(2))
        label = m.group(3).strip()
        raw = bytes(int(tok, 16) for tok in re.findall(r"0x[0-9A-Fa-f]{2}", m.group(4)))
        assert len(raw) == n, (path, m.group(1), len(raw), n)
        out.append((label, np.frombuffer(raw, dtype="<f4").copy()))
    return out


def take(arrs, label, size, idx=0):
    hits = [a for (l, a) in arrs if l == label]
    a = hits[idx]
    assert a.size == size, (label, a.size, size)
    return a


def main():
    gen = os.path.join(REF, "models", "generated")
    exp = os.path.join(REF, "models", "expiry")
    kats = {}
    parts = []

    # vseg MLP 204-50-3 (modelm_befe75da.cpp:16,1719,1731,1759)
    a = read_arrays(os.path.join(gen, "modelm_befe75da.cpp"))
    parts += [take(a, "hidden W", 50 * 204), take(a, "hidden b", 50),
              take(a, "logistic W", 3 * 50), take(a, "logistic b", 3)]
    kats["vseg_in"] = take(a, "test input", 204)
    kats["vseg_out"] = take(a, "test output", 3)

    # three digit CNNs (modelc_*.cpp:22,37,42,1752,1761,1818), in the order
    # n_categorize.cpp:49-51 applies them
    for name in ("5c241121", "01266c1b", "b00bf70c"):
        a = read_arrays(os.path.join(gen, "modelc_%s.cpp" % name))
        parts += [take(a, "conv W", 72), take(a, "conv b", 8),
                  take(a, "hidden W", 32 * 320), take(a, "hidden b", 32),
                  take(a, "logistic W", 10 * 32), take(a, "logistic b", 10)]
        kats["digit_%s_in" % name] = take(a, "test input", 27 * 19)
        kats["digit_%s_out" % name] = take(a, "test output", 10)

    # slash MLP 176-80-2 (modelm_730c4cbd.cpp:19,2369,2393,2423)
    a = read_arrays(os.path.join(exp, "modelm_730c4cbd.cpp"))
    parts += [take(a, "hidden W", 80 * 176), take(a, "hidden b", 80),
              take(a, "logistic W", 2 * 80), take(a, "logistic b", 2)]
    kats["slash_in"] = take(a, "test input", 176)
    kats["slash_out"] = take(a, "test output", 2)

    # expiry CNN (modelc_bf4dd6c8.cpp:24,236,265,8602,8628,12151,12185,12482)
    a = read_arrays(os.path.join(exp, "modelc_bf4dd6c8.cpp"))
    parts += [take(a, "conv W", 50 * 25, 0), take(a, "conv b", 50, 0),
              take(a, "conv W", 40 * 50 * 25, 1), take(a, "conv b", 40, 1),
              take(a, "hidden W", 176 * 120), take(a, "hidden b", 176),
              take(a, "logistic W", 10 * 176), take(a, "logistic b", 10)]
    kats["expiry_in"] = take(a, "test input", 16 * 11)
    kats["expiry_l1"] = take(a, "test output layer 1", 50 * 70)
    kats["expiry_l2"] = take(a, "test output layer 2", 40 * 3)
    kats["expiry_l3"] = take(a, "test output layer 3", 176)
    kats["expiry_out"] = take(a, "test output", 10)

    blob = np.concatenate(parts).astype("<f4")
    wpath = os.path.join(ROOT, "card.io-dmz_amd", "weights", "dmz_models.bin")
    with open(wpath, "wb") as f:
        f.write(b"DMZW0001")
        f.write(struct.pack("<II", blob.size, 0))
        f.write(blob.tobytes())
    kpath = os.path.join(ROOT, "tests", "golden", "model_kats.npz")
    np.savez(kpath, **kats)
    print("wrote %s (%d floats) and %s (%d vectors)" % (wpath, blob.size, kpath, len(kats)))


if __name__ == "__main__":
    sys.exit(main())
