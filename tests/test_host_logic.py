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
