"""The oracle's model forward passes against the reference's own embedded known-answer
vectors (tests/golden/model_kats.npz, extracted by tools/extract_models.py from
models/generated/*.cpp `#if TEST_GENERATED_MODELS` blocks).  Tolerance 1e-5 is the
reference's own (modelm_befe75da.cpp:1839, modelc_5c241121.cpp:2042)."""
import os

import numpy as np
import pytest

KATS = np.load(os.path.join(os.path.dirname(__file__), "golden", "model_kats.npz"))


def test_vseg_mlp_kat(oracle):
    assert np.abs(oracle.applym_vseg(KATS["vseg_in"]) - KATS["vseg_out"]).max() <= 1e-5


@pytest.mark.parametrize("idx,name", [(0, "5c241121"), (1, "01266c1b"), (2, "b00bf70c")])
def test_digit_cnn_kat(oracle, idx, name):
    out = oracle.applyc_digit(idx, KATS["digit_%s_in" % name])
    assert np.abs(out - KATS["digit_%s_out" % name]).max() <= 1e-5
    assert abs(out.sum() - 1.0) < 1e-5


def test_digit_kat_inputs_are_shared():
    # the three digit models embed the same test input
    assert np.array_equal(KATS["digit_5c241121_in"], KATS["digit_01266c1b_in"])
    assert np.array_equal(KATS["digit_5c241121_in"], KATS["digit_b00bf70c_in"])


def test_slash_mlp_kat(oracle):
    assert np.abs(oracle.applym_slash(KATS["slash_in"]) - KATS["slash_out"]).max() <= 1e-5


def test_expiry_cnn_kat_all_layers(oracle):
    out, l1, l2, l3 = oracle.applyc_expiry(KATS["expiry_in"])
    assert np.abs(l1 - KATS["expiry_l1"]).max() <= 1e-5
    assert np.abs(l2 - KATS["expiry_l2"]).max() <= 1e-5
    assert np.abs(l3 - KATS["expiry_l3"]).max() <= 1e-5
    assert np.abs(out - KATS["expiry_out"]).max() <= 1e-5
