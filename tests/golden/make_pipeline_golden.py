#!/usr/bin/env python3
"""Writes tests/golden/pipeline_golden.npz: the ORACLE's outputs for twenty seeded synthetic
frames (result records, expiry records, SHA-256 of each rectified card, one full card).  A regression
guard for the oracle and a committed target for the GPU parity test; regenerate only when
the oracle changes on purpose:  python tests/golden/make_pipeline_golden.py"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import orc  # noqa: E402

SEED, N = 20261001, 20  # (frame 18 is a 15-digit 4-6-5 card)


def main():
    o = orc.Oracle()
    recs = np.zeros(N, orc.RESULT_DTYPE)
    exps = np.zeros(N, orc.EXPIRY_DTYPE)
    hashes, digits = [], []
    card0 = None
    for i in range(N):
        y, d = o.synth_frame(SEED, i)
        recs[i], card = o.scan_frame(y)
        exps[i] = o.scan_card_expiry(card, recs[i])
        hashes.append(hashlib.sha256(card.tobytes()).hexdigest())
        digits.append(d)
        if i == 0:
            card0 = card
    np.savez_compressed(os.path.join(HERE, "pipeline_golden.npz"), seed=SEED, records=recs.view(np.uint8),
                        expiry=exps.view(np.uint8), card_sha256=np.array(hashes), true_digits=np.array(digits), card0=card0)
    print("wrote", N, "records")


if __name__ == "__main__":
    main()
