#!/usr/bin/env python3
"""Developer model (CPU): how often is the hseg search's arg-min decided by less than the rounding noise of the reference's
sequential float sum?  For every candidate of the four passes: the reference's float score, the real (double) value of the same
sum, the digit-position signature.  A pass is AMBIGUOUS when a candidate with another signature lies within
eps = 2 * 429 u S of the best real score (u = 2^-24).   usage: tools/dev/hseg_gap_model.py [cards] [seed]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import orc

T = np.array([0.26228655, 0.30289554, 0.34632607, 0.38725636, 0.42745813, 0.45875135, 0.46498017, 0.45258447, 0.43045216,
              0.42430462, 0.44796554, 0.47726529, 0.48471646, 0.46457738, 0.42799847, 0.38851183, 0.33966308, 0.28802608,
              0.25377602], np.float32)
PAT = {1: [1, 1, 1, 1, 0, 1, 1, 1, 1, 0, 1, 1, 1, 1, 0, 1, 1, 1, 1], 2: [1, 1, 1, 1, 0, 1, 1, 1, 1, 1, 1, 0, 1, 1, 1, 1, 1]}
U = 2.0 ** -24


def candidates(pt, wmin, wmax, wstep, omin, omax, ostep):
    plen = len(PAT[pt])
    out = []
    w = np.float32(wmin)
    while w < np.float32(wmax):
        pw = np.float32(plen) * w
        mmax = (428 - int(np.rint(pw))) & 0xFFFF
        pom = omax & 0xFFFF
        if pom == 0xFFFF or pom > mmax:
            pom = mmax
        off = omin
        while off < pom:
            cs = [(off + int(np.rint(np.float32(pi) * w))) & 0xFFFF for pi in range(plen) if PAT[pt][pi]]
            if all(c + 19 < 428 for c in cs):
                out.append((float(w), off, tuple(cs)))
            off += ostep
        w = np.float32(w + np.float32(wstep))
    return out


def scores(g, cs):
    pat = np.zeros(428, np.float32)
    for c in cs:
        pat[c:c + 19] = T
    terms = np.abs(g - pat)  # float32, as the reference
    return float(np.add.accumulate(terms, dtype=np.float32)[-1]), float(terms.astype(np.float64).sum())


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed = int(sys.argv[2], 0) if len(sys.argv) > 2 else 0xCA4D10
    o = orc.Oracle()
    passes = amb = wrong = 0
    gaps = []
    for i in range(n):
        card, _ = o.synth_card(seed, i)
        s, y, p, _, _ = o.best_n_vseg(card)
        if p == 0 or y < 0 or y + 27 > 270:
            continue
        g = o.hseg_grad_sums(card[y:y + 27])
        best = (428.0, 428.0, None, 0.0, 0)  # float score, real score, signature, width, offset
        spec = [(17.1, 19.7, 0.5, None), (0.5, 0.2, 10), (0.2, 0.1, 3), (0.1, 0.05, 3)]
        for k in range(4):
            if k == 0:
                cl = candidates(p, np.float32(17.1), np.float32(19.7), np.float32(0.5), 0, 0xFFFF, 10)
            else:
                d, st, r = spec[k]
                bw, po = np.float32(best[3]), best[4]
                cl = candidates(p, bw - np.float32(d), bw + np.float32(d), np.float32(st), 0 if po < r else po - r, po + r, 1)
            sc = [scores(g, c[2]) for c in cl]
            # the reference's decision
            ref = best
            for c, (sf, sr) in zip(cl, sc):
                if sf < ref[0]:
                    ref = (sf, sr, c[2], c[0], c[1])
            # the filtered decision: real scores, incumbent first
            allc = [(best[1], best[2])] + [(sr, c[2]) for c, (sf, sr) in zip(cl, sc)]
            m = min(a[0] for a in allc)
            eps = 2 * 429 * U * m * 1.01 + 1e-6
            near = [a for a in allc if a[0] <= m + eps]
            sigs = set(a[1] for a in near)
            passes += 1
            if len(sigs) > 1:
                amb += 1
            else:
                # the earliest holder of that signature wins; check it is the reference's
                if next(iter(sigs)) != ref[2]:
                    wrong += 1
            others = [a[0] for a in allc if a[1] != ref[2]]
            if others:
                gaps.append((min(others) - ref[1]) / max(ref[1], 1e-9))
            best = ref
    gaps = np.array(gaps)
    print(f"passes {passes} ambiguous {amb} wrong-when-clear {wrong}")
    print("relative gap to the best other signature: percentiles 0.1/1/5/50:", np.percentile(gaps, [0.1, 1, 5, 50]))
    print("gaps below 1e-4:", int((gaps < 1e-4).sum()), "below 1e-3:", int((gaps < 1e-3).sum()))


main()
