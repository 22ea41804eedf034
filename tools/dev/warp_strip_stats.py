"""Developer tool (CPU, oracle): how many strips of the synthetic corpus admit k_warp's fast / affine forms, and the
distribution of rho = |M7| / min|W| and alpha = 32 M / M7 (warp.hip: k_warp_windows)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as entry
orc = entry.load_oracle(); o = orc.Oracle()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = 0xCA4D10
rhos, alphas, fast, aff, tot = [], [], 0, 0, 0
for i in range(n):
    f, _ = o.synth_frame(seed, i)
    res, _ = o.scan_frame(f, want_card=False)
    if not res["found_all"]:
        continue
    c = res["corners"].astype(np.float32)  # tl, bl, tr, br pairs (landscape right order 0,2,1,3)
    sp = np.array([c[0], c[1], c[4], c[5], c[2], c[3], c[6], c[7]], np.float32)
    dp = np.array([0, 0, 427, 0, 0, 269, 427, 269], np.float32)
    m = o.calc_persp_transform(sp, dp).astype(np.float64).reshape(3, 3)
    M = np.linalg.inv(m)
    M1, M4, M6, M7, M8 = M[0, 1], M[1, 1], M[2, 0], M[2, 1], M[2, 2]
    for ty in range(3):
        for tx in range(7):
            x, y0 = tx * 64, ty * 90
            ws = [M6 * (x + cx) + M7 * cy + M8 for cx in (0, min(64, 428 - x) - 1) for cy in (y0, y0 + 89)]
            wmin = min(abs(w) for w in ws)
            tot += 1
            rho = abs(M7) / wmin
            rhos.append(rho)
            if abs(M7) * 2048 <= wmin:
                fast += 1
                ax, ay = 32 * M1 / M7, 32 * M4 / M7
                alphas.append(max(abs(ax), abs(ay)))
                if max(abs(ax), abs(ay)) <= 2 ** 28:
                    aff += 1
rhos = np.array(rhos); alphas = np.array(alphas)
print("strips", tot, "fast", fast / tot, "affine", aff / tot)
print("rho percentiles (log2):", np.round(np.log2(np.percentile(rhos, [1, 10, 50, 90, 99, 100])), 2))
print("alpha percentiles (log2):", np.round(np.log2(np.percentile(alphas, [0, 1, 10, 50, 90, 99, 100])), 2))
