#!/usr/bin/env python3
"""Regenerates the per-kernel tables of DESIGN.md section 5.1 from the committed profile the tag in profiles/CURRENT names:
   python tools/design_tables.py            # prints the tables
   python tools/design_tables.py --write    # replaces the text between the GENERATED markers of DESIGN.md
Sources: <tag>_bench_batch65536.json (stage times by hipEvent, gates, headline), <tag>_kernel_stats_batch65536.csv (rocprofv3
--kernel-trace --stats), <tag>_pmc_{FETCH,WRITE}_SIZE_batch4096.txt, <tag>_pmc_SQ_insts_batch8192.txt, <tag>_pmc_SQ_issue_batch8192.txt
(separate --pmc passes, tools/profile_round.sh).  What no counter gives -- what a kernel replaces, its workgroup residency and
the hand-counted floor of its formulation -- is the static table below."""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (ALGO, STAGE_KERNELS and the readers of the PMC summaries)

# kernel -> (file, SURVEY 8(a) rows, waves per CU, floor of the formulation in VALU instructions per frame with its derivation)
META = {
    "k_detect_walk": ("detect.hip", "a3-a7", "28 (hz) / 32 (vert)",
                      "~36 000 (round 5): walk 17 000 (22 per wave-step: 2 alignbyte + 4 dot4 + 1 across, 11 additions for the binomial "
                      "cascades along the walk, 4 to saturate / pack / accumulate |dx|+|dy|), NMS 5 000 (3 per wave-step with no pixel above the "
                      "low threshold, ~35 where one passes), votes 9 000 (map scan 1 500; ~125 voting passes of ~60 per frame), tile load 1 700, "
                      "thresholds / map zeroing / hysteresis 2 500, arg-max over the reachable rho bins 600"),
    "k_geometry": ("geometry.hip", "a8-a10", "1 wave per 64 frames", "-"),
    "k_homography": ("geometry.hip", "a11", "1 wave per 64 frames", "- (round 5: unrolled into registers, evaluated twice; 0.09 -> 0.03 ms)"),
    "k_warp_windows": ("warp.hip", "a12", "-", "-"),
    "k_warp": ("warp.hip", "a12", "27",
               "19.5 per pixel in the loop = 35 200 (round 5: 6 fp64 where the coordinates are affine in the extrapolated reciprocal, 5 where the "
               "reciprocal is linear; 1.5 filter; 12 integer: 3 address, 2 fractions, 1 weight pair, 2 tap packs, 3 horizontal lerp, 1 dot2) + 3.7 "
               "per pixel set-up per wave (window staging, column terms, reciprocal start-up)"),
    "k_vseg": ("vseg.hip", "a13-a18", "28", "~13 000: row features ~60 per row-wave (38 packed-u16 gradient + 18 for the two wave reductions), A operands 0.5 per feature, tanh + 50->3 layer 2 500, scans + norms + softmax 3 000"),
    "k_hseg": ("hseg.hip", "a19-a20", "16", "~4 400: cross gradient 1 900, score table 560, five table-score batches 1 300, the winner's ordered sum 600"),
    "k_digit_patches": ("digits.hip", "a21-a22", "32", "~2 200: 5 per pixel gradient x 36 px per lane + LUT scan"),
    "k_digits": ("digits.hip", "a23", "16", "~4 600 (1 150 per wave; round 6): 90 tile packs, 320 v_max3 for the nine-tile max-pool, 140 tanh x 5 "
                 "(n-tile 1's row-tiles merged), 70 f16 splits x 3, 20 DPP moves; tail 250"),
    "k_expiry_stripes": ("expiry.hip", "a25", "32", "- (HBM: 92 scattered 428-byte rows per card)"),
    "k_expiry_seg": ("expiry.hip", "a25", "12", "- (data-dependent list logic; + the library-order emulation on 19 % of the stripes)"),
    "k_expiry_cat": ("expiry.hip", "a26", "12", "- (MFMA-paced: conv1 A operands 24 per tile, epilogues 56 per tile)"),
}


# stage -> (floor of the formulation in VALU instructions per frame or None = "the measured count" for the data-dependent list /
# MFMA-paced kernels, the exact-semantics requirement that sets it)
CEILING = {
    "detect": (36000, "int16-saturating Sobel-7 on `v_dot4_u32_u8` + binomial cascades (22 per wave-step), per-pixel NMS branches as the reference takes them, LDS-atomic Hough votes that serialise per shared counter"),
    "geometry": (None, "-"),
    "warp": (39500, "cvWarpPerspective's fp64 association per pixel (6 fp64 for a filtered-exact coordinate pair: reciprocal by extrapolation + Newton, two fma) + 12 integer for the 5-bit bilinear blend from four byte taps; 3.7 per pixel of per-wave set-up"),
    "vseg": (13000, "408-column row features with two wave reductions per row (float min / max / sum in the reference's order), exact bf16 operand splits for the hidden layer"),
    "hseg": (4400, "428-term sequential float sums per candidate (bit-exact `hseg_score`), the reference's four passes"),
    "digits": (6800, "5-tap cross gradient + exact 256-bin equalisation per digit; nine-tile max-pool, 140 tanh and the f16 split of their values per wave on the CNN side"),
    "expiry_seg": (None, "data-dependent list logic in the reference's visiting order (incl. the `std::sort` tie order on 19 % of the stripes); round 6 removed the work that was not needed (lazy trimming)"),
    "expiry_cat": (None, "MFMA-paced: operand builds (24 per conv1 tile) and epilogues (56 per tile) around f16 x 3 products that keep 1e-5"),
}
ISSUE_CEILING = 256 * 4 * 2.4e9 / 4 * 0.966  # wave64 VALU instructions / s at the half-rate class's measured saturation


def kernel_stats(tag):
    out = {}
    path = os.path.join(ROOT, "profiles", "%s_kernel_stats_batch65536.csv" % tag)
    if not os.path.exists(path):
        return out
    for r in csv.DictReader(open(path)):
        m = re.search(r"(k_\w+)(<(true|false))?", r["Name"])
        if not m:
            continue
        name = m.group(1)
        if name == "k_detect_walk":  # the two template instances of the benchmark's boxes: top / bottom (hz) and left / right (vert)
            name += "<hz>" if m.group(3) == "false" else "<vert>"
        e = out.setdefault(name, [0, 0.0])
        e[0] += int(r["Calls"])
        e[1] += float(r["TotalDurationNs"])
    return {k: (c, t / c / 1e6) for k, (c, t) in out.items()}


def main():
    tag = open(os.path.join(ROOT, "profiles", "CURRENT")).read().split()[0]
    b = json.load(open(os.path.join(ROOT, "profiles", "%s_bench_batch65536.json" % tag)))
    stats = kernel_stats(tag)
    pmc, _ = bench.load_pmc_traffic()
    valu = bench.load_pmc_valu() or {}
    issue = bench.load_pmc_issue() or {}
    L = []
    L.append("Generated by `tools/design_tables.py` from `profiles/%s_*` (the tag in `profiles/CURRENT`); do not edit by hand.\n" % tag)
    L.append("Headline of that profile: **%.2f M frames/s, %.2f ms per step** (65 536 frames, three queues; stage times below: one queue, "
             "per-kernel hipEvents; boxes differ by +- 1.5 %%).\n" % (b["value"] / 1e6, b["ms_per_step"]))
    L.append("| stage (kernels) | file | replaces (SURVEY 8a) | ms per 65 536 frames (hipEvent; rocprofv3 kernel average) | algorithmic B / frame | HBM fetched + written B / frame (PMC) | VALU instr / frame | VALU active / busy (half-rate ceiling 0.966, full-rate 1.807) | matrix pipe busy | waves / CU |")
    L.append("|---|---|---|---|---|---|---|---|---|---|")
    tot_ms = tot_valu = tot_tr = tot_act = tot_busy = 0.0
    for stage, kernels in bench.STAGE_KERNELS.items():
        if stage not in b["stages"]:
            continue
        st = b["stages"][stage]
        ks = [k for k in kernels if k in META]
        avg = " + ".join("%.2f" % stats[k][1] for k in (["k_detect_walk<hz>", "k_detect_walk<vert>"] if stage == "detect" else ks)
                         if k in stats) or "-"
        tr = bench.stage_traffic(pmc, stage)
        sv = bench.stage_valu(valu, stage)
        si = bench.stage_issue(issue, stage)
        tot_ms += st["ms_per_step"]
        tot_valu += sv or 0
        tot_tr += (tr[0] + tr[1]) if tr else 0
        tot_act += si["_act"] if si else 0
        tot_busy += si["_busy"] if si else 0
        L.append("| %s (%s) | %s | %s | **%.2f**; %s | %d | %s | %s | %s | %s | %s |" % (
            stage, ", ".join("`%s`" % k for k in ks), META[ks[0]][0], ", ".join(sorted({META[k][1] for k in ks})),
            st["ms_per_step"], avg, bench.ALGO[stage][0],
            "%d + %d (x %.2f)" % (tr[0], tr[1], (tr[0] + tr[1]) / bench.ALGO[stage][0]) if tr else "-",
            "%d" % sv if sv else "-", "%.2f" % si["valu_active_per_busy"] if si else "-",
            "%.2f" % si["mfma_busy_frac"] if si else "-", " / ".join(META[k][2] for k in ks)))
    L.append("| **pipeline** | | | **%.2f** (sum of the stages on one queue) | %d | %s | %d | %s | | |" % (
        tot_ms, b["config"]["algorithmic_bytes_per_unit"], "%d (x %.2f)" % (tot_tr, tot_tr / b["config"]["algorithmic_bytes_per_unit"]),
        tot_valu, "%.3f" % (tot_act / tot_busy) if tot_busy else "-"))
    L.append("")
    L.append("Floors of the formulations (hand counts of the VALU instructions the EXACT semantics need, per frame):\n")
    for k, (f, rows, waves, floor) in META.items():
        if floor != "-":
            L.append("* `%s`: %s" % (k, floor))
    text = "\n".join(L) + "\n"
    # the ceiling table (DESIGN.md section 0 and README.md): measured instructions, floor of the formulation, what sets it
    C = ["Generated by `tools/design_tables.py` from `profiles/%s_*`.\n" % tag,
         "| stage | VALU instructions / frame (PMC) | floor of the formulation | ms at the issue ceiling (measured count; 65 536 frames) | measured ms | what exact semantics set the floor |",
         "|---|---|---|---|---|---|"]
    tm = tf = 0.0
    for stage in bench.STAGE_KERNELS:
        if stage not in b["stages"]:
            continue
        sv = bench.stage_valu(valu, stage) or 0.0
        fl, why = CEILING[stage]
        tm += sv
        tf += fl if fl is not None else sv
        C.append("| %s | %d | %s | %.2f | %.2f | %s |" % (stage, sv, ("%d" % fl) if fl is not None else "(= measured)",
                                                        sv * 65536 / ISSUE_CEILING * 1e3, b["stages"][stage]["ms_per_step"], why))
    C.append("| **pipeline** | **%d** | **%d** | **%.2f** | **%.2f** | at the ceiling of %.2e instructions/s: **%.2f M frames/s** with the measured counts, **%.2f M** "
             "with every kernel at its floor = %.2f of the HBM roof (the contract's 0.50 = 9.4 M frames/s needs <= %d instructions per frame) |"
             % (tm, tf, tm * 65536 / ISSUE_CEILING * 1e3, tot_ms, ISSUE_CEILING, ISSUE_CEILING / tm / 1e6, ISSUE_CEILING / tf / 1e6,
                ISSUE_CEILING / tf * b["config"]["algorithmic_bytes_per_unit"] / 8e12, ISSUE_CEILING / 9.4e6))
    ctext = "\n".join(C) + "\n"

    def put(path, a, z, body):
        s = open(path).read()
        if a not in s:
            return False
        i, j = s.index(a) + len(a), s.index(z)
        open(path, "w").write(s[:i] + body + s[j:])
        return True

    if "--write" in sys.argv:
        put(os.path.join(ROOT, "DESIGN.md"), "<!-- GENERATED:kernel-tables (tools/design_tables.py) -->\n", "<!-- /GENERATED:kernel-tables -->", text)
        for f in ("DESIGN.md", "README.md"):
            put(os.path.join(ROOT, f), "<!-- GENERATED:ceiling-table (tools/design_tables.py) -->\n", "<!-- /GENERATED:ceiling-table -->", ctext)
        print("DESIGN.md / README.md updated from profiles/%s_*" % tag)
    else:
        print(text)
        print(ctext)


if __name__ == "__main__":
    main()
