#!/usr/bin/env python3
"""Developer tool: static opcode histogram of one kernel in a device assembly file (tools/asm.sh <file>), weighted with the
issue cost classes measured by tools/ubench/valu_table.hip (profiles/r4_valu_table_gfx950.txt): which instructions the
kernel's issue time is made of.  usage: tools/dev/isa_hist.py /tmp/detect.s <kernel-name-substring> [top]"""
import collections
import re
import sys

FAST = {"v_mov_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_ashrrev_i32",
        "v_lshrrev_b32", "v_add_f32", "v_sub_f32", "v_mul_f32", "v_mul_legacy_f32", "v_fma_f32", "v_fmac_f32", "v_add_co_u32",
        "v_accvgpr_read_b32", "v_accvgpr_write_b32",
        # the two-operand 16-bit forms (the three-operand ones, v_mad_u16 / v_max3_u16 / v_med3_u16, are quarter rate)
        "v_add_u16", "v_sub_u16", "v_max_u16", "v_min_u16", "v_max_i16", "v_min_i16", "v_mul_lo_u16", "v_lshlrev_b16",
        "v_lshrrev_b16", "v_ashrrev_i16", "v_add_f16", "v_sub_f16", "v_mul_f16", "v_max_f16", "v_min_f16"}
COST = {"fast": 2.5, "slow": 4.3, "f64": 5.3, "trans": 8.2}


def klass(op):
    base = op.replace("_e32", "").replace("_e64", "")
    if base.endswith("_dpp") or base.endswith("_sdwa"):
        return "slow"
    if base in FAST:
        return "fast"
    if "_f64" in base or base.startswith("v_pk_") and base.endswith("_f32"):
        return "f64"
    if base in ("v_exp_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_log_f32", "v_rcp_f64", "v_fma_f16", "v_mad_u16",
                "v_max3_u16", "v_min3_u16", "v_med3_u16", "v_max3_i16", "v_min3_i16", "v_med3_i16", "v_mad_i16"):
        return "trans"
    return "slow"


def main():
    path, name = sys.argv[1], sys.argv[2]
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
    inside = False
    hist = collections.Counter()
    other = collections.Counter()
    for line in open(path):
        if re.match(r"^_Z\w*%s\w*:" % re.escape(name), line):
            inside = True
            hist.clear(), other.clear()
            continue
        if inside and "s_endpgm" in line:
            break
        if inside:
            m = re.match(r"^\s+([a-z_0-9]+)", line)
            if m:
                op = m.group(1)
                (hist if op.startswith("v_") and not op.startswith("v_mfma") else other)[op] += 1
    tot = sum(COST[klass(o)] * n for o, n in hist.items())
    print("VALU instructions %d, weighted cycles %.0f (%.2f per instruction); other: %s" % (
        sum(hist.values()), tot, tot / max(1, sum(hist.values())),
        ", ".join("%s %d" % (k, v) for k, v in other.most_common(8))))
    by = collections.Counter()
    for o, n in hist.items():
        by[klass(o)] += n
    print("classes:", dict(by))
    for o, n in sorted(hist.items(), key=lambda t: -COST[klass(t[0])] * t[1])[:top]:
        print("  %-28s %5d  %-5s %6.0f cycles  %4.1f %%" % (o, n, klass(o), COST[klass(o)] * n, 100 * COST[klass(o)] * n / tot))


main()
