#!/usr/bin/env python3
"""Static count of a kernel's vector instructions by the issue-rate class measured with scripts/diag/valurate2 on MI355X
(profiles/r03_valu_rates.md): FULL = one pass of 2 cycles per wave64 (fp32 fma/fmac/add/sub/mul, mov_b32, and/or/xor,
lshrrev, add/sub_u32 -- with VGPR or inline-constant operands), HALF = 4 cycles (an SGPR or literal operand on any of the
above, max/min/med3, lshlrev, three-operand integer ops, 24-bit and 32-bit multiplies, every cvt/floor/fract, compares,
DPP/SDWA, packed and fp64 ops, readlane), QUARTER = 8 cycles (rcp/rsq/sqrt/exp/log/sin/cos, permlane swap).
    isa_rate.py file.s kernel-name-substring   -> static counts over the whole kernel text (loops counted once)"""
import re, sys, collections
FULL = {"v_fma_f32", "v_fmac_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_mov_b32", "v_and_b32", "v_or_b32", "v_xor_b32",
        "v_lshrrev_b32", "v_ashrrev_i32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_not_b32"}
QUARTER = {"v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32", "v_permlane32_swap_b32", "v_rcp_f64", "v_rsq_f64", "v_sqrt_f64", "v_rcp_iflag_f32"}
def classify(line):
    m = re.match(r"\s*(v_[a-z0-9_]+)\s*(.*)", line)
    if not m: return None
    op, rest = m.group(1), m.group(2).split(";")[0]
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
    if base.startswith("v_mfma") or base.startswith("v_accvgpr"): return ("mfma", base)
    if base in QUARTER: return ("quarter", base)
    if op.endswith("_dpp") or " row_" in rest or "quad_perm" in rest or op.endswith("_sdwa"): return ("half", base + " dpp")
    if base in FULL:
        ops = [o.strip() for o in rest.split(",")]
        srcs = ops[1:]
        for s in srcs:
            s = s.strip("|-").replace("neg(", "").replace(")", "")
            if re.match(r"^(s\d+|s\[|vcc|exec|ttmp|m0|src_)", s): return ("half", base + " sgpr")
            if re.match(r"^0x[0-9a-f]+$", s): return ("unknown-literal", base + " literal")
        return ("full", base)
    return ("half", base)
def main():
    path, key = sys.argv[1], sys.argv[2]
    txt = open(path).read().split("\n")
    start = next(i for i, l in enumerate(txt) if l.startswith("_Z") and key in l.split(":")[0])
    end = next(i for i in range(start, len(txt)) if "s_endpgm" in txt[i])
    cls = collections.Counter(); kinds = collections.Counter()
    for l in txt[start:end]:
        c = classify(l)
        if c: cls[c[0]] += 1; kinds[c] += 1
    tot = sum(cls.values())
    print("kernel", txt[start].split(":")[0], "static VALU", tot, dict(cls))
    w = cls["full"] * 2 + (cls["half"] + cls["unknown-literal"]) * 4 + cls["quarter"] * 8
    print("weighted cycles %d = %.2f per instruction" % (w, w / max(tot, 1)))
    for (c, k), n in kinds.most_common(45): print("  %-8s %-28s %d" % (c, k, n))
main()
