#!/usr/bin/env python3
"""Writes valurate2.hip: per-SIMD issue cost of vector instructions with the REGISTERS chosen by hand (bank = number mod 4),
so that operand-bank effects and instruction kinds can be told apart.  Every variant is 8 instructions on 8 destination
registers, repeated 4x per loop trip, all in one asm block; 256 blocks (one per CU) of 4*w waves put w waves on each SIMD.
    python3 gen_valurate2.py > valurate2.hip && hipcc -O3 --offload-arch=gfx950 valurate2.hip -o valurate2"""
D0 = [16, 20, 24, 28, 32, 36, 48, 52]          # bank 0 destinations
D1 = [17, 21, 25, 29, 33, 37, 49, 53]          # bank 1
D2 = [18, 22, 26, 30, 34, 38, 50, 54]
D3 = [19, 23, 27, 31, 35, 39, 51, 55]
PAIR = [16, 20, 24, 28, 32, 36, 48, 52]        # even pairs v[n:n+1]
# constants: v40 (bank 0), v41 (1), v42 (2), v43 (3), v44 (0), v45 (1), v46 (2), v47 (3); pairs v[40:41], v[42:43], v[44:45]
V = []
def var(name, fmt, dsts=D0):
    V.append((name, [fmt.format(d=d, d1=d + 1, e=dsts[(i + 1) % 8]) for i, d in enumerate(dsts)]))
var("fma d(b0)=d*b1+b2 [3 banks]", "v_fma_f32 v{d}, v{d}, v41, v42")
var("fma d(b0)=d*b0+b1 [src0,src1 same bank]", "v_fma_f32 v{d}, v{d}, v40, v41")
var("fma d(b0)=d*b1+b0 [src0,src2 same bank]", "v_fma_f32 v{d}, v{d}, v41, v40")
var("fma d(b0)=d*b1+b1' [src1,src2 same bank]", "v_fma_f32 v{d}, v{d}, v41, v45")
var("fma d(b0)=d*b0+b0' [all one bank]", "v_fma_f32 v{d}, v{d}, v40, v44")
var("fma d(b0)=d*b1+b1 [src1==src2 register]", "v_fma_f32 v{d}, v{d}, v41, v41")
var("fma d(b0)=d*d+b1 [src0==src1 register]", "v_fma_f32 v{d}, v{d}, v{d}, v41")
var("fma d(b0)=d*s4+b1 [sgpr]", "v_fma_f32 v{d}, v{d}, s4, v41")
var("fma d(b0)=d*0.5+b1 [inline const]", "v_fma_f32 v{d}, v{d}, 0.5, v41")
var("fmac d(b0)+=literal*b1", "v_fmac_f32 v{d}, 0x3f8ccccd, v41")
var("fmaak d=d*b1+literal", "v_fmaak_f32 v{d}, v{d}, v41, 0x3f8ccccd")
var("fmamk d=d*literal+b1", "v_fmamk_f32 v{d}, v{d}, 0x3f8ccccd, v41")
var("mul_f32 d=literal*d", "v_mul_f32 v{d}, 0x3f8ccccd, v{d}")
var("add_f32 d=literal+d", "v_add_f32 v{d}, 0x3f8ccccd, v{d}")
var("add_u32 d=literal+d", "v_add_u32 v{d}, 0x12345, v{d}")
var("and_b32 d=literal&d", "v_and_b32 v{d}, 0xffff, v{d}")
var("mov_b32 d=literal", "v_mov_b32 v{d}, 0x3f8ccccd")
var("mov_b32 d=s4", "v_mov_b32 v{d}, s4")
var("add_f32 d=s4+d", "v_add_f32 v{d}, s4, v{d}")
var("mul_f32 d=s4*d", "v_mul_f32 v{d}, s4, v{d}")
var("add_u32 d=s4+d", "v_add_u32 v{d}, s4, v{d}")
var("max_f32 d=max(b1,b2) [dst not read]", "v_max_f32 v{d}, v41, v42")
var("min_f32", "v_min_f32 v{d}, v{d}, v41")
var("max_i32", "v_max_i32 v{d}, v{d}, v41")
var("min_u32", "v_min_u32 v{d}, v{d}, v41")
var("or_b32", "v_or_b32 v{d}, v{d}, v41")
var("xor_b32", "v_xor_b32 v{d}, v{d}, v41")
var("ashrrev_i32 d=d>>2", "v_ashrrev_i32 v{d}, 2, v{d}")
var("lshrrev_b32 d=d>>2", "v_lshrrev_b32 v{d}, 2, v{d}")
var("lshlrev_b32 d=d<<b1", "v_lshlrev_b32 v{d}, v41, v{d}")
var("lshl_add_u64", "v_lshl_add_u64 v[{d}:{d1}], v[{d}:{d1}], 2, v[42:43]", PAIR)
var("add_co_u32", "v_add_co_u32 v{d}, vcc, v{d}, v41")
var("addc_co_u32", "v_addc_co_u32 v{d}, vcc, v{d}, v41, vcc")
var("mul_f32 d=d*b1 mul:2 (omod, VOP3)", "v_mul_f32 v{d}, v{d}, v41 mul:2")
var("add_f32 d=-d+b1 (neg, VOP3)", "v_add_f32 v{d}, -v{d}, v41")
var("add_f32 clamp (VOP3)", "v_add_f32 v{d}, v{d}, v41 clamp")
var("fma_mix? cvt_pk_u8_f32", "v_cvt_pk_u8_f32 v{d}, v{d}, 0, v41")
var("cvt_f16_f32", "v_cvt_f16_f32 v{d}, v{d}")
var("cvt_f32_f16", "v_cvt_f32_f16 v{d}, v{d}")
var("sad_u8", "v_sad_u8 v{d}, v{d}, v41, v42")
var("alignbyte_b32", "v_alignbyte_b32 v{d}, v{d}, v41, 2")
var("alignbit_b32", "v_alignbit_b32 v{d}, v{d}, v41, 8")
var("bfi_b32", "v_bfi_b32 v{d}, v{d}, v41, v42")
var("xad_u32", "v_xad_u32 v{d}, v{d}, v41, v42")
var("mbcnt_lo", "v_mbcnt_lo_u32_b32 v{d}, -1, v{d}")
var("fma d(b0)=d*0.5+1.0 [one vgpr]", "v_fma_f32 v{d}, v{d}, 0.5, 1.0")
var("fma e(b0)=d(b0)*b1+b2 [dst != src]", "v_fma_f32 v{e}, v{d}, v41, v42")
var("fma d over banks 0..3 mixed, consts b1,b2", "v_fma_f32 v{d}, v{d}, v41, v42", [16, 17, 18, 19, 20, 21, 22, 23])
var("fmac d(b0)+=b1*b2 [VOP2, 3 reads]", "v_fmac_f32 v{d}, v41, v42")
var("fmac d(b0)+=b0*b1", "v_fmac_f32 v{d}, v40, v41")
var("fmac d(b0)+=s4*b1", "v_fmac_f32 v{d}, s4, v41")
var("add_f32 d(b0)=d+b1", "v_add_f32 v{d}, v{d}, v41")
var("add_f32 d(b0)=d+b0", "v_add_f32 v{d}, v{d}, v40")
var("add_f32 d(b0)=b1+b2 [dst not read]", "v_add_f32 v{d}, v41, v42")
var("add_f32 VOP3 d=|d|+b1", "v_add_f32 v{d}, |v{d}|, v41")
var("mul_f32 d=d*b1", "v_mul_f32 v{d}, v{d}, v41")
var("sub_f32 d=d-b1", "v_sub_f32 v{d}, v{d}, v41")
var("max_f32 d=max(d,b1)", "v_max_f32 v{d}, v{d}, v41")
var("min3_f32", "v_min3_f32 v{d}, v{d}, v41, v42")
var("med3_f32", "v_med3_f32 v{d}, v{d}, v41, v42")
var("mov_b32 d=b1", "v_mov_b32 v{d}, v41")
var("mov_b32 d=const", "v_mov_b32 v{d}, 1.0")
var("and_b32", "v_and_b32 v{d}, v{d}, v41")
var("and_b32 const", "v_and_b32 v{d}, 15, v{d}")
var("lshlrev_b32 d=d<<2", "v_lshlrev_b32 v{d}, 2, v{d}")
var("lshrrev_b32 d=d>>b1", "v_lshrrev_b32 v{d}, v41, v{d}")
var("add_u32 d=d+b1", "v_add_u32 v{d}, v{d}, v41")
var("add_u32 d=d+const", "v_add_u32 v{d}, 4, v{d}")
var("sub_u32", "v_sub_u32 v{d}, v{d}, v41")
var("add3_u32", "v_add3_u32 v{d}, v{d}, v41, v42")
var("lshl_add_u32 d=(d<<2)+b1", "v_lshl_add_u32 v{d}, v{d}, 2, v41")
var("add_lshl_u32", "v_add_lshl_u32 v{d}, v{d}, v41, 2")
var("lshl_or_b32", "v_lshl_or_b32 v{d}, v{d}, 2, v41")
var("and_or_b32", "v_and_or_b32 v{d}, v{d}, v41, v42")
var("bfe_u32", "v_bfe_u32 v{d}, v{d}, 4, 8")
var("perm_b32", "v_perm_b32 v{d}, v{d}, v41, v42")
var("mul_u32_u24", "v_mul_u32_u24 v{d}, v{d}, v41")
var("mul_i32_i24", "v_mul_i32_i24 v{d}, v{d}, v41")
var("mad_u32_u24", "v_mad_u32_u24 v{d}, v{d}, v41, v42")
var("mad_i32_i24", "v_mad_i32_i24 v{d}, v{d}, v41, v42")
var("mul_lo_u32", "v_mul_lo_u32 v{d}, v{d}, v41")
var("mul_hi_u32", "v_mul_hi_u32 v{d}, v{d}, v41")
var("cvt_f32_i32", "v_cvt_f32_i32 v{d}, v{d}")
var("cvt_f32_u32", "v_cvt_f32_u32 v{d}, v{d}")
var("cvt_i32_f32", "v_cvt_i32_f32 v{d}, v{d}")
var("cvt_f32_ubyte0", "v_cvt_f32_ubyte0 v{d}, v{d}")
var("cvt_flr_i32_f32", "v_cvt_flr_i32_f32 v{d}, v{d}")
var("floor_f32", "v_floor_f32 v{d}, v{d}")
var("fract_f32", "v_fract_f32 v{d}, v{d}")
var("rndne_f32", "v_rndne_f32 v{d}, v{d}")
var("ldexp_f32", "v_ldexp_f32 v{d}, v{d}, v41")
var("cndmask_b32", "v_cndmask_b32 v{d}, v{d}, v41, vcc")
var("cmp_lt_f32 vcc", "v_cmp_lt_f32 vcc, v{d}, v41")
var("cmp_lt_f32 sgpr pair (VOP3)", "v_cmp_lt_f32 s[8:9], v{d}, v41")
var("cmp_lt_i32 vcc", "v_cmp_lt_i32 vcc, v{d}, v41")
var("rcp_f32", "v_rcp_f32 v{d}, v{d}")
var("rsq_f32", "v_rsq_f32 v{d}, v{d}")
var("sqrt_f32", "v_sqrt_f32 v{d}, v{d}")
var("exp_f32", "v_exp_f32 v{d}, v{d}")
var("add_f32 dpp row_shr:1", "v_add_f32_dpp v{d}, v{d}, v{d} row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")
var("add_f32 dpp quad_perm", "v_add_f32_dpp v{d}, v{d}, v{d} quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
var("mov_b32 dpp row_shr:1", "v_mov_b32_dpp v{d}, v41 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")
var("mov_b32 dpp row_bcast15", "v_mov_b32_dpp v{d}, v41 row_bcast:15 row_mask:0xa bank_mask:0xf")
var("add_f32 sdwa", "v_add_f32_sdwa v{d}, v{d}, v41 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0")
var("pk_fma_f32 [3 banks]", "v_pk_fma_f32 v[{d}:{d1}], v[{d}:{d1}], v[42:43], v[46:47]", PAIR)
var("pk_fma_f32 consts same pair bank", "v_pk_fma_f32 v[{d}:{d1}], v[{d}:{d1}], v[40:41], v[44:45]", PAIR)
var("pk_mul_f32", "v_pk_mul_f32 v[{d}:{d1}], v[{d}:{d1}], v[42:43]", PAIR)
var("pk_add_f32", "v_pk_add_f32 v[{d}:{d1}], v[{d}:{d1}], v[42:43]", PAIR)
var("pk_add_f16", "v_pk_add_f16 v{d}, v{d}, v41")
var("pk_fma_f16", "v_pk_fma_f16 v{d}, v{d}, v41, v42")
var("fma_f64", "v_fma_f64 v[{d}:{d1}], v[{d}:{d1}], v[42:43], v[46:47]", PAIR)
var("add_f64", "v_add_f64 v[{d}:{d1}], v[{d}:{d1}], v[42:43]", PAIR)
var("mul_f64", "v_mul_f64 v[{d}:{d1}], v[{d}:{d1}], v[42:43]", PAIR)
var("cvt_f64_f32", "v_cvt_f64_f32 v[{d}:{d1}], v{d}", PAIR)
var("cvt_f32_f64", "v_cvt_f32_f64 v{d}, v[{d}:{d1}]", PAIR)
var("mov_b64", "v_mov_b64 v[{d}:{d1}], v[42:43]", PAIR)
var("lshlrev_b64", "v_lshlrev_b64 v[{d}:{d1}], 2, v[{d}:{d1}]", PAIR)
var("mad_u64_u32", "v_mad_u64_u32 v[{d}:{d1}], s[8:9], v{d}, v42, v[46:47]", PAIR)
var("readfirstlane (to s10)", "v_readfirstlane_b32 s10, v{d}")
var("permlane32_swap", "v_permlane32_swap_b32 v{d}, v{e}")
var("s_nop 0 [scalar slot]", "s_nop 0")
var("s_add_u32 [scalar]", "s_add_u32 s10, s10, 1")
V.append(("mix: fma(3 banks) / add_u32 alternating", ["v_fma_f32 v16, v16, v41, v42", "v_add_u32 v20, v20, v41", "v_fma_f32 v24, v24, v41, v42", "v_add_u32 v28, v28, v41",
                    "v_fma_f32 v32, v32, v41, v42", "v_add_u32 v36, v36, v41", "v_fma_f32 v48, v48, v41, v42", "v_add_u32 v52, v52, v41"]))
V.append(("mix: fma(3 banks) / mul_u32_u24 alternating",
          ["v_fma_f32 v16, v16, v41, v42", "v_mul_u32_u24 v20, v20, v41", "v_fma_f32 v24, v24, v41, v42", "v_mul_u32_u24 v28, v28, v41",
           "v_fma_f32 v32, v32, v41, v42", "v_mul_u32_u24 v36, v36, v41", "v_fma_f32 v48, v48, v41, v42", "v_mul_u32_u24 v52, v52, v41"]))
V.append(("mix: fma(3 banks) / s_add_u32 alternating (counts the 4 fma + 4 scalar)",
          ["v_fma_f32 v16, v16, v41, v42", "s_add_u32 s10, s10, 1", "v_fma_f32 v24, v24, v41, v42", "s_add_u32 s11, s11, 1",
           "v_fma_f32 v32, v32, v41, v42", "s_add_u32 s10, s10, 1", "v_fma_f32 v48, v48, v41, v42", "s_add_u32 s11, s11, 1"]))

def pat(name, seq):
    F = ["v_fma_f32 v16, v16, v41, v42", "v_fma_f32 v20, v20, v41, v42", "v_fma_f32 v24, v24, v41, v42", "v_fma_f32 v28, v28, v41, v42",
         "v_fma_f32 v17, v17, v41, v42", "v_fma_f32 v21, v21, v41, v42", "v_fma_f32 v25, v25, v41, v42", "v_fma_f32 v29, v29, v41, v42"]
    H = ["v_mul_u32_u24 v32, v32, v41", "v_mul_u32_u24 v36, v36, v41", "v_mul_u32_u24 v48, v48, v41", "v_mul_u32_u24 v52, v52, v41",
         "v_mul_u32_u24 v33, v33, v41", "v_mul_u32_u24 v37, v37, v41", "v_mul_u32_u24 v49, v49, v41", "v_mul_u32_u24 v53, v53, v41"]
    S = ["v_fma_f32 v32, v32, s4, v42", "v_fma_f32 v36, v36, s4, v42", "v_fma_f32 v48, v48, s4, v42", "v_fma_f32 v52, v52, s4, v42",
         "v_fma_f32 v33, v33, s4, v42", "v_fma_f32 v37, v37, s4, v42", "v_fma_f32 v49, v49, s4, v42", "v_fma_f32 v53, v53, s4, v42"]
    Q = ["v_rcp_f32 v32, v32", "v_rcp_f32 v36, v36", "v_rcp_f32 v48, v48", "v_rcp_f32 v52, v52", "v_rcp_f32 v33, v33", "v_rcp_f32 v37, v37", "v_rcp_f32 v49, v49", "v_rcp_f32 v53, v53"]
    L = ["ds_read_b32 v32, v43", "ds_read_b32 v36, v43", "ds_read_b32 v48, v43", "ds_read_b32 v52, v43", "ds_read_b32 v33, v43", "ds_read_b32 v37, v43", "ds_read_b32 v49, v43", "ds_read_b32 v53, v43"]
    n = {"F": 0, "H": 0, "S": 0, "Q": 0, "L": 0}
    tab = {"F": F, "H": H, "S": S, "Q": Q, "L": L}
    out = []
    for ch in seq:
        out.append(tab[ch][n[ch] % 8]); n[ch] += 1
    if "L" in seq: out.append("s_waitcnt lgkmcnt(0)")
    V.append((name, out))
pat("pattern FFFFFFFF (fma)", "FFFFFFFF")
pat("pattern HHHHHHHH (mul_u24)", "HHHHHHHH")
pat("pattern FHFHFHFH", "FHFHFHFH")
pat("pattern FFHHFFHH", "FFHHFFHH")
pat("pattern FFFFHHHH", "FFFFHHHH")
pat("pattern FFFHFFFH", "FFFHFFFH")
pat("pattern FFFFFFFH", "FFFFFFFH")
pat("pattern FSFSFSFS (S = fma with sgpr)", "FSFSFSFS")
pat("pattern FFFSFFFS", "FFFSFFFS")
pat("pattern FFFFFFFQ (Q = rcp)", "FFFFFFFQ")
pat("pattern FFFQFFFQ", "FFFQFFFQ")
pat("pattern FFFLFFFL (L = ds_read_b32 + wait at the end; counts 8)", "FFFLFFFL")
var("cndmask d=vcc?b1:d", "v_cndmask_b32 v{d}, v{d}, v41, vcc")
var("cndmask e=vcc?b1:b2 [dst not read]", "v_cndmask_b32 v{d}, v42, v41, vcc")
var("cndmask d=s[8:9]?b1:d (VOP3)", "v_cndmask_b32 v{d}, v{d}, v41, s[8:9]")
var("cndmask d=vcc?0:d", "v_cndmask_b32 v{d}, 0, v{d}, vcc")

V.append(("select pair: v_cmp vcc + v_cndmask vcc (counts 8 = 4 pairs)", ["v_cmp_lt_f32 vcc, v16, v41", "v_cndmask_b32 v20, v20, v42, vcc", "v_cmp_lt_f32 vcc, v24, v41", "v_cndmask_b32 v28, v28, v42, vcc",
           "v_cmp_lt_f32 vcc, v32, v41", "v_cndmask_b32 v36, v36, v42, vcc", "v_cmp_lt_f32 vcc, v48, v41", "v_cndmask_b32 v52, v52, v42, vcc"]))
V.append(("select pair: v_cmp s[8:9] + v_cndmask s[8:9] (counts 8)", ["v_cmp_lt_f32 s[8:9], v16, v41", "v_cndmask_b32 v20, v20, v42, s[8:9]", "v_cmp_lt_f32 s[8:9], v24, v41", "v_cndmask_b32 v28, v28, v42, s[8:9]",
           "v_cmp_lt_f32 s[8:9], v32, v41", "v_cndmask_b32 v36, v36, v42, s[8:9]", "v_cmp_lt_f32 s[8:9], v48, v41", "v_cndmask_b32 v52, v52, v42, s[8:9]"]))
V.append(("select: one v_cmp vcc then 7 v_cndmask vcc", ["v_cmp_lt_f32 vcc, v16, v41", "v_cndmask_b32 v20, v20, v42, vcc", "v_cndmask_b32 v24, v24, v42, vcc", "v_cndmask_b32 v28, v28, v42, vcc",
           "v_cndmask_b32 v32, v32, v42, vcc", "v_cndmask_b32 v36, v36, v42, vcc", "v_cndmask_b32 v48, v48, v42, vcc", "v_cndmask_b32 v52, v52, v42, vcc"]))
V.append(("select: v_cndmask vcc between fmas (FCFC)", ["v_fma_f32 v16, v16, v41, v42", "v_cndmask_b32 v20, v20, v42, vcc", "v_fma_f32 v24, v24, v41, v42", "v_cndmask_b32 v28, v28, v42, vcc",
           "v_fma_f32 v32, v32, v41, v42", "v_cndmask_b32 v36, v36, v42, vcc", "v_fma_f32 v48, v48, v41, v42", "v_cndmask_b32 v52, v52, v42, vcc"]))

# ---- cross-wave mix: the waves of a SIMD run two different loops (threads 0-255 / 512-767 of a 1024-thread block, i.e. every
# other wave of each SIMD, loop A; the rest loop B)
XW = []
F8 = ["v_fma_f32 v16, v16, v41, v42", "v_fma_f32 v20, v20, v41, v42", "v_fma_f32 v24, v24, v41, v42", "v_fma_f32 v28, v28, v41, v42",
      "v_fma_f32 v32, v32, v41, v42", "v_fma_f32 v36, v36, v41, v42", "v_fma_f32 v48, v48, v41, v42", "v_fma_f32 v52, v52, v41, v42"]
H8 = ["v_mul_u32_u24 v16, v16, v41", "v_mul_u32_u24 v20, v20, v41", "v_mul_u32_u24 v24, v24, v41", "v_mul_u32_u24 v28, v28, v41",
      "v_mul_u32_u24 v32, v32, v41", "v_mul_u32_u24 v36, v36, v41", "v_mul_u32_u24 v48, v48, v41", "v_mul_u32_u24 v52, v52, v41"]
XW.append(("cross-wave: half of a SIMD's waves all-F (fma), the other half all-H (mul_u24); per instruction of either kind", F8, H8))
XW.append(("cross-wave control: both halves all-F", F8, F8))
XW.append(("cross-wave control: both halves all-H", H8, H8))
# ---- code size: the same all-F stream as one loop body of 32 .. 8192 instructions (is the one-pass rate an artefact of a loop
# that lives in the instruction buffer?), as VOP3 v_fma_f32 and as VOP2 v_fmac_f32
LONG = []
FM8 = ["v_fmac_f32 v%d, v43, v41" % d for d in (16, 20, 24, 28, 32, 36, 48, 52)]
for reps in (4, 32, 128, 512, 1024):
    LONG.append(("straight-line all-F stream, loop body %d instructions (%d KB of VOP3)" % (8 * reps, 8 * reps * 8 // 1024), F8, reps))
    LONG.append(("straight-line all-F stream as VOP2 (v_fmac_f32), loop body %d instructions (%d KB)" % (8 * reps, 8 * reps * 4 // 1024), FM8, reps))

clob = ", ".join('"v%d"' % i for i in range(16, 56)) + ', "s4", "s8", "s9", "s10", "s11", "vcc", "scc"'
print("// GENERATED by gen_valurate2.py -- do not edit.  See that script for what this measures.")
print("#include <hip/hip_runtime.h>\n#include <cstdio>\n#include <cstdlib>\n#include <cstring>")
print('#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)')
init = "\\n\\t".join(["v_mov_b32 v%d, 1.0" % i for i in range(16, 56)] + ["s_mov_b32 s4, 0x3f000000", "s_mov_b32 s10, 0", "s_mov_b32 s11, 0", "s_mov_b64 vcc, 0x5555", "s_mov_b64 s[8:9], 0x3333", "v_mov_b32 v43, 0"])
for k, (name, ins) in enumerate(V):
    body = "\\n\\t".join(ins * 4)
    print("__global__ __launch_bounds__(1024) void k%d(int iters, long long* cyc) {" % k)
    print('    asm volatile("%s" ::: %s);' % (init, clob))
    print("    long long t0 = __builtin_readcyclecounter();")
    print("    for (int it = 0; it < iters; it++) {")
    print('        asm volatile("%s" ::: %s);' % (body, clob))
    print("    }")
    print("    long long t1 = __builtin_readcyclecounter();")
    print("    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;\n}")
for k, (name, insA, insB) in enumerate(XW):
    print("__global__ __launch_bounds__(1024) void kx%d(int iters, long long* cyc) {" % k)
    print('    asm volatile("%s" ::: %s);' % (init, clob))
    print("    if (((threadIdx.x >> 8) & 1) == 0) {")
    print("        for (int it = 0; it < iters; it++) {")
    print('            asm volatile("%s" ::: %s);' % ("\\n\\t".join(insA * 4), clob))
    print("        }")
    print("    } else {")
    print("        for (int it = 0; it < iters; it++) {")
    print('            asm volatile("%s" ::: %s);' % ("\\n\\t".join(insB * 4), clob))
    print("        }")
    print("    }")
    print("    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = 0;\n}")
for k, (name, ins, reps) in enumerate(LONG):
    print("__global__ __launch_bounds__(1024) void kl%d(int iters, long long* cyc) {" % k)
    print('    asm volatile("%s" ::: %s);' % (init, clob))
    print("    for (int it = 0; it < iters * 4 / %d; it++) {" % reps)
    print('        asm volatile("%s" ::: %s);' % ("\\n\\t".join(ins * reps), clob))
    print("    }")
    print("    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = 0;\n}")
print("typedef void (*kern_t)(int, long long*);")
print("static const struct { const char* name; kern_t k; } VAR[] = {")
for k, (name, ins) in enumerate(V):
    print('    {"%s", k%d},' % (name, k))
for k, (name, insA, insB) in enumerate(XW):
    print('    {"%s", kx%d},' % (name, k))
for k, (name, ins, reps) in enumerate(LONG):
    print('    {"%s", kl%d},' % (name, k))
print("};")
print(r'''
int main(int argc, char** argv) {
    const char* only = argc > 1 ? argv[1] : nullptr;
    long long* d_cyc; CK(hipMalloc(&d_cyc, 64));
    int khz = 0; CK(hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0));
    const double ghz = khz * 1e-6;
    const int iters = 20000;
    printf("| instruction (8 independent destinations x 4 per trip) | 1 wave/SIMD | 2 | 4 | 8 | one wave's own ticks/instr at 4 |\n|---|---|---|---|---|---|\n");
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (auto& v : VAR) {
        if (only && !strstr(v.name, only)) continue;
        printf("| %s |", v.name);
        double own4 = 0;
        for (int wps : {1, 2, 4, 8}) {
            const int threads = 64 * 4 * (wps > 4 ? 4 : wps), blocks = 256 * (wps > 4 ? wps / 4 : 1);
            hipLaunchKernelGGL(v.k, dim3(blocks), dim3(threads), 0, 0, iters / 8, d_cyc);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(v.k, dim3(blocks), dim3(threads), 0, 0, iters, d_cyc);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            long long wc; CK(hipMemcpy(&wc, d_cyc, 8, hipMemcpyDeviceToHost));
            printf(" %.2f |", ms * 1e-3 * ghz * 1e9 / ((double)iters * 32 * wps));
            if (wps == 4) own4 = (double)wc / ((double)iters * 32);
        }
        printf(" %.2f |\n", own4);
        fflush(stdout);
    }
    return 0;
}''')
