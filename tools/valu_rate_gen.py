# generates tools/valu_rate.hip test list
import sys
tests = []
def body8(fmt, regs="f"):
    # fmt uses {d} dest/src0 reg index, {n} next reg index, {c} const reg, {cd} const double
    s = ""
    for i in range(8):
        if regs == "f":
            d, n = "%%%d" % i, "%%%d" % ((i + 1) % 8)
        else:
            d, n = "%%%d" % (8 + i), "%%%d" % (8 + (i + 1) % 8)
        s += fmt.format(d=d, n=n, c="%16", cd="%17", f="%%%d" % i, dd="%%%d" % (8 + i), s="s[%d:%d]" % (20 + 2 * (i % 4), 21 + 2 * (i % 4))) + "\\n "
    return s
def T(name, fmt, regs="f", valu=8):
    tests.append((name, body8(fmt, regs), valu))
def Tdep(name, fmt, regs="f"):
    s = ""
    for i in range(8):
        d = "%0" if regs == "f" else "%8"
        s += fmt.format(d=d, n=d, c="%16", cd="%17", f="%0", dd="%8", s="s[20:21]") + "\\n "
    tests.append((name, s, 8))

T("v_add_f32", "v_add_f32 {d}, {d}, {c}")
Tdep("v_add_f32_dep", "v_add_f32 {d}, {d}, {c}")
T("v_sub_f32", "v_sub_f32 {d}, {d}, {c}")
T("v_mul_f32", "v_mul_f32 {d}, {d}, {c}")
T("v_fma_f32_2same", "v_fma_f32 {d}, {d}, {c}, {c}")
T("v_fma_f32_3distinct", "v_fma_f32 {d}, {d}, {c}, {n}")
T("v_fmac_f32", "v_fmac_f32 {d}, {n}, {c}")
T("v_mac_like_mul_add", "v_mul_f32 {d}, {d}, {c}")
T("v_max_f32", "v_max_f32 {d}, {d}, {c}")
T("v_min_f32", "v_min_f32 {d}, {d}, {c}")
T("v_min3_f32", "v_min3_f32 {d}, {d}, {c}, {n}")
T("v_med3_f32", "v_med3_f32 {d}, {d}, {c}, {n}")
T("v_mov_b32", "v_mov_b32 {d}, {n}")
T("v_and_b32", "v_and_b32 {d}, {d}, {c}")
T("v_or_b32", "v_or_b32 {d}, {d}, {c}")
T("v_xor_b32", "v_xor_b32 {d}, {d}, {c}")
T("v_bfi_b32", "v_bfi_b32 {d}, {d}, {c}, {n}")
T("v_and_or_b32", "v_and_or_b32 {d}, {d}, {c}, {n}")
T("v_lshlrev_b32", "v_lshlrev_b32 {d}, 3, {d}")
T("v_lshrrev_b32", "v_lshrrev_b32 {d}, 3, {d}")
T("v_ashrrev_i32", "v_ashrrev_i32 {d}, 31, {d}")
T("v_bfe_i32", "v_bfe_i32 {d}, {d}, 0, 16")
T("v_add_u32", "v_add_u32 {d}, {d}, {c}")
T("v_sub_u32", "v_sub_u32 {d}, {d}, {c}")
T("v_add3_u32", "v_add3_u32 {d}, {d}, {c}, {n}")
T("v_lshl_add_u32", "v_lshl_add_u32 {d}, {d}, 2, {c}")
T("v_add_co_u32", "v_add_co_u32 {d}, vcc, {d}, {c}")
T("v_addc_co_u32", "v_addc_co_u32 {d}, vcc, {d}, {c}, vcc")
T("v_mul_lo_u32", "v_mul_lo_u32 {d}, {d}, {c}")
T("v_mul_u32_u24", "v_mul_u32_u24 {d}, {d}, {c}")
T("v_mad_u32_u24", "v_mad_u32_u24 {d}, {d}, {c}, {n}")
T("v_perm_b32", "v_perm_b32 {d}, {d}, {c}, {n}")
T("v_min_i32", "v_min_i32 {d}, {d}, {c}")
T("v_max_u32", "v_max_u32 {d}, {d}, {c}")
T("v_pk_add_f32", "v_pk_add_f32 {d}, {d}, {cd}", "d")
T("v_pk_mul_f32", "v_pk_mul_f32 {d}, {d}, {cd}", "d")
T("v_pk_fma_f32", "v_pk_fma_f32 {d}, {d}, {cd}, {cd}", "d")
T("v_pk_mov_b32", "v_pk_mov_b32 {d}, {n}, {cd}", "d")
T("v_pk_add_u16", "v_pk_add_u16 {d}, {d}, {c}")
T("v_pk_max_i16", "v_pk_max_i16 {d}, {d}, {c}")
T("v_add_f64", "v_add_f64 {d}, {d}, {cd}", "d")
Tdep("v_add_f64_dep", "v_add_f64 {d}, {d}, {cd}", "d")
T("v_mul_f64", "v_mul_f64 {d}, {d}, {cd}", "d")
T("v_fma_f64", "v_fma_f64 {d}, {d}, {cd}, {cd}", "d")
T("v_max_f64", "v_max_f64 {d}, {d}, {cd}", "d")
T("v_lshlrev_b64", "v_lshlrev_b64 {d}, 1, {d}", "d")
T("v_cvt_f64_f32", "v_cvt_f64_f32 {dd}, {f}")
T("v_cvt_f32_f64", "v_cvt_f32_f64 {f}, {dd}")
T("v_cvt_f32_i32", "v_cvt_f32_i32 {d}, {n}")
T("v_cvt_f32_u32", "v_cvt_f32_u32 {d}, {n}")
T("v_cvt_i32_f32", "v_cvt_i32_f32 {d}, {n}")
T("v_cvt_f32_ubyte0", "v_cvt_f32_ubyte0 {d}, {n}")
T("v_cvt_f32_i32_sdwa_w0", "v_cvt_f32_i32_sdwa {d}, sext({n}) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0")
T("v_cvt_f32_f16", "v_cvt_f32_f16 {d}, {n}")
T("v_cvt_f64_i32", "v_cvt_f64_i32 {dd}, {f}")
T("v_rndne_f32", "v_rndne_f32 {d}, {n}")
T("v_ldexp_f32", "v_ldexp_f32 {d}, {d}, 3")
T("v_frexp_mant_f32", "v_frexp_mant_f32 {d}, {n}")
T("v_rsq_f32", "v_rsq_f32 {d}, {d}")
T("v_rcp_f32", "v_rcp_f32 {d}, {d}")
T("v_sqrt_f32", "v_sqrt_f32 {d}, {d}")
T("v_rsq_f64", "v_rsq_f64 {d}, {d}", "d")
T("v_rcp_f64", "v_rcp_f64 {d}, {d}", "d")
T("v_sqrt_f64", "v_sqrt_f64 {d}, {d}", "d")
T("v_cmp_gt_f32_sgpr", "v_cmp_gt_f32 {s}, {d}, {c}")
T("v_cmp_gt_f32_vcc", "v_cmp_gt_f32 vcc, {d}, {c}")
T("v_cmp_lt_i32_vcc", "v_cmp_lt_i32 vcc, {d}, {c}")
T("v_cmp_gt_f64_vcc", "v_cmp_gt_f64 vcc, {dd}, {cd}")
T("v_cmp_class_f32", "v_cmp_class_f32 vcc, {d}, {c}")
T("v_cndmask_e64_sgpr", "v_cndmask_b32 {d}, {d}, {c}, {s}")
T("v_cndmask_e32_vcc", "v_cndmask_b32 {d}, {d}, {c}, vcc")
T("v_cndmask_e64_neg", "v_cndmask_b32 {d}, -{d}, {c}, {s}")
# chains / mixes
tests.append(("cmp_vcc_then_cndmask_vcc_dep", "".join("v_cmp_gt_f32 vcc, %0, %16\\n v_cndmask_b32 %0, %0, %1, vcc\\n " for _ in range(4)), 8))
tests.append(("sub_ashr_bfi_select_dep", "".join("v_sub_f32 %2, %0, %16\\n v_ashrrev_i32 %2, 31, %2\\n v_bfi_b32 %0, %2, %1, %0\\n " for _ in range(2)) + "v_sub_f32 %2, %0, %16\\n v_ashrrev_i32 %2, 31, %2\\n ", 8))
tests.append(("valu_add_salu_1to1", "".join("v_add_f32 %%%d, %%%d, %%16\\n s_and_b64 s[20:21], s[22:23], s[24:25]\\n " % (i, i) for i in range(4)), 4))
tests.append(("valu_add_salu_3to1", "v_add_f32 %0, %0, %16\\n v_add_f32 %1, %1, %16\\n v_add_f32 %2, %2, %16\\n s_and_b64 s[20:21], s[22:23], s[24:25]\\n v_add_f32 %3, %3, %16\\n v_add_f32 %4, %4, %16\\n v_add_f32 %5, %5, %16\\n s_or_b64 s[22:23], s[20:21], s[24:25]\\n ", 6))
tests.append(("valu_f64_salu_1to1", "".join("v_add_f64 %%%d, %%%d, %%17\\n s_and_b64 s[20:21], s[22:23], s[24:25]\\n " % (8 + i, 8 + i) for i in range(4)), 4))
tests.append(("valu_f64_salu_3to1", "v_add_f64 %8, %8, %17\\n v_add_f64 %9, %9, %17\\n v_add_f64 %10, %10, %17\\n s_and_b64 s[20:21], s[22:23], s[24:25]\\n v_add_f64 %11, %11, %17\\n v_add_f64 %12, %12, %17\\n v_add_f64 %13, %13, %17\\n s_or_b64 s[22:23], s[20:21], s[24:25]\\n ", 6))
tests.append(("salu_only", "".join("s_and_b64 s[20:21], s[22:23], s[24:25]\\n s_or_b64 s[22:23], s[20:21], s[26:27]\\n " for _ in range(4)), 0))
tests.append(("mix_add_f32_add_f64", "".join("v_add_f32 %%%d, %%%d, %%16\\n v_add_f64 %%%d, %%%d, %%17\\n " % (i, i, 8 + i, 8 + i) for i in range(4)), 8))
tests.append(("mix_cvt_fma_f64", "v_cvt_f64_f32 %8, %0\\n v_fma_f64 %9, %9, %17, %17\\n v_cvt_f32_f64 %1, %10\\n v_fma_f64 %11, %11, %17, %17\\n v_cvt_f64_f32 %12, %2\\n v_fma_f64 %13, %13, %17, %17\\n v_cvt_f32_f64 %3, %14\\n v_fma_f64 %15, %15, %17, %17\\n ", 8))
tests.append(("add_f32_x7_ds_write_b64", "v_add_f32 %0, %0, %16\\n v_add_f32 %1, %1, %16\\n v_add_f32 %2, %2, %16\\n ds_write_b64 %18, %8\\n v_add_f32 %4, %4, %16\\n v_add_f32 %5, %5, %16\\n v_add_f32 %6, %6, %16\\n v_add_f32 %7, %7, %16\\n ", 7))
tests.append(("add_f64_x7_ds_write_b128", "v_add_f64 %8, %8, %17\\n v_add_f64 %9, %9, %17\\n v_add_f64 %10, %10, %17\\n ds_write_b64 %18, %8\\n v_add_f64 %12, %12, %17\\n v_add_f64 %13, %13, %17\\n v_add_f64 %14, %14, %17\\n v_add_f64 %15, %15, %17\\n ", 7))

out = []
for name, body, valu in tests:
    out.append('DEFINE_KERNEL(%s, "%s")' % (name, body))
out.append("static const Test tests[] = {")
for name, body, valu in tests:
    out.append('    {"%s", k_%s, %d},' % (name, name, valu))
out.append("};")
print("\n".join(out))
