// tstat_math_r3.h -- the arithmetic of the round-3 "decisions, not values" experiment (tools/proto/event_r3/README.md).
// Include after sigtk_amd/csrc/tstat_math.h.  Checked by oracle/verify_math.cpp (#10..#12).
#pragma once
// ================================================================ round-3 forms (event_kernels.hip: FastPass)
// The t-statistics never leave the kernel: the detector's DECISIONS do (three threshold tests per index,
// events.c:383-440), and through them the peak positions.  So the fast pass no longer reproduces every rounding of
// compute_tstat -- only the ones that are signal at the scale of a decision:
//   * mean1, mean2 exactly as the reference rounds them (their float roundings are ~1e-6 of a typical delta);
//   * m1sq, m2sq, q2 exactly (float roundings of ~1e4-sized terms inside a variance of a few pA^2: ~1e-4 relative);
//   * the variance accumulation itself ((sumsq1/w - m1sq) + q2) - m2sq is evaluated as ONE correctly rounded f32
//     result of the real-number expression w*cv = (Sq1 - w*m1sq) + w*(q2 - m2sq): the first bracket by an f64 FMA
//     on the exact window sum (then rounded to f32), the second by an f32 subtraction that is exact whenever the
//     two terms are within a factor of two (Sterbenz; always so on a signal whose variance is below its squared mean)
//     and otherwise has a relative error of 2^-24 of a variance-sized term, the sum by an f32 FMA;
//   * the tail |delta| / sqrt(cv/w) = w * |delta| * rsq(w*cv) with the hardware's v_rsq_f32 and one multiply.
// Result: tq = (t/w)(1 + eps), |eps| <= 2^-20.9 (budget below), where t is the reference's float.  The automaton
// runs on tq against thresholds scaled by 1/w; a decision closer to its threshold than the band SGK_BAND(v) is
// "uncertain" (~1e-5 of the indices) and is taken with the reference expression instead (tstat_exact_at).  NaN / inf
// (non-positive or tiny variance: the reference's FLT_MIN floor, constant stretches) fail the band test by
// construction (v_cmp_ngt) and take the exact path as well.
//
// Error budget (u = 2^-24), all relative to t:
//   mine:  rsq 4u (assumed 2^-22; measured 2^-23.3 on gfx950, tools/rsq_check.hip) + product 1u
//          + w*cv: three roundings (va3, d2 when inexact, the FMA) of variance-sized terms, 3.2u, halved by the sqrt: 1.6u
//   reference vs the real-number value: RN32(cv) and RN32(cv/w) (1u each, halved: 1u), the final RN32 (1u),
//          its f64 steps < 2^-50 of a 1e4-sized term: below 2^-26 of cv once cv3 > 2^-17 sumsq (the cv_ok test)
//   total 8.6u = 2^-20.9  ->  SGK_RHO = 2^-20 (band = 2.5 rho (v + ph/w), see sgk_band)
#define SGK_RHO 9.5367431640625e-07f /* 2^-20 */

struct SgkA3 {
    float mean1;  // RN32(RN64(S / w)), as the reference rounds it
    float va3;    // RN32(Sq - w * mean1^2): w times (sumsq1/w - mean1*mean1)
};
// (float)(S * RN64(1/w)) == (float)(S / (double)w) for every S that is an exact sum of w floats spanning at most
// 2^16 in magnitude: RN64(1/w) = (1/w)(1 - 2^-54) exactly for w = 3, 6, 7, 14, so exact quotients come out exact, and
// every other quotient N/w (N an integer below 2^45 in units of the smallest ulp) is further than 2^-49 from a float
// rounding boundary.  oracle/verify_math.cpp (#9) checks it on 10^9 sums per w.
template <int W>
SGK_TM SgkA3 sgk_a3(double S, double Sq) {
    constexpr double r = 1.0 / (double)W;
    SgkA3 a;
    a.mean1 = (float)(S * r);
    const float m1sq = a.mean1 * a.mean1;
    a.va3 = (float)fma(-(double)W, (double)m1sq, Sq);
    return a;
}
// tq = (t / w)(1 + eps) for the window whose B side has the float sums s = RN32(S), sq = RN32(Sq) and whose A side is
// `a`.  cv_ok == false (variance at or near the reference's floor, negative, NaN): take the exact expression.
template <int W>
SGK_TM float sgk_tq(float s, float sq, const SgkA3 &a, bool &cv_ok) {
    const float mean2 = sgk_div_f32<W>(s);
    const float m2sq = mean2 * mean2;
    const float q2 = sgk_div_f32<W>(sq);
    const float d2 = q2 - m2sq;
    const float cv3 = fmaf((float)W, d2, a.va3);
    cv_ok = cv3 > sq * 4.76837158203125e-07f;  // 2^-21: also false for NaN
    const float delta = mean2 - a.mean1;
    const float y = SGK_RSQ32(cv3);
    return fabsf(delta) * y;
}
// half-width of the uncertainty band around a threshold, in units of t/w; phs = peak_height / w
SGK_TM float sgk_band(float v, float phs) { return fmaf(v, 2.5f * SGK_RHO, 2.5f * SGK_RHO * phs); }

// ---- lazy long detector, round 3: the long window [q, q+2w) is two short windows whose FLOAT sums the pass has
// anyway, so the estimates come from two f32 additions instead of two f64 subtractions and two conversions.
//   m~ = RN(sa + sb) = S(1 + 2.01u),  q' = RN((sqa + sqb) * W(1 - k)),  z = RN(q' - m~^2)  (FMA),  k = 2^-19
//   cold  <=  c D~^2 < Z,  D~ = m~B - m~A,  Z = zA + zB,  c = (W/81)(1 + 2^-9)
// Derivation as for sgk_long_cold (the reference's roundings cost <= 7.1u(QA+QB) in the variance and 3.6u(|mA|+|mB|) in
// delta); the estimates add 7.1u Qs + 5.7u Ms, k = 32u covers both variance terms twice, the delta slack goes into the
// 2^-9 by (a+b)^2 <= (1+2^-10) a^2 + 1025 b^2 with b^2 <= 65 u^2 Qs.  A NaN or a non-positive bound fails: hot.
struct SgkL2 {
    float m, z;
};
template <int W>  // W = the long window
SGK_TM SgkL2 sgk_l2(float sa, float sb, float sqa, float sqb) {
    SgkL2 l;
    l.m = sa + sb;
    const float qp = (sqa + sqb) * ((float)W * (1.0f - 1.9073486328125e-06f));
    l.z = fmaf(-l.m, l.m, qp);
    return l;
}
template <int W>
SGK_TM bool sgk_cold2(const SgkL2 &a, const SgkL2 &b) {
    constexpr float rc = (W == 6) ? 0.272432f : ((W == 14) ? 0.41615f : 0.0f);  // > sqrt((W/81)(1 + 2^-9))
    static_assert(W == 6 || W == 14, "long window of the two presets");
    const float D = (b.m - a.m) * rc;
    const float Z = a.z + b.z;
    return fmaf(-D, D, Z) > 0.0f;
}

template <int W>
SGK_TM float sgk_tstat_ref_inl(double A, double A2, double B, double B2) {
    const float wf = (float)W;
    const float sum2 = (float)B;
    const float sumsq2 = (float)B2;
    const float mean1 = (float)(A / (double)wf);
    const float mean2 = sum2 / wf;
    const float m1sq = mean1 * mean1;
    const float m2sq = mean2 * mean2;
    const float q2 = sumsq2 / wf;
    double acc = A2 / (double)wf;
    acc = acc - (double)m1sq;
    acc = acc + (double)q2;
    acc = acc - (double)m2sq;
    float cv = (float)acc;
    cv = fmaxf(cv, 1.17549435e-38f);
    const float delta = mean2 - mean1;
    const float cvw = cv / wf;
    return (float)(fabs((double)delta) / sqrt((double)cvw));
}
