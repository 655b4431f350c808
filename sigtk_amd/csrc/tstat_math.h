// tstat_math.h -- bit-exact fast forms of the arithmetic in compute_tstat (src/events.c:338-361).
// Shared by the device kernels and by the host-side verification tool (oracle/verify_math.c), so the
// exactness claims below are checked against plain IEEE division / sqrt on the CPU:
//
//  div3/6/7/14 in f64 and f32: correctly rounded division by a small constant in three FMA-class ops
//      q = a*r; rem = fma(-q, d, a); q' = fma(rem, r, q)          (r = RN(1/d))
//    (Markstein's correction step: rem is exact, q' = RN(a/d)).  f32 inputs below 2^-100 take a true
//    division (the correction is not exact in the subnormal range).  Verified exhaustively for f32
//    and on 4e9 random/structured inputs for f64 (oracle/verify_math.c).
//
//  tail(delta, cvw) = (float)( fabs((double)delta) / sqrt((double)cvw) ): the reference rounds the
//    sqrt and the quotient to double and then the quotient to float.  Fast path: y = rsqrt(cvw)
//    refined by one Newton step in f64 (relative error < 2^-40, certified at run time from the
//    residual e = 1 - v*y0^2), q = |delta|*y.  The reference's double quotient is within 2^-39 of q,
//    so unless q lies within 2^15 double-ulps of a float rounding midpoint, (float)q IS the reference
//    result.  Otherwise (probability ~2^-13), or when the residual test fails or the result is
//    outside the normal float range, the exact expression is evaluated.
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define SGK_TM __host__ __device__ inline
#define SGK_TM_COLD __host__ __device__ __attribute__((noinline))
#else
#define SGK_TM static inline
#define SGK_TM_COLD static __attribute__((noinline))
#endif

#ifndef SGK_RSQ64
#if defined(__HIP_DEVICE_COMPILE__)
#define SGK_RSQ64(v) __builtin_amdgcn_rsq(v)
#else
#define SGK_RSQ64(v) (1.0 / sqrt(v))
#endif
#endif

SGK_TM uint64_t sgk_d2u(double x) {
    uint64_t u;
    memcpy(&u, &x, 8);
    return u;
}

template <int W>
SGK_TM double sgk_div_f64(double a) {
    constexpr double d = (double)W;
    constexpr double r = 1.0 / d;
    const double q = a * r;
    const double rem = fma(-q, d, a);
    return fma(rem, r, q);
}

// a / (float)W for |a| >= 2^-100 (callers route smaller inputs, zero included, to the exact path)
template <int W>
SGK_TM float sgk_div_f32(float a) {
    constexpr float d = (float)W;
    constexpr float r = 1.0f / d;
    const float q = a * r;
    const float rem = fmaf(-q, d, a);
    return fmaf(rem, r, q);
}
#define SGK_F32_TINY 7.8886090522101181e-31f /* 2^-100 */

SGK_TM uint32_t sgk_hi32(double x) { return (uint32_t)(sgk_d2u(x) >> 32); }

// Fast form of (float)( fabs((double)delta) / sqrt((double)cvw) ) plus its certificate.
// The range / residual / midpoint tests are done on the bit patterns (integer ops issue at the f32
// rate, f64 compares at the f64 rate).
SGK_TM float sgk_tail_fast(float delta, float cvw, bool &ok) {
    const double ad = fabs((double)delta);
    const double v = (double)cvw;
    const double y0 = SGK_RSQ64(v);
    const double t = v * y0;
    const double e = fma(-t, y0, 1.0);
    const double y1 = fma(y0, 0.5 * e, y0);
    const double q = ad * y1;
    // |e| < 2^-20  <=>  biased exponent of e below that of 2^-20 (0x3EB00000 is the high word of 2^-20)
    const bool e_ok = (sgk_hi32(e) & 0x7FFFFFFFu) < 0x3EB00000u;
    // 2^-120 <= q < 2^127 (high words 0x38700000 / 0x47E00000): float result comfortably normal
    const bool q_ok = (sgk_hi32(q) - 0x38700000u) < (0x47E00000u - 0x38700000u);
    // q not within 2^15 double-ulps of a float rounding midpoint (low 29 bits near 2^28)
    const uint32_t lo29 = (uint32_t)sgk_d2u(q) & 0x1FFFFFFFu;
    const bool mid = (uint32_t)(lo29 - (0x10000000u - 0x8000u)) <= 0x10000u;
    // delta == 0 gives exactly 0 whenever the residual certificate holds (v is finite and positive)
    ok = e_ok & ((q_ok & !mid) | (delta == 0.0f));
    return (float)q;
}
// standalone tail with its exact fallback (used by the verification tool)
SGK_TM float sgk_tstat_tail(float delta, float cvw) {
    bool ok;
    const float tq = sgk_tail_fast(delta, cvw, ok);
    if (ok) return tq;
    __asm__ volatile("" ::: "memory");
    return (float)(fabs((double)delta) / sqrt((double)cvw));
}

template <int W>
SGK_TM_COLD float sgk_tstat_ref(double A, double A2, double B, double B2);

// compute_tstat's expression tree for one index, given the four exact window sums: fast value and
// its certificate.  ok == false means "evaluate the reference expression instead".
template <int W>
SGK_TM float sgk_tstat_try(double A, double A2, double B, double B2, bool &ok) {
    const float sum2 = (float)B;
    const float sumsq2 = (float)B2;
    const float mean1 = (float)sgk_div_f64<W>(A);
    const float mean2 = sgk_div_f32<W>(sum2);
    const float m1sq = mean1 * mean1;
    const float m2sq = mean2 * mean2;
    const float q2 = sgk_div_f32<W>(sumsq2);
    double acc = sgk_div_f64<W>(A2);
    acc = acc - (double)m1sq;
    acc = acc + (double)q2;
    acc = acc - (double)m2sq;
    float cv = (float)acc;
    cv = fmaxf(cv, 1.17549435e-38f);  // FLT_MIN
    const float delta = mean2 - mean1;
    // the variance floor is by far the most common tiny input: its quotient is a constant
    constexpr float floor_q = 1.17549435e-38f / (float)W;
    const bool at_floor = cv == 1.17549435e-38f;
    const float cvw = at_floor ? floor_q : sgk_div_f32<W>(cv);
    bool tail_ok;
    const float tq = sgk_tail_fast(delta, cvw, tail_ok);
    // tiny (or zero) dividends take the exact path; the variance floor is handled above
    const float cvx = at_floor ? 1.0f : cv;
    ok = (fminf(fminf(fabsf(sum2), sumsq2), cvx) >= SGK_F32_TINY) & tail_ok;
    return tq;
}

// Both windows of one index at once (W and 2W, as in both detector presets).  On the device the f32
// part of the two expression trees is evaluated on 2-vectors so that it maps onto the packed f32
// instructions (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32: two IEEE operations per issue slot);
// every lane of a packed operation is the same correctly rounded IEEE operation as its scalar form,
// so the results are those of sgk_tstat_try<W> and sgk_tstat_try<2W>.
template <int W>
SGK_TM void sgk_tstat_try_pair(double A1, double A1q, double B1, double B1q, double A2, double A2q, double B2,
                               double B2q, float &v1, float &v2, bool &ok1, bool &ok2) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef float f2 __attribute__((ext_vector_type(2)));
    constexpr float FMIN = 1.17549435e-38f;
    const f2 d = {(float)W, (float)(2 * W)};
    const f2 r = {1.0f / (float)W, 1.0f / (float)(2 * W)};
    auto div2 = [&](f2 a) -> f2 {
        const f2 q = a * r;
        const f2 rem = __builtin_elementwise_fma(-q, d, a);
        return __builtin_elementwise_fma(rem, r, q);
    };
    const f2 sum2 = {(float)B1, (float)B2};
    const f2 sumsq2 = {(float)B1q, (float)B2q};
    const f2 mean1 = {(float)sgk_div_f64<W>(A1), (float)sgk_div_f64<2 * W>(A2)};
    const f2 mean2 = div2(sum2);
    const f2 m1sq = mean1 * mean1;
    const f2 m2sq = mean2 * mean2;
    const f2 q2 = div2(sumsq2);
    double acc1 = sgk_div_f64<W>(A1q);
    double acc2 = sgk_div_f64<2 * W>(A2q);
    acc1 = acc1 - (double)m1sq.x;
    acc2 = acc2 - (double)m1sq.y;
    acc1 = acc1 + (double)q2.x;
    acc2 = acc2 + (double)q2.y;
    acc1 = acc1 - (double)m2sq.x;
    acc2 = acc2 - (double)m2sq.y;
    f2 cv = {fmaxf((float)acc1, FMIN), fmaxf((float)acc2, FMIN)};
    const f2 delta = mean2 - mean1;
    const f2 cvd = div2(cv);
    const bool fl1 = cv.x == FMIN, fl2 = cv.y == FMIN;
    const float cvw1 = fl1 ? FMIN / (float)W : cvd.x;
    const float cvw2 = fl2 ? FMIN / (float)(2 * W) : cvd.y;
    bool t1ok, t2ok;
    v1 = sgk_tail_fast(delta.x, cvw1, t1ok);
    v2 = sgk_tail_fast(delta.y, cvw2, t2ok);
    ok1 = (fminf(fminf(fabsf(sum2.x), sumsq2.x), fl1 ? 1.0f : cv.x) >= SGK_F32_TINY) & t1ok;
    ok2 = (fminf(fminf(fabsf(sum2.y), sumsq2.y), fl2 ? 1.0f : cv.y) >= SGK_F32_TINY) & t2ok;
#else
    v1 = sgk_tstat_try<W>(A1, A1q, B1, B1q, ok1);
    v2 = sgk_tstat_try<2 * W>(A2, A2q, B2, B2q, ok2);
#endif
}

// ================================================================ round-2 forms (event_kernels.hip: LazyPass)
// Measured issue costs on gfx950 (tools/valu_rate.hip, profiles/archive/r02_valu_rate.txt): plain f32 add/sub/mul/fma,
// logic and int add run at 2.3 cycles per wave64 instruction; everything f64, every conversion, v_cmp, v_cndmask,
// v_max/min and the packed f32 forms take 4.45; v_rsq_f32 8.5; v_rsq_f64 16.2.  The forms below keep the f64 work to
// what exactness needs (window sums, the two constant divisions of the A side, the three-term accumulation) and do
// the tail in f32 with error-free transformations on the fast FMA.

#ifndef SGK_RSQ32
#if defined(__HIP_DEVICE_COMPILE__)
#define SGK_RSQ32(v) __builtin_amdgcn_rsqf(v)
#else
#define SGK_RSQ32(v) ((float)(1.0 / sqrt((double)(v))))
#endif
#endif

// (float)( fabs((double)delta) / sqrt((double)cvw) ) in f32 arithmetic, with a certificate.
// Domain (guaranteed by the callers: cv > 2^-90 and the read-level range guard, non-zero |x| in [2^-20, 2^20]):
// delta == 0 or |delta| in [2^-69, 2^21), cvw in [2^-94, 2^41) -- no intermediate below is subnormal there.
//   y0 = rsq(c) (relative error eta <= 2^-22 assumed; measured exhaustively on the hardware: 2^-23.3,
//        tools/rsq_check.hip, asserted by tests/test_gpu_math.py)
//   sqrt(c) = s0 + sl,  s0 = RN(c*y0),  sl = RN((c - s0^2) * y0/2)       relative error <= 2.5 (eta+u)^2
//   q = |d| / sqrt(c) = q0 + ql,  q0 = RN(|d|*y0),  ql = RN((|d| - q0*(s0+sl)) * y0)   relative error < 2^-41
// (residuals by FMA; with the hardware's eta = 2^-23.3, tools/rsq_check.hip, the total is below 2^-43).  The
// reference value is RN32 of a double within 2^-51 of the true quotient.  m = 2^-39 q0:
// if RN(q0 + (ql - m)) == RN(q0 + (ql + m)) no float rounding boundary lies within the error band, so that common
// value IS the reference's float (rounding is monotone); otherwise (probability ~2^-14) the caller evaluates the
// reference expression.  NaN anywhere fails the equality and takes the exact path as well.
SGK_TM float sgk_tail_f32(float delta, float cvw, bool &ok) {
    const float ad = fabsf(delta);
    const float y0 = SGK_RSQ32(cvw);
    const float s0 = cvw * y0;
    const float e = fmaf(-s0, s0, cvw);
    const float yh = 0.5f * y0;
    const float sl = e * yh;
    const float q0 = ad * y0;
    const float r1 = fmaf(-q0, s0, ad);
    const float rho = fmaf(-q0, sl, r1);
    const float ql = rho * y0;
    const float m = q0 * 1.8189894035458565e-12f;  // 2^-39
    const float lo = q0 + (ql - m);
    const float hi = q0 + (ql + m);
    ok = lo == hi;
    return lo;
}

// The A side of compute_tstat for one window position: mean1 and (sumsq1/w - mean1*mean1) as the reference
// rounds them (events.c:341-352); computed when the window sum is formed and used W indices later.
struct SgkARole {
    float mean1;
    double va;
};
// SHORT: S is an exact sum of w floats spanning at most 2^16 in magnitude (the fast path's range guard).  Then
// (float)(S * RN64(1/w)) == (float)(S / (double)w): RN64(1/w) = (1/w)(1 - 2^-54) exactly for w = 3, 6, 7, 14, so exact
// quotients come out exact, and every other quotient N/w (N an integer below 2^45 in units of the smallest ulp) is
// further than 2^-49 from a float rounding boundary -- one f64 multiply instead of the three-operation division.
// oracle/verify_math.cpp (#9) checks it on 10^9 sums per w.
template <int W, bool SHORT = false>
SGK_TM SgkARole sgk_arole(double S, double Sq) {
    SgkARole a;
    if (SHORT) {
        constexpr double r = 1.0 / (double)W;
        a.mean1 = (float)(S * r);
    } else {
        a.mean1 = (float)sgk_div_f64<W>(S);
    }
    const float m1sq = a.mean1 * a.mean1;
    a.va = sgk_div_f64<W>(Sq) - (double)m1sq;
    return a;
}
#define SGK_CV_MIN 8.0779356694631609e-28f /* 2^-90: below it (variance floor included) the exact path runs */

// One t-statistic from the B-side window sums (S, Sq: exact) and the ringed A side.  ok == false: evaluate the
// reference expression instead.  Preconditions (read-level guard, event_kernels.hip): every non-zero |x| in
// [2^-20, 2^20], so window sums are 0 or >= 2^-43, the f32 constant divisions below never see a non-zero dividend
// under 2^-100, and a non-zero delta is >= 2^-69.
template <int W>
SGK_TM float sgk_tstat_try_ab(double S, double Sq, const SgkARole &a, bool &ok) {
    const float sum2 = (float)S;
    const float sumsq2 = (float)Sq;
    const float mean2 = sgk_div_f32<W>(sum2);
    const float m2sq = mean2 * mean2;
    const float q2 = sgk_div_f32<W>(sumsq2);
    double acc = a.va + (double)q2;
    acc = acc - (double)m2sq;
    const float cv = (float)acc;
    const bool cv_ok = cv > SGK_CV_MIN;  // also false for NaN
    const float delta = mean2 - a.mean1;
    const float cvw = sgk_div_f32<W>(cv);
    bool tail_ok;
    const float v = sgk_tail_f32(delta, cvw, tail_ok);
    ok = cv_ok & tail_ok;
    return v;
}

// Fallback kernel only (reads outside the range guard): are this evaluation's intermediates inside the domain the
// certificates above assume?  (Same expressions as sgk_tstat_try_ab: the compiler merges them.)
SGK_TM uint32_t sgk_f2u(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    return u;
}
SGK_TM bool sgk_band60(float x) {  // x == 0 or 2^-60 <= |x| < 2^60
    const uint32_t e = (sgk_f2u(x) >> 23) & 0xffu;
    return x == 0.0f || (e - 67u) < 120u;
}
template <int W>
SGK_TM bool sgk_try_domain(double S, double Sq, const SgkARole &a) {
    const float sum2 = (float)S;
    const float sumsq2 = (float)Sq;
    const float mean2 = sgk_div_f32<W>(sum2);
    const float m2sq = mean2 * mean2;
    const float q2 = sgk_div_f32<W>(sumsq2);
    double acc = a.va + (double)q2;
    acc = acc - (double)m2sq;
    const float cv = (float)acc;
    const float delta = mean2 - a.mean1;
    const uint32_t ed = (sgk_f2u(delta) >> 23) & 0xffu;
    const bool d_ok = delta == 0.0f || (ed - 67u) < 100u;  // 0 or 2^-60 <= |delta| < 2^40
    return sgk_band60(sum2) & sgk_band60(sumsq2) & d_ok & (cv < 1152921504606846976.0f);  // cv < 2^60
}

// ---- create_event (events.c:457-473): two correctly rounded f32 divisions by the same event length ----
// The core of the IEEE f32 division expansion (reciprocal refined once, quotient refined twice, all residuals by
// FMA) without its scaling / fix-up wrapper: valid while nothing under- or overflows, which the range guard of the
// fast path ensures (len is an integer in [1, 2^24); |a| is 0 or in [2^-43, 2^44]).  Verified against `/` by
// oracle/verify_math.cpp with the hardware reciprocal modelled with 1 ulp of error.
#ifndef SGK_RCP32
#if defined(__HIP_DEVICE_COMPILE__)
#define SGK_RCP32(v) __builtin_amdgcn_rcpf(v)
#else
#define SGK_RCP32(v) (1.0f / (v))
#endif
#endif
SGK_TM float sgk_refined_rcp(float b) {
    const float r0 = SGK_RCP32(b);
    const float e = fmaf(-b, r0, 1.0f);
    return fmaf(e, r0, r0);
}
SGK_TM float sgk_div_with_rcp(float a, float b, float r1) {
    const float q0 = a * r1;
    const float rem0 = fmaf(-b, q0, a);
    const float q1 = fmaf(rem0, r1, q0);
    const float rem1 = fmaf(-b, q1, a);
    return fmaf(rem1, r1, q1);
}

// ---- lazy long detector: a rigorous "cannot exceed thr2" test ----------------------------------------
// The long detector (events.c:371-443 with the second window) can only emit a peak when some t-statistic it saw
// since its last reset exceeded thr2 = 9.0 (valid_peak needs peak_value > threshold).  On nanopore data it is reset
// by the short detector every few samples and that almost never happens (2e-4 of the indices), so the kernel
// evaluates the long window exactly only inside such runs and otherwise proves, per index, that the reference's
// value cannot exceed 9:
//   per window position, from the exact sums:  m = RN32(S), q = RN(RN32(Sq)*w), v = RN(q - m*m) (FMA): w times the
//   mean, w^2 times the mean square and the variance -- the test is homogeneous, so nothing is divided
//   reference:  cv >= (vA + vB) - E,  E <= 7.1u(QA+QB)  (its float roundings of mean^2 and sumsq/w), our estimates
//   add <= 10.3u(QA+QB) + u V;  |delta| <= |mB - mA| + 3.6u(|mA|+|mB|) likewise;  tstat <= |delta| sqrt(w/cv) (1+3u).
//   With u = 2^-24 the slack terms below (2^-19 (qA+qB), 2^-20 (|mA|+|mB|), 2^-16 relative) cover all of it
//   several times over; a NaN or a non-positive variance bound makes the test fail, i.e. the index counts as hot.
struct SgkLSide {
    float m, q, v;
};
template <int W>
SGK_TM SgkLSide sgk_lside(double S, double Sq) {
    // (everything scaled by W: the test below is homogeneous, so the sums are used as they are)
    SgkLSide s;
    s.m = (float)S;
    s.q = (float)Sq * (float)W;
    s.v = fmaf(-s.m, s.m, s.q);
    return s;
}
SGK_TM bool sgk_lside_domain(const SgkLSide &s) {  // fallback kernel only: estimates free of subnormal effects
    return sgk_band60(s.q) & sgk_band60(s.m);
}
template <int W>
SGK_TM bool sgk_long_cold(const SgkLSide &a, const SgkLSide &b) {
    const float V = a.v + b.v;
    const float Qs = a.q + b.q;
    const float rhs = fmaf(Qs, -81.0f * 1.9073486328125e-06f, 81.0f * V);  // 81 (V - 2^-19 (qA+qB))
    const float D = b.m - a.m;
    const float Ms = fabsf(a.m) + fabsf(b.m);
    const float Dub = fmaf(Ms, 9.5367431640625e-07f, fabsf(D));             // |D| + 2^-20 (|mA|+|mB|)
    const float lhs = (Dub * Dub) * ((float)W * (1.0f + 1.52587890625e-05f));
    return lhs < rhs;
}

template <int W>
SGK_TM float sgk_tstat_fast(double A, double A2, double B, double B2) {
    bool ok;
    const float tq = sgk_tstat_try<W>(A, A2, B, B2, ok);
    if (ok) return tq;
    __asm__ volatile("" ::: "memory");
    return sgk_tstat_ref<W>(A, A2, B, B2);
}

// the reference expression, operator by operator (used as the exact slow path and as the yardstick)
template <int W>
SGK_TM_COLD float sgk_tstat_ref(double A, double A2, double B, double B2) {
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
