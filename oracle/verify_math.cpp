// oracle/verify_math.cpp -- TEST INFRASTRUCTURE ONLY.
// Checks the fast arithmetic of sigtk_amd/csrc/tstat_math.h against plain IEEE division / sqrt:
//   1. sgk_div_f32<W>  == a / (float)W   EXHAUSTIVELY over all finite floats with |a| >= 2^-100 (W = 3, 6, 7, 14, and 2000 for the jnnv2 rolling mean)
//   2. sgk_div_f64<W>  == a / (double)W  on random and structured doubles
//   3. sgk_tstat_tail  == (float)(fabs((double)d)/sqrt((double)v)) with the hardware rsqrt modelled
//      as 1/sqrt(v) perturbed by up to 2^-22 relative (robustness of the run-time certificate)
//   4. sgk_tstat_fast<W> == sgk_tstat_ref<W> on window sums of pA-like data
//   5. sgk_tail_f32 (f32 error-free-transformation tail, v_rsq_f32 modelled with 2^-22 relative error): certified => equal
//   6. sgk_tstat_try_ab<W> (A side ringed, B side fresh; what LazyPass evaluates): certified => equal to sgk_tstat_ref<W>
//   8. sgk_div_with_rcp == IEEE f32 division by an event length (v_rcp_f32 modelled with +-1 ulp)
//   7. sgk_long_cold<W> true => sgk_tstat_ref<W> <= 9.0 (the lazy long detector's bound), incl. values steered to ~9
// Build/run:  g++ -O2 -mfma -ffp-contract=off -fopenmp -o /tmp/verify_math oracle/verify_math.cpp && /tmp/verify_math [quick]
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <cfloat>

static inline uint64_t rng_next(uint64_t &s) {
    s += 0x9E3779B97F4A7C15ULL;
    uint64_t z = s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static thread_local uint64_t g_pert_state = 12345;
static inline double perturbed_rsq(double v) {
    const uint64_t r = rng_next(g_pert_state);
    const double u = ((double)(int64_t)(r >> 11) / 9007199254740992.0) * 2.0 - 1.0;  // [-1,1)
    return (1.0 / sqrt(v)) * (1.0 + u * 2.384185791015625e-07);                     // 2^-22
}
#define SGK_RSQ64(v) perturbed_rsq(v)
// v_rsq_f32 model: correctly rounded 1/sqrt perturbed by up to 2^-22 relative; every 4th call sits on the edge of that band
static inline float perturbed_rsq32(float v) {
    const uint64_t r = rng_next(g_pert_state);
    double u = ((double)(int64_t)(r >> 11) / 9007199254740992.0) * 2.0 - 1.0;  // [-1,1)
    if ((r & 3) == 0) u = (r & 4) ? 1.0 : -1.0;
    return (float)((1.0 / sqrt((double)v)) * (1.0 + u * 2.384185791015625e-07));
}
#define SGK_RSQ32(v) perturbed_rsq32(v)
// v_rcp_f32 model: the correctly rounded reciprocal moved by -1, 0 or +1 ulp
static inline float perturbed_rcp32(float v) {
    const uint64_t r = rng_next(g_pert_state);
    float x = 1.0f / v;
    uint32_t u; memcpy(&u, &x, 4);
    u += (uint32_t)((int)(r % 3) - 1);
    memcpy(&x, &u, 4);
    return x;
}
#define SGK_RCP32(v) perturbed_rcp32(v)
#include "../sigtk_amd/csrc/tstat_math.h"

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline double u2d(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }

template <int W>
static uint64_t check_div32(uint32_t step) {
    uint64_t bad = 0;
#pragma omp parallel for reduction(+ : bad) schedule(static)
    for (int64_t k = 0; k < 0x7F800000LL; k += step) {
        const float a = u2f((uint32_t)k);
        if (a < SGK_F32_TINY) continue;  // below 2^-100 the kernels take the exact path
        const float want = a / (float)W;
        if (f2u(sgk_div_f32<W>(a)) != f2u(want)) bad++;
        if (f2u(sgk_div_f32<W>(-a)) != f2u(-a / (float)W)) bad++;
    }
    return bad;
}

template <int W>
static uint64_t check_div64(uint64_t count) {
    uint64_t bad = 0;
#pragma omp parallel for reduction(+ : bad) schedule(static)
    for (int64_t t = 0; t < 64; ++t) {
        uint64_t s = 0xABCDEF12345ULL * (t + 1) + W;
        for (uint64_t k = 0; k < count / 64; ++k) {
            double a;
            const uint64_t r = rng_next(s);
            if (k & 1) {  // random mantissa, exponent in a wide range
                const uint64_t mant = r & ((1ULL << 52) - 1);
                const uint64_t ex = 1023 - 300 + (rng_next(s) % 600);
                a = u2d((r & (1ULL << 63)) | (ex << 52) | mant);
            } else {      // exact sum of W float values (what the kernel feeds it)
                a = 0.0;
                uint64_t q = r;
                for (int j = 0; j < W; ++j) {
                    q = rng_next(s);
                    float f = u2f((uint32_t)((q & 0x007FFFFF) | ((uint32_t)(120 + (q >> 40) % 24) << 23)));
                    if (q & (1ULL << 62)) f = f * f;
                    a += (double)f;
                }
            }
            const double want = a / (double)W;
            double got = sgk_div_f64<W>(a);
            if (memcmp(&want, &got, 8) != 0) bad++;
        }
    }
    return bad;
}

static uint64_t check_tail(uint64_t count, double *fast_frac) {
    uint64_t bad = 0, slow = 0;
#pragma omp parallel for reduction(+ : bad, slow) schedule(static)
    for (int64_t t = 0; t < 64; ++t) {
        uint64_t s = 0x13579BDFULL * (t + 3);
        g_pert_state = 777 + t;
        for (uint64_t k = 0; k < count / 64; ++k) {
            const uint64_t r = rng_next(s), r2 = rng_next(s);
            float d, v;
            switch (k & 7) {
                case 0: d = u2f((uint32_t)r & 0x7FFFFFFF); v = u2f((uint32_t)(r2 & 0x7FFFFFFF)); break;  // anything
                case 1: d = (float)((int64_t)(r % 200001) - 100000) * 1e-3f; v = 1.17549435e-38f / 3.0f; break;  // variance floor
                case 2: d = (float)((int64_t)(r % 200001) - 100000) * 1e-3f; v = 1.17549435e-38f / 14.0f; break;
                case 3: d = 0.0f; v = u2f((uint32_t)(r2 & 0x7FFFFFFF)); break;
                default: d = u2f((uint32_t)((r & 0x807FFFFF) | ((uint32_t)(110 + (r >> 40) % 30) << 23)));
                         v = u2f((uint32_t)((r2 & 0x007FFFFF) | ((uint32_t)(105 + (r2 >> 40) % 40) << 23)));
            }
            if (!(v > 0.0f) || !std::isfinite(v) || !std::isfinite(d)) continue;
            const float want = (float)(fabs((double)d) / sqrt((double)v));
            const float got = sgk_tstat_tail(d, v);
            if (f2u(want) != f2u(got)) bad++;
        }
    }
    *fast_frac = 1.0 - (double)slow / (double)count;
    return bad;
}

template <int W>
static uint64_t check_full(uint64_t count) {
    uint64_t bad = 0;
#pragma omp parallel for reduction(+ : bad) schedule(static)
    for (int64_t t = 0; t < 64; ++t) {
        uint64_t s = 0x2468ACE0ULL * (t + 5) + W;
        g_pert_state = 999 + t;
        const float unit = 1402.882324f / 8192.0f;
        for (uint64_t k = 0; k < count / 64; ++k) {
            // two adjacent windows of pA-like samples: level jump of random size, small noise
            const uint64_t r = rng_next(s);
            const int base = 200 + (int)(r % 600), jump = (int)((r >> 20) % 200) - 100, noise = 1 + (int)((r >> 40) % 12);
            const float off = (float)((r >> 52) % 20);
            double A = 0, A2 = 0, B = 0, B2 = 0;
            for (int j = 0; j < 2 * W; ++j) {
                const uint64_t q = rng_next(s);
                int raw = base + (j >= W ? jump : 0) + (int)(q % (uint64_t)(2 * noise + 1)) - noise;
                if ((k % 97) == 0) raw = base;  // exactly constant window -> variance floor
                const float x = ((float)raw + off) * unit;
                const float xq = x * x;
                if (j < W) { A += (double)x; A2 += (double)xq; } else { B += (double)x; B2 += (double)xq; }
            }
            const float want = sgk_tstat_ref<W>(A, A2, B, B2);
            const float got = sgk_tstat_fast<W>(A, A2, B, B2);
            if (f2u(want) != f2u(got)) bad++;
        }
    }
    return bad;
}


// 5. sgk_tail_f32: whenever it certifies, the value equals the reference tail; reports how often it does not certify
static uint64_t check_tail32(uint64_t count, double *uncert_frac) {
    uint64_t bad = 0, slow = 0, tot = 0;
#pragma omp parallel for reduction(+ : bad, slow, tot) schedule(static)
    for (int64_t t = 0; t < 64; ++t) {
        uint64_t s = 0x5151BDFULL * (t + 3);
        g_pert_state = 4242 + t;
        for (uint64_t k = 0; k < count / 64; ++k) {
            const uint64_t r = rng_next(s), r2 = rng_next(s);
            float d, v;
            switch (k & 7) {
                // the domain the kernels guarantee (range guard: non-zero |x| in [2^-20, 2^20]; cv > 2^-90):
                // |delta| = 0 or in [2^-69, 2^21), cvw in [2^-94, 2^41)
                case 0: d = u2f((uint32_t)((r & 0x807FFFFF) | ((uint32_t)(127 - 69 + (r >> 40) % 90) << 23)));
                        v = u2f((uint32_t)((r2 & 0x007FFFFF) | ((uint32_t)(127 - 94 + (r2 >> 40) % 135) << 23))); break;
                case 1: d = 0.0f; v = u2f((uint32_t)((r2 & 0x007FFFFF) | ((uint32_t)(127 - 94 + (r2 >> 40) % 135) << 23))); break;
                case 2: {  // quotient close to a power of two (binade edge of the result)
                    v = u2f((uint32_t)((r2 & 0x007FFFFF) | ((uint32_t)(110 + (r2 >> 40) % 30) << 23)));
                    const double q = ldexp(1.0, (int)(r % 20) - 6) * (1.0 + ((double)(int64_t)((r >> 8) % 65) - 32.0) * 5.9604644775390625e-08);
                    d = (float)(q * sqrt((double)v));
                    break;
                }
                default: d = u2f((uint32_t)((r & 0x807FFFFF) | ((uint32_t)(110 + (r >> 40) % 30) << 23)));
                         v = u2f((uint32_t)((r2 & 0x007FFFFF) | ((uint32_t)(105 + (r2 >> 40) % 40) << 23)));
            }
            if (!(v > 0.0f) || !std::isfinite(v) || !std::isfinite(d)) continue;
            const float want = (float)(fabs((double)d) / sqrt((double)v));
            bool ok;
            const float got = sgk_tail_f32(d, v, ok);
            tot++;
            if (!ok) { slow++; continue; }
            if (f2u(want) != f2u(got)) bad++;
        }
    }
    *uncert_frac = (double)slow / (double)(tot ? tot : 1);
    return bad;
}

// pA-like (or wide-range) adjacent windows -> exact sums
template <int W>
static inline void make_windows(uint64_t &s, uint64_t k, double &A, double &A2, double &B, double &B2) {
    const float unit = 1402.882324f / 8192.0f;
    const uint64_t r = rng_next(s);
    const int base = 200 + (int)(r % 600), jump = (int)((r >> 20) % 200) - 100, noise = 1 + (int)((r >> 40) % 12);
    const float off = (float)((r >> 52) % 20) + (((r >> 57) & 1) ? 0.25f : 0.0f);
    // every 16th case: a wild scale (still inside the read-level range guard 2^-40..2^40)
    const float scale = ((k & 15) == 7) ? ldexpf(1.0f, (int)((r >> 58) % 60) - 30) : 1.0f;
    A = A2 = B = B2 = 0.0;
    for (int j = 0; j < 2 * W; ++j) {
        const uint64_t q = rng_next(s);
        int raw = base + (j >= W ? jump : 0) + (int)(q % (uint64_t)(2 * noise + 1)) - noise;
        if ((k % 97) == 0) raw = base;  // exactly constant window -> variance floor
        if ((k % 89) == 0 && j == 1) raw = -(int)off;  // a zero (or tiny) sample
        const float x = (((float)raw + off) * unit) * scale;
        const float xq = x * x;
        if (j < W) { A += (double)x; A2 += (double)xq; } else { B += (double)x; B2 += (double)xq; }
    }
}

// 6. sgk_tstat_try_ab<W>: certified values equal the reference expression
template <int W>
static uint64_t check_try_ab(uint64_t count, double *uncert_frac) {
    uint64_t bad = 0, slow = 0;
#pragma omp parallel for reduction(+ : bad, slow) schedule(static)
    for (int64_t t = 0; t < 64; ++t) {
        uint64_t s = 0x77AA33ULL * (t + 5) + W;
        g_pert_state = 31337 + t;
        for (uint64_t k = 0; k < count / 64; ++k) {
            double A, A2, B, B2;
            make_windows<W>(s, k, A, A2, B, B2);
            const float want = sgk_tstat_ref<W>(A, A2, B, B2);
            const SgkARole ar = sgk_arole<W>(A, A2);
            bool ok;
            const float got = sgk_tstat_try_ab<W>(B, B2, ar, ok);
            if (!ok) { slow++; continue; }
            if (f2u(want) != f2u(got)) bad++;
        }
    }
    *uncert_frac = (double)slow / (double)count;
    return bad;
}

// 7. sgk_long_cold<W> == true  =>  the reference t-statistic is <= 9.0; reports how often the test says "hot" when
//    the reference value is in fact <= 8 (its slack)
template <int W>
static uint64_t check_long_cold(uint64_t count, double *false_hot) {
    uint64_t bad = 0, fh = 0, low = 0;
#pragma omp parallel for reduction(+ : bad, fh, low) schedule(static)
    for (int64_t t = 0; t < 64; ++t) {
        uint64_t s = 0x99BB11ULL * (t + 7) + W;
        for (uint64_t k = 0; k < count / 64; ++k) {
            double A, A2, B, B2;
            make_windows<W>(s, k, A, A2, B, B2);
            if ((k & 3) == 1) {
                // steer the statistic to the neighbourhood of 9: shift window B by the right amount
                const float t0 = sgk_tstat_ref<W>(A, A2, B, B2);
                if (t0 > 0.5f && t0 < 1e6f) {
                    const double target = 9.0 * (1.0 + ((double)(int64_t)(rng_next(s) % 2001) - 1000.0) * 1e-5);
                    const float dx = (float)((B - A) / W * (target / t0 - 1.0));
                    const float unit = 1402.882324f / 8192.0f;
                    const float dq = roundf(dx / unit) * unit;
                    // shifting every sample of B by dq changes the mean, not the variance (to first order)
                    double nB = 0, nB2 = 0;
                    const float mb = (float)(B / W);
                    for (int j = 0; j < W; ++j) { const float x = mb + dq + (float)(j - W / 2) * unit; nB += (double)x; nB2 += (double)(x * x); }
                    (void)nB; (void)nB2;  // keep the original variance structure: only move the sums consistently
                    B2 = B2 + 2.0 * (double)dq * B + (double)W * (double)dq * (double)dq;
                    B = B + (double)W * (double)dq;
                }
            }
            const float ref = sgk_tstat_ref<W>(A, A2, B, B2);
            const bool cold = sgk_long_cold<W>(sgk_lside<W>(A, A2), sgk_lside<W>(B, B2));
            if (cold && !(ref <= 9.0f)) bad++;
            if (ref <= 8.0f) { low++; if (!cold) fh++; }
        }
    }
    *false_hot = (double)fh / (double)(low ? low : 1);
    return bad;
}

// 8. sgk_div_with_rcp(a, len, sgk_refined_rcp(len)) == a / len for integer len in [1, 2^24) and a = 0 or 2^-43 <= |a| <= 2^44
static uint64_t check_div_len(uint64_t count) {
    uint64_t bad = 0;
#pragma omp parallel for reduction(+ : bad) schedule(static)
    for (int64_t t = 0; t < 64; ++t) {
        uint64_t s = 0xD1F1ULL * (t + 11);
        g_pert_state = 90210 + t;
        for (uint64_t k = 0; k < count / 64; ++k) {
            const uint64_t r = rng_next(s), r2 = rng_next(s);
            float len;
            switch (k & 3) {
                case 0: len = (float)(1 + (r % 64)); break;            // typical event lengths
                case 1: len = (float)(1 + (r % 4096)); break;
                default: len = (float)(1 + (r % 16777215));
            }
            float a = u2f((uint32_t)((r2 & 0x807FFFFF) | ((uint32_t)(127 - 43 + (r2 >> 40) % 88) << 23)));
            if ((k % 1001) == 0) a = 0.0f;
            if ((k & 7) == 5) a = len * (float)(1 + (r2 % 100000)) * 0.171249f;  // sums of pA-like values
            const float want = a / len;
            const float got = sgk_div_with_rcp(a, len, sgk_refined_rcp(len));
            if (f2u(want) != f2u(got)) bad++;
        }
    }
    return bad;
}


// ---------------------------------------------------------------- round 3 (the one piece of it that is in the product)
// 9. the A-side mean (sgk_arole<W, SHORT = true>): (float)(S * RN64(1/W)) == (float)(S / (double)W) for exact sums of W
//    floats spanning <= 2^16
template <int W>
static uint64_t check_mean1(uint64_t count) {
    uint64_t bad = 0;
#pragma omp parallel for reduction(+ : bad) schedule(static)
    for (int64_t t = 0; t < 64; ++t) {
        uint64_t s = 0xA3A3ULL * (t + 1) + W;
        for (uint64_t k = 0; k < count / 64; ++k) {
            const uint64_t r = rng_next(s);
            const int e0 = 100 + (int)(r % 40), span = 1 + (int)((r >> 8) % 16);
            double S = 0.0, Sq = 0.0;
            float base = 0.0f;
            for (int j = 0; j < W; ++j) {
                const uint64_t q = rng_next(s);
                float f = u2f((uint32_t)((q & 0x007FFFFF) | ((uint32_t)(e0 + (q >> 40) % span) << 23)));
                if ((k & 7) == 3) { if (j == 0) base = f; else f = base + (float)((int)(q % 7) - 3) * (base * 1.1920929e-07f); }  // near-equal samples
                if ((k & 7) == 5) f = u2f(f2u(f) & 0xFFFFF000u);                                      // short mantissas: exact quotients, ties
                S += (double)f;
                Sq += (double)(f * f);
            }
            const SgkARole a = sgk_arole<W, true>(S, Sq);   // the product's form (event_kernels.hip: LazyPass, fast path)
            const float want = (float)(S / (double)(float)W);
            if (f2u(a.mean1) != f2u(want)) bad++;
        }
    }
    return bad;
}

int main(int argc, char **argv) {
    const bool quick = argc > 1 && strcmp(argv[1], "quick") == 0;
    const uint32_t step = quick ? 257 : 1;
    const uint64_t n64 = quick ? 4000000ULL : 1000000000ULL;
    uint64_t bad = 0, b;
    double uf;
    b = check_div32<3>(step);  printf("div_f32<3>   mismatches: %llu\n", (unsigned long long)b); bad += b;
    b = check_div32<6>(step);  printf("div_f32<6>   mismatches: %llu\n", (unsigned long long)b); bad += b;
    b = check_div32<7>(step);  printf("div_f32<7>   mismatches: %llu\n", (unsigned long long)b); bad += b;
    b = check_div32<14>(step); printf("div_f32<14>  mismatches: %llu\n", (unsigned long long)b); bad += b;
    b = check_div32<2000>(step); printf("div_f32<2000> mismatches: %llu\n", (unsigned long long)b); bad += b;  // jnnv2 rolling mean
    b = check_div64<3>(n64);   printf("div_f64<3>   mismatches: %llu\n", (unsigned long long)b); bad += b;
    b = check_div64<6>(n64);   printf("div_f64<6>   mismatches: %llu\n", (unsigned long long)b); bad += b;
    b = check_div64<7>(n64);   printf("div_f64<7>   mismatches: %llu\n", (unsigned long long)b); bad += b;
    b = check_div64<14>(n64);  printf("div_f64<14>  mismatches: %llu\n", (unsigned long long)b); bad += b;
    double ff;
    b = check_tail(n64, &ff);  printf("tstat_tail   mismatches: %llu\n", (unsigned long long)b); bad += b;
    b = check_full<3>(n64 / 4);  printf("tstat_fast<3>  mismatches: %llu\n", (unsigned long long)b); bad += b;
    b = check_full<6>(n64 / 4);  printf("tstat_fast<6>  mismatches: %llu\n", (unsigned long long)b); bad += b;
    b = check_full<7>(n64 / 4);  printf("tstat_fast<7>  mismatches: %llu\n", (unsigned long long)b); bad += b;
    b = check_full<14>(n64 / 4); printf("tstat_fast<14> mismatches: %llu\n", (unsigned long long)b); bad += b;
    b = check_tail32(n64, &uf);        printf("tail_f32     mismatches: %llu (uncertified %.3g)\n", (unsigned long long)b, uf); bad += b;
    b = check_try_ab<3>(n64 / 4, &uf); printf("tstat_try_ab<3>  mismatches: %llu (uncertified %.3g)\n", (unsigned long long)b, uf); bad += b;
    b = check_try_ab<7>(n64 / 4, &uf); printf("tstat_try_ab<7>  mismatches: %llu (uncertified %.3g)\n", (unsigned long long)b, uf); bad += b;
    b = check_long_cold<6>(n64 / 4, &uf);  printf("long_cold<6>  violations: %llu (hot although ref <= 8: %.3g)\n", (unsigned long long)b, uf); bad += b;
    b = check_long_cold<14>(n64 / 4, &uf); printf("long_cold<14> violations: %llu (hot although ref <= 8: %.3g)\n", (unsigned long long)b, uf); bad += b;
    b = check_div_len(n64);            printf("div_with_rcp (event length) mismatches: %llu\n", (unsigned long long)b); bad += b;
    // the A side's mean by one multiply (sgk_arole<W, SHORT>, round 3)
    b = check_mean1<3>(n64);  printf("arole mean1<3>  mismatches: %llu\n", (unsigned long long)b); bad += b;
    b = check_mean1<6>(n64);  printf("arole mean1<6>  mismatches: %llu\n", (unsigned long long)b); bad += b;
    b = check_mean1<7>(n64);  printf("arole mean1<7>  mismatches: %llu\n", (unsigned long long)b); bad += b;
    b = check_mean1<14>(n64); printf("arole mean1<14> mismatches: %llu\n", (unsigned long long)b); bad += b;
    printf("%s\n", bad ? "FAILED" : "ALL EXACT");
    return bad ? 1 : 0;
}
