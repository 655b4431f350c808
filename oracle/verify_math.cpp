// oracle/verify_math.cpp -- TEST INFRASTRUCTURE ONLY.
// Checks the fast arithmetic of sigtk_amd/csrc/tstat_math.h against plain IEEE division / sqrt:
//   1. sgk_div_f32<W>  == a / (float)W   EXHAUSTIVELY over all finite floats with |a| >= 2^-100 (W = 3, 6, 7, 14, and 2000 for the jnnv2 rolling mean)
//   2. sgk_div_f64<W>  == a / (double)W  on random and structured doubles
//   3. sgk_tstat_tail  == (float)(fabs((double)d)/sqrt((double)v)) with the hardware rsqrt modelled
//      as 1/sqrt(v) perturbed by up to 2^-22 relative (robustness of the run-time certificate)
//   4. sgk_tstat_fast<W> == sgk_tstat_ref<W> on window sums of pA-like data
// Build/run:  g++ -O2 -mfma -ffp-contract=off -fopenmp -o /tmp/verify_math oracle/verify_math.cpp && /tmp/verify_math [quick]
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>

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

int main(int argc, char **argv) {
    const bool quick = argc > 1 && strcmp(argv[1], "quick") == 0;
    const uint32_t step = quick ? 257 : 1;
    const uint64_t n64 = quick ? 4000000ULL : 1000000000ULL;
    uint64_t bad = 0, b;
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
    printf("%s\n", bad ? "FAILED" : "ALL EXACT");
    return bad ? 1 : 0;
}
