/* fmt.h -- exact fast number formatting for the TSV writers.
 *
 * The reference prints every float with printf("%f") (src/cfunc.c:26,56,93,150,...), i.e. the float
 * promoted to double, six decimals, the exact decimal expansion rounded half-to-even (glibc rounds
 * the exact value in the current rounding mode).  For a value that started life as a float this
 * needs no big-number arithmetic:
 *   v = ip + frac with ip = trunc(v) exact (|v| < 2^53) and frac = v - ip exact; frac has at most 24
 *   significant bits, and 10^6 = 15625 * 2^6, so frac * 10^6 is a product of a 24-bit and a 14-bit
 *   integer times a power of two: EXACT in double.  rint() of it (round-half-even) is therefore the
 *   correctly rounded 6-decimal fraction; a carry into ip when it reaches 10^6.
 * Anything outside |v| < 1e15 (incl. inf/nan) goes through snprintf.  `sigtk-amd _fmtcheck stride [first last]`
 * checks the routine against snprintf; it has been run over ALL 2^32 float bit patterns (stride 1, eight ranges in
 * parallel, 0 mismatches), and a strided run plus the tie/carry cases is part of tests/test_cli_cpu.py.
 */
#ifndef SGK_FMT_H
#define SGK_FMT_H

#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

static const char FMT_DIGITS2[201] =
    "00010203040506070809101112131415161718192021222324252627282930313233343536373839"
    "40414243444546474849505152535455565758596061626364656667686970717273747576777879"
    "8081828384858687888990919293949596979899";

/* decimal digits of v; returns the end pointer */
static inline char *fmt_u64(char *p, uint64_t v) {
    char tmp[20];
    int n = 0;
    while (v >= 100) {
        const unsigned d = (unsigned)(v % 100);
        v /= 100;
        tmp[n++] = FMT_DIGITS2[d * 2 + 1];
        tmp[n++] = FMT_DIGITS2[d * 2];
    }
    if (v >= 10) {
        tmp[n++] = FMT_DIGITS2[v * 2 + 1];
        tmp[n++] = FMT_DIGITS2[v * 2];
    } else {
        tmp[n++] = (char)('0' + v);
    }
    while (n) *p++ = tmp[--n];
    return p;
}

static inline char *fmt_i64(char *p, int64_t v) {
    if (v < 0) {
        *p++ = '-';
        return fmt_u64(p, (uint64_t)0 - (uint64_t)v);
    }
    return fmt_u64(p, (uint64_t)v);
}

/* printf("%f", (double)f): at most 48 bytes are written in the fast range */
static inline char *fmt_f6(char *p, float f) {
    double v = (double)f;
    if (!(fabs(v) < 1e15)) return p + sprintf(p, "%f", v); /* huge, inf, nan */
    if (signbit(v)) {
        *p++ = '-';
        v = -v;
    }
    uint64_t ip = (uint64_t)v;
    const double frac = v - (double)ip;
    uint32_t fr = (uint32_t)rint(frac * 1e6); /* exact product, round-half-even */
    if (fr >= 1000000u) {
        fr -= 1000000u;
        ip += 1;
    }
    p = fmt_u64(p, ip);
    *p++ = '.';
    const unsigned a = fr / 10000u, b = (fr / 100u) % 100u, c = fr % 100u;
    memcpy(p, FMT_DIGITS2 + a * 2, 2);
    memcpy(p + 2, FMT_DIGITS2 + b * 2, 2);
    memcpy(p + 4, FMT_DIGITS2 + c * 2, 2);
    return p + 6;
}

#endif
