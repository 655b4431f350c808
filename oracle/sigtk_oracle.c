/* oracle/sigtk_oracle.c -- TEST INFRASTRUCTURE ONLY (see sigtk_oracle.h).
 *
 * Plain-C restatement of the sigtk hot path.  Every function names the reference
 * file:line whose behaviour it follows.  Arithmetic notes that matter for parity:
 *   - the reference is built -std=c99 on x86-64/SSE2: FLT_EVAL_METHOD == 0 and no
 *     FMA contraction, so every C operator rounds once to its C type.  This file
 *     must be compiled with -ffp-contract=off (oracle/Makefile does).
 *   - float/double mixing is spelled out with explicit casts where the reference
 *     relies on the usual arithmetic conversions.
 */
#include "sigtk_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ pA scaling */

/* src/misc.c:15-32: the three doubles are narrowed to float first (:17-19), the unit
 * is a float division (:26), each sample is (float)raw + offset, then * unit (:28). */
void orc_pa(const int16_t *raw, int64_t n, double digitisation, double offset, double range, float *out) {
    const float rangef = (float)range;
    const float digf = (float)digitisation;
    const float offf = (float)offset;
    const float unit = rangef / digf;
    for (int64_t j = 0; j < n; j++) {
        const float shifted = (float)raw[j] + offf;
        out[j] = shifted * unit;
    }
}

/* ------------------------------------------------------------------ selection */

#define ORC_DEFINE_SELECT(NAME, T)                                   \
    static T NAME(T *a, int64_t n, int64_t k) {                      \
        int64_t lo = 0, hi = n - 1;                                  \
        while (lo < hi) {                                            \
            const T pivot = a[lo + (hi - lo) / 2];                   \
            int64_t i = lo, j = hi;                                  \
            while (i <= j) {                                         \
                while (a[i] < pivot) i++;                            \
                while (pivot < a[j]) j--;                            \
                if (i <= j) {                                        \
                    const T tmp = a[i];                              \
                    a[i] = a[j];                                     \
                    a[j] = tmp;                                      \
                    i++;                                             \
                    j--;                                             \
                }                                                    \
            }                                                        \
            if (k <= j) hi = j;                                      \
            else if (k >= i) lo = i;                                 \
            else break;                                              \
        }                                                            \
        return a[k];                                                 \
    }

ORC_DEFINE_SELECT(select_f32, float)
ORC_DEFINE_SELECT(select_i16, int16_t)

/* ------------------------------------------------------------------ stat.h */

/* src/stat.h:17-24: one float accumulator, sequential; sum / n with n converted to float */
float orc_meanf(const float *x, int64_t n) {
    float acc = 0.0f;
    for (int64_t i = 0; i < n; i++) acc = acc + x[i];
    return acc / (float)(int)n;
}

/* src/stat.h:26-33 */
float orc_meani16(const int16_t *x, int64_t n) {
    float acc = 0.0f;
    for (int64_t i = 0; i < n; i++) acc = acc + (float)x[i];
    return acc / (float)(int)n;
}

/* src/stat.h:36-44: population std around the sequential-float mean */
float orc_stdvf(const float *x, int64_t n) {
    const float m = orc_meanf(x, n);
    float acc = 0.0f;
    for (int64_t i = 0; i < n; i++) {
        const float d = x[i] - m;
        const float dd = d * d;
        acc = acc + dd;
    }
    return sqrtf(acc / (float)(int)n);
}

/* src/stat.h:46-54 */
float orc_stdvi16(const int16_t *x, int64_t n) {
    const float m = orc_meani16(x, n);
    float acc = 0.0f;
    for (int64_t i = 0; i < n; i++) {
        const float d = (float)x[i] - m;
        const float dd = d * d;
        acc = acc + dd;
    }
    return sqrtf(acc / (float)(int)n);
}

/* src/stat.h:56-64 + src/ksort.h:233-259: element of rank n/2 (0-based, upper median) */
float orc_medianf(const float *x, int64_t n) {
    if (n <= 0) return 0.0f;
    float *tmp = (float *)malloc(sizeof(float) * (size_t)n);
    memcpy(tmp, x, sizeof(float) * (size_t)n);
    const float m = select_f32(tmp, n, n / 2);
    free(tmp);
    return m;
}

/* src/stat.h:66-73 */
int16_t orc_mediani16(const int16_t *x, int64_t n) {
    if (n <= 0) return 0;
    int16_t *tmp = (int16_t *)malloc(sizeof(int16_t) * (size_t)n);
    memcpy(tmp, x, sizeof(int16_t) * (size_t)n);
    const int16_t m = select_i16(tmp, n, n / 2);
    free(tmp);
    return m;
}

/* src/cfunc.c:132-139 */
void orc_stat(const int16_t *raw, int64_t n, double digitisation, double offset, double range,
              float *out5, int32_t *raw_median) {
    out5[0] = orc_meani16(raw, n);
    out5[2] = orc_stdvi16(raw, n);
    *raw_median = orc_mediani16(raw, n);
    float *pa = (float *)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
    orc_pa(raw, n, digitisation, offset, range, pa);
    out5[1] = orc_meanf(pa, n);
    out5[3] = orc_stdvf(pa, n);
    out5[4] = orc_medianf(pa, n);
    free(pa);
}

/* ------------------------------------------------------------------ events.c */

/* src/events.c:293-303.  The square is a float*float product rounded to float,
 * then widened; both running sums are double and strictly sequential. */
void orc_prefix_sums(const float *x, int64_t n, double *sum, double *sumsq) {
    sum[0] = 0.0;
    sumsq[0] = 0.0;
    for (int64_t i = 0; i < n; i++) {
        const float sq = x[i] * x[i];
        sum[i + 1] = sum[i] + (double)x[i];
        sumsq[i + 1] = sumsq[i] + (double)sq;
    }
}

/* src/events.c:315-364.  Expression tree (each line rounds once to the type on its left):
 *   double sum1   = sum[i] - sum[i-w]            (the i>w guard at :341 is a no-op: sum[0]==0)
 *   double sumsq1 = sumsq[i] - sumsq[i-w]
 *   float  sum2   = (float)(sum[i+w] - sum[i])
 *   float  sumsq2 = (float)(sumsq[i+w] - sumsq[i])
 *   float  mean1  = (float)(sum1 / (double)wf)
 *   float  mean2  = sum2 / wf
 *   float  cv     = (float)(((sumsq1/(double)wf - (double)(mean1*mean1)) + (double)(sumsq2/wf)) - (double)(mean2*mean2))
 *   cv = fmaxf(cv, FLT_MIN)
 *   float  delta  = mean2 - mean1
 *   t[i] = (float)(fabs((double)delta) / sqrt((double)(cv / wf)))
 * Indices < w and > n-w stay 0; everything is 0 when n < 2w or w < 2. */
void orc_tstat(const double *sum, const double *sumsq, int64_t n, int w, float *t) {
    for (int64_t i = 0; i < n; i++) t[i] = 0.0f;
    if (n < 2 * (int64_t)w || w < 2) return;
    const float wf = (float)w;
    for (int64_t i = w; i <= n - w; i++) {
        double sum1 = sum[i];
        double sumsq1 = sumsq[i];
        if (i > w) {
            sum1 = sum1 - sum[i - w];
            sumsq1 = sumsq1 - sumsq[i - w];
        }
        const float sum2 = (float)(sum[i + w] - sum[i]);
        const float sumsq2 = (float)(sumsq[i + w] - sumsq[i]);
        const float mean1 = (float)(sum1 / (double)wf);
        const float mean2 = sum2 / wf;
        const float m1sq = mean1 * mean1;
        const float m2sq = mean2 * mean2;
        const float q2 = sumsq2 / wf;
        double acc = sumsq1 / (double)wf;
        acc = acc - (double)m1sq;
        acc = acc + (double)q2;
        acc = acc - (double)m2sq;
        float cv = (float)acc;
        cv = fmaxf(cv, FLT_MIN);
        const float delta = mean2 - mean1;
        const float cvw = cv / wf;
        t[i] = (float)(fabs((double)delta) / sqrt((double)cvw));
    }
}

typedef struct {
    int64_t masked_to; /* size_t in the reference, starts at 0 (so index 0 is always skipped) */
    int64_t peak_pos;  /* -1 = none */
    float peak_value;  /* FLT_MAX when reset by the other detector / at start */
    int valid;
    float threshold;
    int64_t window;
} orc_det_t;

/* src/events.c:371-443.  Two automata stepped short-then-long at every index. */
int64_t orc_peaks(const float *t1, const float *t2, int64_t n, int w1, int w2, float thr1, float thr2,
                  float peak_height, int64_t *peaks) {
    orc_det_t det[2];
    const float *sig[2] = {t1, t2};
    det[0].masked_to = 0; det[0].peak_pos = -1; det[0].peak_value = FLT_MAX; det[0].valid = 0;
    det[0].threshold = thr1; det[0].window = w1;
    det[1] = det[0];
    det[1].threshold = thr2; det[1].window = w2;
    int64_t count = 0;
    for (int64_t i = 0; i < n; i++) {
        for (int k = 0; k < 2; k++) {
            orc_det_t *d = &det[k];
            if (d->masked_to >= i) continue;
            const float v = sig[k][i];
            if (d->peak_pos < 0) {
                if (v < d->peak_value) {
                    d->peak_value = v;
                } else if (v - d->peak_value > peak_height) {
                    d->peak_value = v;
                    d->peak_pos = i;
                }
            } else {
                if (v > d->peak_value) {
                    d->peak_value = v;
                    d->peak_pos = i;
                }
                if (k == 0 && d->peak_value > d->threshold) {
                    det[1].masked_to = d->peak_pos + d->window;
                    det[1].peak_pos = -1;
                    det[1].peak_value = FLT_MAX;
                    det[1].valid = 0;
                }
                if (d->peak_value - v > peak_height && d->peak_value > d->threshold) d->valid = 1;
                if (d->valid && (i - d->peak_pos) > d->window / 2) {
                    peaks[count++] = d->peak_pos;
                    d->peak_pos = -1;
                    d->peak_value = v;
                    d->valid = 0;
                }
            }
        }
    }
    return count;
}

/* src/events.c:457-473 */
static void make_event(int64_t s, int64_t e, const double *sum, const double *sumsq, uint64_t *start,
                       float *length, float *mean, float *stdv) {
    *start = (uint64_t)s;
    const float len = (float)(uint64_t)(e - s);
    *length = len;
    const float m = (float)(sum[e] - sum[s]) / len;
    *mean = m;
    const float dsq = (float)(sumsq[e] - sumsq[s]);
    const float var = dsq / len - m * m;
    *stdv = sqrtf(fmaxf(var, 0.0f));
}

/* The discarded trimming pass (events.c:213-265 via :563): per 100-sample chunk the MAD
 * (two rank-n/2 selections), then a 0-quantile over the chunk MADs.  Only run for timing. */
static float trim_pass_for_timing(const float *x, int64_t n) {
    const int chunk = 100;
    const int64_t nchunk = n / chunk;
    if (nchunk <= 0) return 0.0f;
    float *mad = (float *)malloc(sizeof(float) * (size_t)nchunk);
    float buf[100], dev[100];
    for (int64_t c = 0; c < nchunk; c++) {
        memcpy(buf, x + c * chunk, sizeof buf);
        const float med = select_f32(buf, chunk, chunk / 2);
        for (int i = 0; i < chunk; i++) dev[i] = fabsf(x[c * chunk + i] - med);
        mad[c] = select_f32(dev, chunk, chunk / 2) * 1.4826f;
    }
    float lo = mad[0];
    for (int64_t c = 1; c < nchunk; c++) lo = mad[c] < lo ? mad[c] : lo;
    free(mad);
    return lo;
}

static volatile float orc_sink;

/* src/events.c:506-573 */
int64_t orc_getevents(const float *pa, int64_t n, int rna, int faithful, uint64_t *start, float *length,
                      float *mean, float *stdv, int64_t cap) {
    if (n <= 0) return 0;
    if (faithful) orc_sink = trim_pass_for_timing(pa, n);
    /* presets: src/events.c:43-54 */
    const int w1 = rna ? 7 : 3, w2 = rna ? 14 : 6;
    const float thr1 = rna ? 2.5f : 1.4f, thr2 = 9.0f, ph = rna ? 1.0f : 0.2f;

    double *sum = (double *)malloc(sizeof(double) * (size_t)(n + 1));
    double *sumsq = (double *)malloc(sizeof(double) * (size_t)(n + 1));
    float *t1 = (float *)malloc(sizeof(float) * (size_t)n);
    float *t2 = (float *)malloc(sizeof(float) * (size_t)n);
    int64_t *peaks = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
    orc_prefix_sums(pa, n, sum, sumsq);
    orc_tstat(sum, sumsq, n, w1, t1);
    orc_tstat(sum, sumsq, n, w2, t2);
    const int64_t np = orc_peaks(t1, t2, n, w1, w2, thr1, thr2, ph, peaks);

    /* src/events.c:475-504: event k = [peaks[k-1], peaks[k]), first starts at 0, last ends at n */
    const int64_t nev = np + 1;
    int64_t prev = 0;
    for (int64_t k = 0; k < nev; k++) {
        const int64_t end = (k < np) ? peaks[k] : n;
        if (k < cap) make_event(prev, end, sum, sumsq, &start[k], &length[k], &mean[k], &stdv[k]);
        prev = end;
    }
    free(peaks); free(t2); free(t1); free(sumsq); free(sum);
    return nev;
}

int64_t orc_event_raw(const int16_t *raw, int64_t n, double digitisation, double offset, double range,
                      int rna, int faithful, uint64_t *start, float *length, float *mean, float *stdv,
                      int64_t cap) {
    if (n <= 0) return 0;
    float *pa = (float *)malloc(sizeof(float) * (size_t)n);
    orc_pa(raw, n, digitisation, offset, range, pa);
    const int64_t nev = orc_getevents(pa, n, rna, faithful, start, length, mean, stdv, cap);
    free(pa);
    return nev;
}

int64_t orc_event_batch_count(const int16_t *samples, const uint64_t *offsets, uint32_t n_reads,
                              const double *dig, const double *off, const double *range, int rna,
                              int faithful) {
    int64_t total = 0;
    for (uint32_t r = 0; r < n_reads; r++) {
        const int64_t n = (int64_t)(offsets[r + 1] - offsets[r]);
        const int64_t cap = n / 2 + 2;
        uint64_t *st = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)cap);
        float *ln = (float *)malloc(sizeof(float) * (size_t)cap * 3);
        total += orc_event_raw(samples + offsets[r], n, dig[r], off[r], range[r], rna, faithful, st, ln,
                               ln + cap, ln + 2 * cap, cap);
        free(ln);
        free(st);
    }
    return total;
}

/* ------------------------------------------------------------------ jnn.c */

orc_jnn_param_t orc_jnn_preset(int which) {
    orc_jnn_param_t p;
    p.top = 0.0f;
    p.bot = 0.0f;
    if (which == 1) { /* JNNV1_DRNA_R9_PARAM, src/jnn.h:29-38 */
        p.std_scale = 0.75f; p.corrector = 50; p.seg_dist = 50; p.window = 1000; p.stall_len = 1.0f; p.error = 5;
    } else if (which == 2) { /* JNNV1_R9_POLYA == JNNV1_RNA004_POLYA, src/jnn.h:52-72 */
        p.std_scale = -1.0f; p.corrector = 50; p.seg_dist = 200; p.window = 250; p.stall_len = 1.0f; p.error = 30;
    } else { /* JNNV1_CDNA_R9_PARAM, src/jnn.h:40-49 */
        p.std_scale = 0.75f; p.corrector = 50; p.seg_dist = 50; p.window = 150; p.stall_len = 0.25f; p.error = 5;
    }
    return p;
}

/* src/jnn.c:190-278.  State: in-segment flag, total/consecutive error counts, run counter c,
 * corrector w (never reset, :209/:230), current start.  A segment still open at the end of the
 * read is dropped.  The reference's loop index is int; samples beyond INT_MAX are not a case. */
int64_t orc_jnn_core(const float *sig, int64_t n, orc_jnn_param_t p, int64_t *x, int64_t *y, int64_t cap) {
    float top, bot;
    if (p.std_scale > 0) {
        const float mn = orc_meanf(sig, n);
        const float sd = orc_stdvf(sig, n);
        const float band = sd * p.std_scale;
        top = mn + band;
        bot = mn - band;
    } else {
        top = p.top;
        bot = p.bot;
    }
    int open = 0, err = 0, run_err = 0, c = 0, w = p.corrector;
    int64_t start = 0, nseg = 0, last_y = 0;
    const float first_min = (float)p.window * p.stall_len;
    for (int64_t i = 0; i < n; i++) {
        const float a = sig[i];
        if (a < top && a > bot) {
            if (!open) { start = i; open = 1; }
            c++; w++;
            run_err = 0;
            if (c >= p.window && c >= w && (c % w) == 0) err--;
        } else if (open && err < p.error) {
            c++; err++; run_err++;
            if (c >= p.window && c >= w && (c % w) == 0) err--;
        } else if (open && (c >= p.window || (nseg == 0 && (float)c >= first_min))) {
            const int64_t end = i - run_err;
            open = 0;
            if (nseg > 0 && start - last_y < p.seg_dist) {
                last_y = end;
                if (nseg - 1 < cap) y[nseg - 1] = end;
            } else {
                if (nseg < cap) { x[nseg] = start; y[nseg] = end; }
                last_y = end;
                nseg++;
            }
            c = 0; err = 0; run_err = 0;
        } else if (open) {
            open = 0; c = 0; err = 0; run_err = 0;
        }
    }
    return nseg;
}

/* src/jnn.c:61-77 (rm_outlier) + :282-293 */
int64_t orc_jnn_raw(const int16_t *raw, int64_t n, int rna, int64_t *x, int64_t *y, int64_t cap) {
    if (n <= 0) return 0;
    float *sig = (float *)malloc(sizeof(float) * (size_t)n);
    for (int64_t i = 0; i < n; i++) {
        const int v = raw[i];
        sig[i] = v > 1200 ? 1200.0f : (v < 0 ? 0.0f : (float)v);
    }
    const int64_t k = orc_jnn_core(sig, n, orc_jnn_preset(rna ? 1 : 0), x, y, cap);
    free(sig);
    return k;
}

/* src/jnn.c:79-95 (rm_outlierf) + :295-306 */
int64_t orc_jnn_pa(const float *pa, int64_t n, orc_jnn_param_t p, int64_t *x, int64_t *y, int64_t cap) {
    if (n <= 0) return 0;
    float *sig = (float *)malloc(sizeof(float) * (size_t)n);
    for (int64_t i = 0; i < n; i++) {
        const float v = pa[i];
        sig[i] = v > 1200.0f ? 1200.0f : (v < 0.0f ? 0.0f : v);
    }
    const int64_t k = orc_jnn_core(sig, n, p, x, y, cap);
    free(sig);
    return k;
}

/* src/jnn.c:99-188.  Presets src/jnn.h:84-98: R9 {0.5,1500,2000,hi 200000,lo 2000},
 * RNA004 {0.7,1500,2000,200000,500}. */
void orc_find_adaptor(const int16_t *raw, int64_t n, int pore, int64_t *xy) {
    const float std_scale = (pore == 2) ? 0.7f : 0.5f;
    const int seg_dist = 1500, window = 2000, hi = 200000;
    const int lo = (pore == 2) ? 500 : 2000;
    if (n <= window) { xy[0] = -1; xy[1] = -1; return; }
    const int64_t m = n - window;
    float *t = (float *)malloc(sizeof(float) * (size_t)m);
    /* rolling_window (jnn.c:20-56) over rm_outlier(raw): float running total, exact for
     * integers in [0,1200] with a 2000-wide window (< 2^24) */
    float tot = 0.0f;
#define CLAMPI(v) ((v) > 1200 ? 1200.0f : ((v) < 0 ? 0.0f : (float)(v)))
    for (int i = 0; i < window; i++) tot = tot + CLAMPI(raw[i]);
    t[0] = tot / (float)window;
    for (int64_t i = 1; i < m; i++) {
        tot = tot - CLAMPI(raw[i - 1]);
        tot = tot + CLAMPI(raw[i + window - 1]);
        t[i] = tot / (float)window;
    }
#undef CLAMPI
    const float mn = orc_meanf(t, m);
    const float sd = orc_stdvf(t, m);
    const float bot = mn - sd * std_scale;
    int in_run = 0;
    int64_t start = 0, end = 0;
    /* the reference keeps every run and then picks the first whose length is in [lo,hi]
     * (jnn.c:153-167); merging only ever touches the most recent run, so we can decide a run
     * as soon as the next distinct one is appended, but for clarity keep the whole list. */
    int64_t cap = 1024, ns = 0;
    int64_t *sx = (int64_t *)malloc(sizeof(int64_t) * (size_t)cap);
    int64_t *sy = (int64_t *)malloc(sizeof(int64_t) * (size_t)cap);
    for (int64_t j = 0; j < m; j++) {
        const float v = t[j];
        if (v < bot && !in_run) {
            start = j;
            in_run = 1;
        } else if (v < bot) {
            end = j;
        } else if (v > bot && in_run) {
            if (ns > 0 && start - sy[ns - 1] < seg_dist) {
                sy[ns - 1] = end;
            } else {
                if (ns >= cap) {
                    cap *= 2;
                    sx = (int64_t *)realloc(sx, sizeof(int64_t) * (size_t)cap);
                    sy = (int64_t *)realloc(sy, sizeof(int64_t) * (size_t)cap);
                }
                sx[ns] = start;
                sy[ns] = end;
                ns++;
            }
            start = 0; end = 0; in_run = 0;
        }
    }
    xy[0] = 0; xy[1] = 0;
    for (int64_t i = 0; i < ns; i++) {
        const int64_t len = sy[i] - sx[i];
        if (len > hi || len < lo) continue;
        xy[0] = sx[i] + window / 2 - 1;
        xy[1] = sy[i] + window / 2 - 1;
        break;
    }
    free(sy); free(sx); free(t);
}

/* src/jnn.c:352-374: first segment of jnn_pa with fixed thresholds */
void orc_find_polya(const float *pa, int64_t n, float top, float bot, int pore, int64_t *xy) {
    (void)pore; /* both polyA presets are identical (jnn.h:52-72) */
    orc_jnn_param_t p = orc_jnn_preset(2);
    p.top = top;
    p.bot = bot;
    int64_t x = -1, y = -1;
    const int64_t k = orc_jnn_pa(pa, n, p, &x, &y, 1);
    xy[0] = (k > 0) ? x : -1;
    xy[1] = (k > 0) ? y : -1;
}

/* src/cfunc.c:169-216 */
void orc_prefix(const int16_t *raw, int64_t n, double digitisation, double offset, double range, int rna,
                int pore, orc_prefix_t *out) {
    memset(out, 0, sizeof *out);
    int64_t a[2];
    orc_find_adaptor(raw, n, pore, a);
    out->adapt_x = a[0];
    out->adapt_y = a[1];
    out->polya_x = -1;
    out->polya_y = -1;
    if (a[1] <= 0) return;
    float *pa = (float *)malloc(sizeof(float) * (size_t)n);
    orc_pa(raw, n, digitisation, offset, range, pa);
    const int64_t alen = a[1] - a[0];
    out->adapt_mean = orc_meanf(pa + a[0], alen);
    out->adapt_std = orc_stdvf(pa + a[0], alen);
    out->adapt_median = orc_medianf(pa + a[0], alen);
    if (rna) {
        /* cfunc.c:191: m_a+30+20 and m_a+30-20, evaluated left to right in float */
        const float mid = out->adapt_mean + 30.0f;
        const float top = mid + 20.0f;
        const float bot = mid - 20.0f;
        int64_t pxy[2];
        orc_find_polya(pa + a[1], n - a[1], top, bot, pore, pxy);
        out->polya_x = pxy[0];
        out->polya_y = pxy[1];
        if (pxy[1] > 0) {
            const float *reg = pa + pxy[0] + a[1];
            const int64_t plen = pxy[1] - pxy[0];
            out->polya_mean = orc_meanf(reg, plen);
            out->polya_std = orc_stdvf(reg, plen);
            out->polya_median = orc_medianf(reg, plen);
        }
    }
    free(pa);
}

/* ---- ent: the three entropies of `sigtk ent` (src/ent.c:25-50 entropy(), :107-164 entmain loop) ----
 * entropy(x, len): 65536-bin histogram of the samples reinterpreted as uint16 (:32-35); for every non-empty bin
 * in increasing bin order p = count/len (doubles), ent -= p*log2(p) (:41-46).
 *   out[0] raw:   entropy(raw, n)                                                    (:111)
 *   out[1] delta: zigzag(raw[i] - raw[i-1]) with prev = 0 for i = 0 (:52-61, :120-126), truncated to int16
 *                 (:130-132); entropy over the FIRST n-1 of them (:134 -- the last delta is dropped)
 *   out[2] bytes: high bytes a/256 and low bytes a%256 of those n-1 values as two int16 arrays; the sum of their
 *                 entropies (:143-152)
 * n == 0 makes the reference index with len-1 = 2^64-1 (undefined); here it yields three zeros. */
static double orc_entropy_u16(const uint16_t *x, uint64_t len, int64_t *counts) {
    memset(counts, 0, 65536 * sizeof(int64_t));
    for (uint64_t i = 0; i < len; i++) counts[x[i]]++;
    double ent = 0;
    for (uint64_t i = 0; i < 65536; i++) {
        if (counts[i] > 0) {
            double p = (double)counts[i] / (double)len;
            ent -= p * log2(p);
        }
    }
    return ent;
}

void orc_ent(const int16_t *raw, int64_t n, double *out) {
    out[0] = out[1] = out[2] = 0.0;
    if (n <= 0) return;
    int64_t *counts = (int64_t *)malloc(65536 * sizeof(int64_t));
    uint16_t *a = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)n);
    uint16_t *b = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)n);
    out[0] = orc_entropy_u16((const uint16_t *)raw, (uint64_t)n, counts);
    int32_t prev = 0;
    for (int64_t i = 0; i < n; i++) {
        const int32_t val = (int32_t)raw[i] - prev;
        const uint32_t zz = (uint32_t)((val + val) ^ (val >> 31));
        a[i] = (uint16_t)(int16_t)zz;
        prev = raw[i];
    }
    const uint64_t m = (uint64_t)n - 1;
    out[1] = orc_entropy_u16(a, m, counts);
    for (uint64_t i = 0; i < m; i++) b[i] = (uint16_t)(a[i] / 256);
    double e = orc_entropy_u16(b, m, counts);
    for (uint64_t i = 0; i < m; i++) b[i] = (uint16_t)(a[i] % 256);
    e = e + orc_entropy_u16(b, m, counts);
    out[2] = e;
    free(counts); free(a); free(b);
}
