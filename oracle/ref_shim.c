/* oracle/ref_shim.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Flat (ctypes-friendly) entry points around the REAL reference functions.
 * This file is the repo's own code; it is compiled only into
 * oracle/_ref/libsigtk_ref.so together with the reference objects built from
 * /root/reference (see oracle/Makefile).  Nothing here restates an algorithm:
 * every function forwards to the reference symbol named in its comment.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "sigtk.h"
#include "jnn.h"
#include "stat.h"

static slow5_rec_t shim_rec(const int16_t *raw, uint64_t n, double dig, double off, double range) {
    slow5_rec_t rec;
    memset(&rec, 0, sizeof rec);
    rec.digitisation = dig;
    rec.offset = off;
    rec.range = range;
    rec.sampling_rate = 4000.0;
    rec.len_raw_signal = n;
    rec.raw_signal = (int16_t *)raw;
    return rec;
}

/* signal_in_picoamps (src/misc.c:15) */
void ref_pa(const int16_t *raw, uint64_t n, double dig, double off, double range, float *out) {
    slow5_rec_t rec = shim_rec(raw, n, dig, off, range);
    float *pa = signal_in_picoamps(&rec);
    memcpy(out, pa, sizeof(float) * n);
    free(pa);
}

/* getevents (src/events.c:553) on a caller-supplied pA array.
 * Returns the number of events; fills at most cap entries. */
int64_t ref_getevents(const float *pa, uint64_t n, int rna, uint64_t *start, float *length,
                      float *mean, float *stdv, int64_t cap) {
    float *copy = (float *)malloc(sizeof(float) * n);
    memcpy(copy, pa, sizeof(float) * n);
    event_table et = getevents(n, copy, (int8_t)rna);
    int64_t m = (int64_t)et.n < cap ? (int64_t)et.n : cap;
    for (int64_t i = 0; i < m; i++) {
        start[i] = et.event[i].start;
        length[i] = et.event[i].length;
        mean[i] = et.event[i].mean;
        stdv[i] = et.event[i].stdv;
    }
    int64_t total = (int64_t)et.n;
    free(et.event);
    free(copy);
    return total;
}

/* event_func's compute part (src/cfunc.c:72-78): signal_in_picoamps + getevents. */
int64_t ref_event_raw(const int16_t *raw, uint64_t n, double dig, double off, double range, int rna,
                      uint64_t *start, float *length, float *mean, float *stdv, int64_t cap) {
    slow5_rec_t rec = shim_rec(raw, n, dig, off, range);
    float *pa = signal_in_picoamps(&rec);
    event_table et = getevents(n, pa, (int8_t)rna);
    int64_t m = (int64_t)et.n < cap ? (int64_t)et.n : cap;
    for (int64_t i = 0; i < m; i++) {
        start[i] = et.event[i].start;
        length[i] = et.event[i].length;
        mean[i] = et.event[i].mean;
        stdv[i] = et.event[i].stdv;
    }
    int64_t total = (int64_t)et.n;
    free(et.event);
    free(pa);
    return total;
}

/* Timing helper for bench.py's cpu_baseline ("reference" kind): runs
 * event_func's compute part over a packed batch, returns the total event count. */
int64_t ref_event_batch_count(const int16_t *samples, const uint64_t *offsets, uint32_t n_reads,
                              const double *dig, const double *off, const double *range, int rna) {
    int64_t total = 0;
    for (uint32_t r = 0; r < n_reads; r++) {
        uint64_t n = offsets[r + 1] - offsets[r];
        slow5_rec_t rec = shim_rec(samples + offsets[r], n, dig[r], off[r], range[r]);
        float *pa = signal_in_picoamps(&rec);
        event_table et = getevents(n, pa, (int8_t)rna);
        total += (int64_t)et.n;
        free(et.event);
        free(pa);
    }
    return total;
}

/* stat_func's compute part (src/cfunc.c:132-139) via the stat.h inlines. */
void ref_stat(const int16_t *raw, uint64_t n, double dig, double off, double range, float *out5,
              int32_t *raw_median) {
    float m1 = meani16(raw, n);
    float s1 = stdvi16(raw, n);
    int16_t k1 = mediani16(raw, n);
    slow5_rec_t rec = shim_rec(raw, n, dig, off, range);
    float *cur = signal_in_picoamps(&rec);
    float m2 = meanf(cur, n);
    float s2 = stdvf(cur, n);
    float k2 = medianf(cur, n);
    free(cur);
    out5[0] = m1;
    out5[1] = m2;
    out5[2] = s1;
    out5[3] = s2;
    out5[4] = k2;
    *raw_median = k1;
}

/* meanf / stdvf / medianf (src/stat.h:17,36,56) on a float array. */
void ref_statf(const float *x, int n, float *out3) {
    out3[0] = meanf(x, n);
    out3[1] = stdvf(x, n);
    out3[2] = medianf(x, n);
}

/* jnn_raw (src/jnn.c:282) with the preset jnn_print picks (src/jnn.c:313-319). */
int ref_jnn_raw(const int16_t *raw, int64_t n, int rna, int64_t *x, int64_t *y, int cap) {
    jnn_param_t param;
    if (rna) {
        jnn_param_t tmp = JNNV1_DRNA_R9_PARAM;
        param = tmp;
    } else {
        jnn_param_t tmp = JNNV1_CDNA_R9_PARAM;
        param = tmp;
    }
    int nseg = 0;
    jnn_pair_t *segs = jnn_raw(raw, n, param, &nseg);
    for (int i = 0; i < nseg && i < cap; i++) {
        x[i] = segs[i].x;
        y[i] = segs[i].y;
    }
    free(segs);
    return nseg;
}

/* find_adaptor (src/jnn.c:181) */
void ref_find_adaptor(const int16_t *raw, int64_t n, int pore, int64_t *xy) {
    slow5_rec_t rec = shim_rec(raw, n, 8192.0, 0.0, 1400.0);
    jnn_pair_t p = find_adaptor(&rec, (int8_t)pore);
    xy[0] = p.x;
    xy[1] = p.y;
}

/* find_polya (src/jnn.c:352) */
void ref_find_polya(const float *pa, int64_t n, float top, float bot, int pore, int64_t *xy) {
    jnn_pair_t p = find_polya(pa, n, top, bot, (int8_t)pore);
    xy[0] = p.x;
    xy[1] = p.y;
}
