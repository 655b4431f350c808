// ent_kernels.hip -- the histograms behind `sigtk ent` (src/ent.c:25-50 entropy(), :107-164 per-record loop).
//
// The reference builds three kinds of histograms per read -- the raw samples, the int16-truncated zigzag
// deltas (all but the last one), and the high / low bytes of those -- and sums -p*log2(p) over the bins in
// increasing bin order in double precision with libm's log2.  The counting is the O(n) part and runs here;
// the O(#distinct values) finish (sgk_ent_finish) is host arithmetic on the counts, evaluated with the same
// operations in the same order as the reference, so the printed numbers are identical.
//
// One 256-thread workgroup per read.  Counts of values below the window sizes live in LDS (one bin per
// value); the rare values outside (raw >= 8192 or negative, |delta| >= 2048) are appended to per-read lists
// in global memory, which the finish sorts and counts.  The byte planes always fit (256 bins each).
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "sgk_common.h"
#include "stat_args.h"

namespace sgk {

constexpr int EW_RAW = SGK_ENT_RAW_WINDOW, EW_DELTA = SGK_ENT_DELTA_WINDOW;
constexpr int ENT_WORDS = EW_RAW + EW_DELTA + 512;

__global__ __launch_bounds__(256) void k_ent(sgk_batch_t b, sgk_ent_hist_t *out, uint16_t *over_raw, uint16_t *over_delta) {
    __shared__ uint32_t h[ENT_WORDS];
    __shared__ uint32_t n_over[2];
    const uint32_t r = blockIdx.x;
    const int t = threadIdx.x;
    const int64_t n = (int64_t)b.lengths[r];
    const uint64_t o = b.offsets[r];
    const int16_t *x = b.samples + o;
    for (int i = t; i < ENT_WORDS; i += 256) h[i] = 0;
    if (t < 2) n_over[t] = 0;
    __syncthreads();
    uint32_t *hraw = h, *hdel = h + EW_RAW, *hhi = hdel + EW_DELTA, *hlo = hhi + 256;
    for (int64_t i = t; i < n; i += 256) {
        const int32_t v = (int32_t)x[i];
        const uint32_t rv = (uint32_t)(uint16_t)v;                       // "uint16_t a = raw_signal[i]" (ent.c:33)
        if (rv < (uint32_t)EW_RAW) atomicAdd(&hraw[rv], 1u);
        else over_raw[o + atomicAdd(&n_over[0], 1u)] = (uint16_t)rv;
        if (i < n - 1) {                                                   // the last delta is dropped (ent.c:134)
            const int32_t prev = i ? (int32_t)x[i - 1] : 0;
            const int32_t d = v - prev;
            const uint32_t zz = (uint32_t)((d + d) ^ (d >> 31));         // _zigzag_encode_32 (ent.c:52-54)
            const uint32_t a = zz & 0xffffu;                               // "out[i] = delta[i]" as int16 (ent.c:131)
            if (a < (uint32_t)EW_DELTA) atomicAdd(&hdel[a], 1u);
            else over_delta[o + atomicAdd(&n_over[1], 1u)] = (uint16_t)a;
            atomicAdd(&hhi[a >> 8], 1u);                                   // a/256, a%256 (ent.c:147-148)
            atomicAdd(&hlo[a & 255u], 1u);
        }
    }
    __syncthreads();
    sgk_ent_hist_t *rec = out + r;
    uint32_t *dst = rec->raw;  // raw[], delta[], hi[], lo[] are contiguous in the record
    for (int i = t; i < ENT_WORDS; i += 256) dst[i] = h[i];
    if (t == 0) {
        rec->n = (uint32_t)n;
        rec->n_over_raw = n_over[0];
        rec->n_over_delta = n_over[1];
        rec->reserved = 0;
    }
}

int launch_ent(const sgk_batch_t *b, sgk_ent_hist_t *out, uint16_t *over_raw, uint16_t *over_delta, hipStream_t st) {
    if (b->n_reads == 0) return SGK_OK;
    {
        ProfScope ps("k_ent", st);
        hipLaunchKernelGGL(k_ent, dim3(b->n_reads), dim3(256), 0, st, *b, out, over_raw, over_delta);
    }
    SGK_HIP_TRY(hipGetLastError());
    return SGK_OK;
}

// entropy() of src/ent.c:25-50 on counts: bins in increasing order, p = count/len, ent -= p*log2(p)
static double entropy_counts(const uint32_t *win, int nwin, uint16_t *over, uint32_t n_over, uint64_t len) {
    double ent = 0;
    for (int i = 0; i < nwin; i++) {
        if (win[i] > 0) {
            double p = (double)win[i] / (double)len;
            ent -= p * log2(p);
        }
    }
    if (n_over) {
        std::sort(over, over + n_over);  // all of them are >= nwin: they continue the bin order
        uint32_t i = 0;
        while (i < n_over) {
            uint32_t j = i;
            while (j < n_over && over[j] == over[i]) ++j;
            double p = (double)(j - i) / (double)len;
            ent -= p * log2(p);
            i = j;
        }
    }
    return ent;
}

}  // namespace sgk

using namespace sgk;

extern "C" {

int sgk_ent(const sgk_batch_t *b, sgk_ent_hist_t *out, uint16_t *over_raw, uint16_t *over_delta, void *stream) {
    const int rc = check_batch(b);
    if (rc != SGK_OK) return rc;
    if (b->n_reads == 0) return SGK_OK;
    if (!out || !over_raw || !over_delta) return SGK_ERR_ARG;
    return launch_ent(b, out, over_raw, over_delta, static_cast<hipStream_t>(stream));
}

void sgk_ent_finish(const sgk_ent_hist_t *h, uint16_t *over_raw, uint16_t *over_delta, double *out) {
    out[0] = out[1] = out[2] = 0.0;
    if (!h || h->n == 0) return;  // the reference is undefined for an empty read (len-1 underflows)
    const uint64_t n = h->n, m = n - 1;
    out[0] = entropy_counts(h->raw, SGK_ENT_RAW_WINDOW, over_raw, h->n_over_raw, n);
    out[1] = entropy_counts(h->delta, SGK_ENT_DELTA_WINDOW, over_delta, h->n_over_delta, m);
    double e = entropy_counts(h->hi, 256, nullptr, 0, m);
    e = e + entropy_counts(h->lo, 256, nullptr, 0, m);
    out[2] = e;
}

}  // extern "C"
