// shims.hip -- per-read entry points with the reference's own signatures (SURVEY.md 8b "signatures to keep"):
// jnn_raw / jnn_pa / jnnv2 / find_adaptor / find_polya (src/jnn.h:104-109) and the six stat.h inlines
// (src/stat.h:17-73).  Each is a batch of one over the batched kernels (or, for float input, over the single-array
// compatibility kernels of stat_kernels.hip); results are malloc'd by the callee and freed by the caller, as in the
// reference.  They exist so that a maintainer can diff every function against the original; throughput comes from
// the batch API.  On any failure (no device, out of memory, unsupported parameter) they return NULL / {-1,-1} / NaN
// and sgk_shim_status() tells why.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "host_util.h"
#include "sgk_common.h"
#include "stat_args.h"

using namespace sgk;

static thread_local int g_shim_rc = SGK_OK;

namespace {

JnnP to_jnnp(const sgk_jnn_param_t &q) {
    JnnP p;
    p.std_scale = q.std_scale; p.corrector = q.corrector; p.seg_dist = q.seg_dist; p.window = q.window;
    p.stall_len = q.stall_len; p.error = q.error; p.top = q.top; p.bot = q.bot;
    return p;
}

// one int16 read on the device, 64-sample aligned with room around it
int upload_one(const int16_t *raw, int64_t n, DeviceBatch &db) {
    if (n < 0 || n > 0x7fffffffLL) return SGK_ERR_ARG;
    const uint64_t offs[2] = {0, (uint64_t)n};
    const double one = 1.0, zero = 0.0;
    sgk_host_batch_t hb = {raw, offs, &one, &zero, &one, 1};
    return db.upload(&hb);
}

sgk_jnn_pair_t *collect_segments(DevBuf &d_x, DevBuf &d_y, uint32_t ns, int *n) {
    sgk_jnn_pair_t *out = (sgk_jnn_pair_t *)malloc(sizeof(sgk_jnn_pair_t) * (ns ? ns : 1));
    if (!out) { g_shim_rc = SGK_ERR_NOMEM; return nullptr; }
    std::vector<int32_t> x(ns ? ns : 1), y(ns ? ns : 1);
    if (ns) {
        if (hipMemcpy(x.data(), d_x.p, (size_t)ns * 4, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(y.data(), d_y.p, (size_t)ns * 4, hipMemcpyDeviceToHost) != hipSuccess) {
            free(out);
            g_shim_rc = SGK_ERR_HIP;
            return nullptr;
        }
    }
    for (uint32_t k = 0; k < ns; ++k) { out[k].x = x[k]; out[k].y = y[k]; }
    *n = (int)ns;
    return out;
}

int stat_i16(const int16_t *x, int n, sgk_stat_rec_t *rec) {
    if (!x || n <= 0) return SGK_ERR_ARG;
    const uint64_t offs[2] = {0, (uint64_t)n};
    const double one = 1.0, zero = 0.0;
    sgk_host_batch_t hb = {x, offs, &one, &zero, &one, 1};
    return sgk_stat_host(&hb, rec);
}

int stat_f32(const float *x, int n, float *out3) {
    if (!x || n <= 0) return SGK_ERR_ARG;
    if (sgk_device_count() <= 0) return SGK_ERR_NODEVICE;
    DevBuf d_x, d_o;
    int rc;
    if ((rc = d_x.alloc((size_t)n * sizeof(float))) != SGK_OK) return rc;
    if ((rc = d_o.alloc(3 * sizeof(float))) != SGK_OK) return rc;
    SGK_HIP_TRY(hipMemcpy(d_x.p, x, (size_t)n * sizeof(float), hipMemcpyHostToDevice));
    if ((rc = launch_stat_f32(d_x.as<float>(), n, d_o.as<float>(), nullptr)) != SGK_OK) return rc;
    SGK_HIP_TRY(hipDeviceSynchronize());
    SGK_HIP_TRY(hipMemcpy(out3, d_o.p, 3 * sizeof(float), hipMemcpyDeviceToHost));
    return SGK_OK;
}

}  // namespace

// a per-read call on a long read takes the long-read path of the batch API as well (a 3 000 001-sample read: 1 ms instead
// of 10): the workspace it needs, sized for this one read
static int shim_long(StatArgs &a, DevBuf &ws, LongRule auto_div) {
    if (a.b.max_read_len < LC_LONG_MIN / 2) return SGK_OK;   // (the lowest per-batch threshold, stat_args.h: LongRule)
    const size_t bytes = order_workspace_bytes(a.b.n_reads) + long_workspace_bytes(a.b.n_samples, a.b.max_read_len);
    int rc = ws.alloc(bytes);
    if (rc != SGK_OK) return rc;
    return prepare_long(a, ws.p, bytes, 0, auto_div, nullptr);
}

extern "C" {

int sgk_shim_status(void) { return g_shim_rc; }

sgk_jnn_pair_t *sgk_jnn_raw(const int16_t *raw, int64_t nsample, sgk_jnn_param_t param, int *n) {
    g_shim_rc = SGK_OK;
    if (n) *n = 0;
    if (!n || nsample <= 0 || !raw) {  // jnn_raw returns NULL with *n = 0 for an empty read (src/jnn.c:284-291)
        if (!n || (nsample > 0 && !raw)) g_shim_rc = SGK_ERR_ARG;
        return nullptr;
    }
    DeviceBatch db;
    if ((g_shim_rc = upload_one(raw, nsample, db)) != SGK_OK) return nullptr;
    const uint64_t cap = sgk_jnn_slots_for((uint32_t)nsample);
    const uint64_t slots[2] = {0, cap};
    DevBuf d_slots, d_x, d_y, d_n, d_ws;
    if ((g_shim_rc = d_slots.alloc(16)) != SGK_OK || (g_shim_rc = d_x.alloc(cap * 4)) != SGK_OK ||
        (g_shim_rc = d_y.alloc(cap * 4)) != SGK_OK || (g_shim_rc = d_n.alloc(4)) != SGK_OK ||
        (g_shim_rc = d_ws.alloc(64)) != SGK_OK)
        return nullptr;
    if (hipMemcpy(d_slots.p, slots, 16, hipMemcpyHostToDevice) != hipSuccess) { g_shim_rc = SGK_ERR_HIP; return nullptr; }
    StatArgs a;
    memset(&a, 0, sizeof a);
    a.b = db.view;
    a.seg_slots = d_slots.as<uint64_t>();
    a.seg_x = d_x.as<int32_t>();
    a.seg_y = d_y.as<int32_t>();
    a.n_segs = d_n.as<uint32_t>();
    a.err_count = d_ws.as<uint32_t>();
    DevBuf d_long;
    if ((g_shim_rc = shim_long(a, d_long, LC_AUTO_DIV_JNN)) != SGK_OK) return nullptr;
    if ((g_shim_rc = launch_jnn(a, to_jnnp(param), nullptr)) != SGK_OK) return nullptr;
    uint32_t ns = 0, nerr = 0;
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(&ns, d_n.p, 4, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(&nerr, d_ws.p, 4, hipMemcpyDeviceToHost) != hipSuccess) {
        g_shim_rc = SGK_ERR_HIP;
        return nullptr;
    }
    if (nerr) { g_shim_rc = SGK_ERR_CAPACITY; return nullptr; }
    return collect_segments(d_x, d_y, ns, n);
}

sgk_jnn_pair_t *sgk_jnn_pa(const float *raw, int64_t nsample, sgk_jnn_param_t param, int *n) {
    g_shim_rc = SGK_OK;
    if (n) *n = 0;
    if (!n || nsample <= 0 || !raw) {
        if (!n || (nsample > 0 && !raw)) g_shim_rc = SGK_ERR_ARG;
        return nullptr;
    }
    if (nsample > 0x7fffffffLL) { g_shim_rc = SGK_ERR_ARG; return nullptr; }
    if (sgk_device_count() <= 0) { g_shim_rc = SGK_ERR_NODEVICE; return nullptr; }
    const uint32_t cap = (uint32_t)sgk_jnn_slots_for((uint32_t)nsample);
    DevBuf d_in, d_x, d_y, d_n;
    if ((g_shim_rc = d_in.alloc((size_t)nsample * 4)) != SGK_OK || (g_shim_rc = d_x.alloc((size_t)cap * 4)) != SGK_OK ||
        (g_shim_rc = d_y.alloc((size_t)cap * 4)) != SGK_OK || (g_shim_rc = d_n.alloc(8)) != SGK_OK)
        return nullptr;
    if (hipMemcpy(d_in.p, raw, (size_t)nsample * 4, hipMemcpyHostToDevice) != hipSuccess) { g_shim_rc = SGK_ERR_HIP; return nullptr; }
    if ((g_shim_rc = launch_jnn_f32(d_in.as<float>(), nsample, to_jnnp(param), d_x.as<int32_t>(), d_y.as<int32_t>(), cap,
                                    d_n.as<uint32_t>(), nullptr)) != SGK_OK)
        return nullptr;
    uint32_t res[2] = {0, 0};
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(res, d_n.p, 8, hipMemcpyDeviceToHost) != hipSuccess) {
        g_shim_rc = SGK_ERR_HIP;
        return nullptr;
    }
    if (res[1]) { g_shim_rc = SGK_ERR_CAPACITY; return nullptr; }
    return collect_segments(d_x, d_y, res[0], n);
}

sgk_jnn_pair_t sgk_jnnv2(const int16_t *sig, int64_t nsample, sgk_jnnv2_param_t param) {
    sgk_jnn_pair_t p = {-1, -1};
    g_shim_rc = SGK_OK;
    if (param.window != 2000) { g_shim_rc = SGK_ERR_ARG; return p; }  // both presets of the reference (src/jnn.h:84-98)
    if (nsample <= param.window) return p;  // "Not enough data to trim" (src/jnn.c:172-176)
    if (!sig) { g_shim_rc = SGK_ERR_ARG; return p; }
    DeviceBatch db;
    if ((g_shim_rc = upload_one(sig, nsample, db)) != SGK_OK) return p;
    DevBuf d_out;
    if ((g_shim_rc = d_out.alloc(sizeof(sgk_prefix_rec_t))) != SGK_OK) return p;
    StatArgs a;
    memset(&a, 0, sizeof a);
    a.b = db.view;
    a.prefix = d_out.as<sgk_prefix_rec_t>();
    AdaptP ap;
    ap.std_scale = param.std_scale; ap.seg_dist = param.seg_dist; ap.lo_thresh = param.lo_thresh; ap.hi_thresh = param.hi_thresh;
    DevBuf d_long;
    if ((g_shim_rc = shim_long(a, d_long, LC_AUTO_DIV_PREFIX)) != SGK_OK) return p;
    if ((g_shim_rc = launch_adaptor(a, ap, nullptr)) != SGK_OK) return p;
    sgk_prefix_rec_t rec;
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(&rec, d_out.p, sizeof rec, hipMemcpyDeviceToHost) != hipSuccess) {
        g_shim_rc = SGK_ERR_HIP;
        return p;
    }
    p.x = rec.adapt_x;
    p.y = rec.adapt_y;
    return p;
}

sgk_jnn_pair_t sgk_find_adaptor(const int16_t *raw, int64_t nsample, int8_t pore) {
    const AdaptP ap = adaptor_preset(pore);
    sgk_jnnv2_param_t q;
    q.std_scale = ap.std_scale; q.seg_dist = ap.seg_dist; q.window = 2000; q.stall_len = 0.0f;
    q.hi_thresh = ap.hi_thresh; q.lo_thresh = ap.lo_thresh;
    return sgk_jnnv2(raw, nsample, q);
}

sgk_jnn_pair_t sgk_find_polya(const float *raw, int64_t nsample, float top, float bot, int8_t pore) {
    (void)pore;  // JNNV1_R9_POLYA and JNNV1_RNA004_POLYA hold the same values (src/jnn.h:52-72)
    sgk_jnn_pair_t p = {-1, -1};
    const JnnP pp = jnn_polya_preset();
    sgk_jnn_param_t q;
    q.std_scale = pp.std_scale; q.corrector = pp.corrector; q.seg_dist = pp.seg_dist; q.window = pp.window;
    q.stall_len = pp.stall_len; q.error = pp.error; q.top = top; q.bot = bot;
    int ns = 0;
    sgk_jnn_pair_t *segs = sgk_jnn_pa(raw, nsample, q, &ns);
    if (segs) {
        if (ns > 0) p = segs[0];
        free(segs);
    }
    return p;
}

float sgk_meani16(const int16_t *x, int n) {
    sgk_stat_rec_t r;
    return (g_shim_rc = stat_i16(x, n, &r)) == SGK_OK ? r.raw_mean : NAN;
}
float sgk_stdvi16(const int16_t *x, int n) {
    sgk_stat_rec_t r;
    return (g_shim_rc = stat_i16(x, n, &r)) == SGK_OK ? r.raw_std : NAN;
}
int16_t sgk_mediani16(const int16_t *x, int n) {
    sgk_stat_rec_t r;
    return (g_shim_rc = stat_i16(x, n, &r)) == SGK_OK ? (int16_t)r.raw_median : (int16_t)0;
}
float sgk_meanf(const float *x, int n) {
    float o[3];
    return (g_shim_rc = stat_f32(x, n, o)) == SGK_OK ? o[0] : NAN;
}
float sgk_stdvf(const float *x, int n) {
    float o[3];
    return (g_shim_rc = stat_f32(x, n, o)) == SGK_OK ? o[1] : NAN;
}
float sgk_medianf(const float *x, int n) {
    float o[3];
    return (g_shim_rc = stat_f32(x, n, o)) == SGK_OK ? o[2] : NAN;
}

}  // extern "C"
