// api_stat.hip -- C ABI entry points of stat / stat_pa / jnn / prefix (device and host layers).
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "host_util.h"
#include "sgk_common.h"
#include "stat_args.h"

using namespace sgk;

static StatArgs make_args(const sgk_batch_t *b) {
    StatArgs a;
    memset(&a, 0, sizeof a);
    a.b = *b;
    return a;
}

extern "C" {

// 64 bytes of counters + room for the longest-first dispatch order (a smaller workspace, down to 64 bytes -- none for
// stat / prefix -- is accepted: the kernels then take the reads in batch order)
// ... and the records of the batch's long reads (k_long_chains; without that room long reads run on one wavefront)
static size_t stat_ws(uint32_t n_reads, uint64_t n_samples, uint32_t max_len) {
    return order_workspace_bytes(n_reads) + long_workspace_bytes(n_samples, max_len);
}
size_t sgk_stat_workspace_bytes(uint32_t n_reads, uint64_t n_samples, uint32_t max_len) { return stat_ws(n_reads, n_samples, max_len); }
size_t sgk_jnn_workspace_bytes(uint32_t n_reads, uint64_t n_samples, uint32_t max_len) { return stat_ws(n_reads, n_samples, max_len); }
size_t sgk_prefix_workspace_bytes(uint32_t n_reads, uint64_t n_samples, uint32_t max_len) { return stat_ws(n_reads, n_samples, max_len); }

int sgk_stat_opt(const sgk_batch_t *b, sgk_stat_rec_t *out, void *ws, size_t ws_bytes, void *stream,
                 const sgk_stat_options_t *opt) {
    int rc = check_batch(b);
    if (rc != SGK_OK) return rc;
    if (b->n_reads == 0) return SGK_OK;
    if (!out) return SGK_ERR_ARG;
    StatArgs a = make_args(b);
    a.kernels = opt ? opt->kernels : 0;
    a.long_fault = opt ? opt->debug_fault : 0u;
    a.stat = out;
    if ((rc = prepare_order(a, ws, ws_bytes, static_cast<hipStream_t>(stream))) != SGK_OK) return rc;
    if ((rc = prepare_long(a, ws, ws_bytes, opt ? opt->long_min : 0, LC_AUTO_DIV_STAT, static_cast<hipStream_t>(stream))) != SGK_OK) return rc;
    return launch_stat(a, static_cast<hipStream_t>(stream));
}

int sgk_stat(const sgk_batch_t *b, sgk_stat_rec_t *out, void *ws, size_t ws_bytes, void *stream) {
    return sgk_stat_opt(b, out, ws, ws_bytes, stream, nullptr);
}

int sgk_stat_pa_opt(const sgk_batch_t *b, sgk_stat_rec_t *out, float *pa_out, void *ws, size_t ws_bytes, void *stream,
                    const sgk_stat_options_t *opt) {
    int rc = check_batch(b);
    if (rc != SGK_OK) return rc;
    if (b->n_reads == 0) return SGK_OK;
    if (!out || !pa_out) return SGK_ERR_ARG;
    if (reinterpret_cast<uintptr_t>(pa_out) & 15u) return SGK_ERR_ALIGN;
    StatArgs a = make_args(b);
    a.kernels = opt ? opt->kernels : 0;
    a.long_fault = opt ? opt->debug_fault : 0u;
    a.stat = out;
    a.pa_out = pa_out;  // written by the first pass of k_stat_wave (lane-per-read kernels: by the median pass)
    if ((rc = prepare_order(a, ws, ws_bytes, static_cast<hipStream_t>(stream))) != SGK_OK) return rc;
    if ((rc = prepare_long(a, ws, ws_bytes, opt ? opt->long_min : 0, LC_AUTO_DIV_STAT, static_cast<hipStream_t>(stream))) != SGK_OK) return rc;
    return launch_stat(a, static_cast<hipStream_t>(stream));
}

int sgk_stat_pa(const sgk_batch_t *b, sgk_stat_rec_t *out, float *pa_out, void *ws, size_t ws_bytes, void *stream) {
    return sgk_stat_pa_opt(b, out, pa_out, ws, ws_bytes, stream, nullptr);
}

int sgk_jnn_opt(const sgk_batch_t *b, int rna, const uint64_t *seg_slots, int32_t *seg_x, int32_t *seg_y,
                uint32_t *n_segs, void *ws, size_t ws_bytes, void *stream, const sgk_stat_options_t *opt) {
    int rc = check_batch(b);
    if (rc != SGK_OK) return rc;
    if (b->n_reads == 0) return SGK_OK;
    if (!seg_slots || !seg_x || !seg_y || !n_segs || !ws) return SGK_ERR_ARG;
    if (ws_bytes < 64) return SGK_ERR_WORKSPACE;
    StatArgs a = make_args(b);
    a.kernels = opt ? opt->kernels : 0;
    a.long_fault = opt ? opt->debug_fault : 0u;
    a.seg_slots = seg_slots;
    a.seg_x = seg_x;
    a.seg_y = seg_y;
    a.n_segs = n_segs;
    a.err_count = static_cast<uint32_t *>(ws);
    if ((rc = prepare_order(a, ws, ws_bytes, static_cast<hipStream_t>(stream))) != SGK_OK) return rc;
    if ((rc = prepare_long(a, ws, ws_bytes, opt ? opt->long_min : 0, LC_AUTO_DIV_JNN, static_cast<hipStream_t>(stream))) != SGK_OK) return rc;
    return launch_jnn(a, jnn_preset(rna), static_cast<hipStream_t>(stream));
}

int sgk_jnn(const sgk_batch_t *b, int rna, const uint64_t *seg_slots, int32_t *seg_x, int32_t *seg_y,
            uint32_t *n_segs, void *ws, size_t ws_bytes, void *stream) {
    return sgk_jnn_opt(b, rna, seg_slots, seg_x, seg_y, n_segs, ws, ws_bytes, stream, nullptr);
}

int sgk_prefix_opt(const sgk_batch_t *b, int rna, int pore, sgk_prefix_rec_t *out, void *ws, size_t ws_bytes,
                   void *stream, const sgk_stat_options_t *opt) {
    int rc = check_batch(b);
    if (rc != SGK_OK) return rc;
    if (b->n_reads == 0) return SGK_OK;
    if (!out) return SGK_ERR_ARG;
    StatArgs a = make_args(b);
    a.kernels = opt ? opt->kernels : 0;
    a.long_fault = opt ? opt->debug_fault : 0u;
    a.prefix = out;
    if ((rc = prepare_order(a, ws, ws_bytes, static_cast<hipStream_t>(stream))) != SGK_OK) return rc;
    if ((rc = prepare_long(a, ws, ws_bytes, opt ? opt->long_min : 0, LC_AUTO_DIV_PREFIX, static_cast<hipStream_t>(stream))) != SGK_OK) return rc;
    return launch_prefix(a, rna, pore, static_cast<hipStream_t>(stream));
}

int sgk_prefix(const sgk_batch_t *b, int rna, int pore, sgk_prefix_rec_t *out, void *ws, size_t ws_bytes,
               void *stream) {
    return sgk_prefix_opt(b, rna, pore, out, ws, ws_bytes, stream, nullptr);
}

int sgk_stat_plan(int tool, uint32_t n_reads, uint64_t n_samples, uint32_t max_read_len, const sgk_stat_options_t *opt,
                  sgk_stat_plan_t *out) {
    if (!out || tool < 0 || tool > 3) return SGK_ERR_ARG;
    memset(out, 0, sizeof *out);
    const int kernels = opt ? opt->kernels : 0;
    const int32_t lm_opt = opt ? opt->long_min : 0;
    out->kernels = stat_lane_per_read(tool, kernels, n_reads, n_samples, max_read_len) ? 1u : 2u;
    out->workspace_bytes = stat_ws(n_reads, n_samples, max_read_len);
    const uint32_t lm = long_threshold(n_samples, lm_opt, tool == 1 ? LC_AUTO_DIV_JNN : (tool == 2 ? LC_AUTO_DIV_PREFIX : LC_AUTO_DIV_STAT));
    // (the long-read path belongs to the wave-per-read kernels and needs a read that long in the batch)
    if (out->kernels == 2u && lm_opt >= 0 && max_read_len >= lm) {
        out->long_min = lm;
        out->long_max_reads = lm_opt == 0 ? LC_AUTO_MAX_READS : LC_CAP;
    }
    return SGK_OK;
}

int sgk_stat_lane_rules(sgk_stat_lane_rule_t *out, int cap) {
    for (int k = 0; k < N_LANE_RULES && k < cap && out; ++k) {
        out[k].tool = (uint32_t)LANE_RULES[k].tool;
        out[k].min_reads = LANE_RULES[k].min_reads;
        out[k].slope_x1024 = LANE_RULES[k].slope_x1024;
        out[k].intercept = LANE_RULES[k].intercept;
        out[k].cap = LANE_RULES[k].cap;
        out[k].reserved = 0;
    }
    return N_LANE_RULES;
}

int sgk_stat_long_status(const void *ws, size_t ws_bytes, uint32_t n_reads, sgk_long_status_t *out) {
    if (!out) return SGK_ERR_ARG;
    memset(out, 0, sizeof *out);
    const size_t off = order_workspace_bytes(n_reads);
    if (!ws || ws_bytes < off + long_workspace_bytes(0, 0)) return SGK_OK;
    LongHdr h;
    SGK_HIP_TRY(hipMemcpy(&h, static_cast<const char *>(ws) + off, sizeof h, hipMemcpyDeviceToHost));
    out->n_long_reads = h.n_long;
    out->n_tiles = h.n_tiles;
    out->n_true_tiles = h.n_true;
    out->n_timeouts = h.n_declined;  // (h.n_timeout: the waits given up -- none when a fault was only injected)
    return SGK_OK;
}

// ---------------------------------------------------------------- host layer

int sgk_stat_host_opt(const sgk_host_batch_t *hb, sgk_stat_rec_t *out, const sgk_stat_options_t *opt) {
    DeviceBatch db;
    int rc = db.upload(hb);
    if (rc != SGK_OK) return rc;
    const size_t nr = hb->n_reads;
    if (nr == 0) return SGK_OK;
    if (!out) return SGK_ERR_ARG;
    DevBuf d_out, d_ws;
    if ((rc = d_out.alloc(nr * sizeof(sgk_stat_rec_t))) != SGK_OK) return rc;
    if ((rc = d_ws.alloc(stat_ws(hb->n_reads, db.view.n_samples, db.view.max_read_len))) != SGK_OK) return rc;
    if ((rc = sgk_stat_opt(&db.view, d_out.as<sgk_stat_rec_t>(), d_ws.p, stat_ws(hb->n_reads, db.view.n_samples, db.view.max_read_len), nullptr, opt)) != SGK_OK) return rc;
    SGK_HIP_TRY(hipDeviceSynchronize());
    SGK_HIP_TRY(hipMemcpy(out, d_out.p, nr * sizeof(sgk_stat_rec_t), hipMemcpyDeviceToHost));
    return SGK_OK;
}

int sgk_stat_host(const sgk_host_batch_t *hb, sgk_stat_rec_t *out) { return sgk_stat_host_opt(hb, out, nullptr); }

int sgk_prefix_host_opt(const sgk_host_batch_t *hb, int rna, int pore, sgk_prefix_rec_t *out,
                        const sgk_stat_options_t *opt) {
    DeviceBatch db;
    int rc = db.upload(hb);
    if (rc != SGK_OK) return rc;
    const size_t nr = hb->n_reads;
    if (nr == 0) return SGK_OK;
    if (!out) return SGK_ERR_ARG;
    DevBuf d_out, d_ws;
    if ((rc = d_out.alloc(nr * sizeof(sgk_prefix_rec_t))) != SGK_OK) return rc;
    if ((rc = d_ws.alloc(stat_ws(hb->n_reads, db.view.n_samples, db.view.max_read_len))) != SGK_OK) return rc;
    if ((rc = sgk_prefix_opt(&db.view, rna, pore, d_out.as<sgk_prefix_rec_t>(), d_ws.p, stat_ws(hb->n_reads, db.view.n_samples, db.view.max_read_len), nullptr,
                             opt)) != SGK_OK)
        return rc;
    SGK_HIP_TRY(hipDeviceSynchronize());
    SGK_HIP_TRY(hipMemcpy(out, d_out.p, nr * sizeof(sgk_prefix_rec_t), hipMemcpyDeviceToHost));
    return SGK_OK;
}

int sgk_prefix_host(const sgk_host_batch_t *hb, int rna, int pore, sgk_prefix_rec_t *out) {
    return sgk_prefix_host_opt(hb, rna, pore, out, nullptr);
}

int sgk_jnn_host_opt(const sgk_host_batch_t *hb, int rna, sgk_segs_host_t *out, const sgk_stat_options_t *opt) {
    if (!out) return SGK_ERR_ARG;
    memset(out, 0, sizeof *out);
    DeviceBatch db;
    int rc = db.upload(hb);
    if (rc != SGK_OK) return rc;
    const uint32_t nr = hb->n_reads;
    out->n_reads = nr;
    out->seg_offsets = (uint64_t *)calloc((size_t)nr + 1, sizeof(uint64_t));
    if (!out->seg_offsets) return SGK_ERR_NOMEM;
    if (nr == 0) return SGK_OK;
    const std::vector<uint64_t> slots = make_slots(db.lengths, [](uint32_t n) { return sgk_jnn_slots_for(n); });
    const uint64_t nslots = slots[nr];
    DevBuf d_slots, d_x, d_y, d_n, d_ws;
    if ((rc = d_slots.alloc((nr + 1) * sizeof(uint64_t))) != SGK_OK) return rc;
    if ((rc = d_x.alloc(nslots * 4)) != SGK_OK) return rc;
    if ((rc = d_y.alloc(nslots * 4)) != SGK_OK) return rc;
    if ((rc = d_n.alloc((size_t)nr * 4)) != SGK_OK) return rc;
    if ((rc = d_ws.alloc(stat_ws(nr, db.view.n_samples, db.view.max_read_len))) != SGK_OK) return rc;
    SGK_HIP_TRY(hipMemcpy(d_slots.p, slots.data(), (nr + 1) * sizeof(uint64_t), hipMemcpyHostToDevice));
    rc = sgk_jnn_opt(&db.view, rna, d_slots.as<uint64_t>(), d_x.as<int32_t>(), d_y.as<int32_t>(), d_n.as<uint32_t>(),
                     d_ws.p, stat_ws(nr, db.view.n_samples, db.view.max_read_len), nullptr, opt);
    if (rc != SGK_OK) return rc;
    SGK_HIP_TRY(hipDeviceSynchronize());
    uint32_t nerr = 0;
    SGK_HIP_TRY(hipMemcpy(&nerr, d_ws.p, 4, hipMemcpyDeviceToHost));
    if (nerr) return SGK_ERR_CAPACITY;
    std::vector<uint32_t> ns(nr);
    SGK_HIP_TRY(hipMemcpy(ns.data(), d_n.p, (size_t)nr * 4, hipMemcpyDeviceToHost));
    uint64_t tot = 0;
    for (uint32_t r = 0; r < nr; ++r) {
        out->seg_offsets[r] = tot;
        tot += ns[r];
    }
    out->seg_offsets[nr] = tot;
    out->x = (int32_t *)malloc((tot ? tot : 1) * 4);
    out->y = (int32_t *)malloc((tot ? tot : 1) * 4);
    if (!out->x || !out->y) return SGK_ERR_NOMEM;
    // one bulk copy per array (capacity layout), compacted on the host
    std::vector<int32_t> tmp(nslots ? (size_t)nslots : 1);
    for (int a = 0; a < 2; ++a) {
        SGK_HIP_TRY(hipMemcpy(tmp.data(), a ? d_y.p : d_x.p, (size_t)nslots * 4, hipMemcpyDeviceToHost));
        int32_t *o = a ? out->y : out->x;
        for (uint32_t r = 0; r < nr; ++r)
            if (ns[r]) memcpy(o + out->seg_offsets[r], tmp.data() + slots[r], (size_t)ns[r] * 4);
    }
    return SGK_OK;
}

int sgk_jnn_host(const sgk_host_batch_t *hb, int rna, sgk_segs_host_t *out) { return sgk_jnn_host_opt(hb, rna, out, nullptr); }

void sgk_segs_host_free(sgk_segs_host_t *s) {
    if (!s) return;
    free(s->seg_offsets);
    free(s->x);
    free(s->y);
    memset(s, 0, sizeof *s);
}

}  // extern "C"
