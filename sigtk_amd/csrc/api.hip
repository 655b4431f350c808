// api.hip -- C ABI of libsigtk_gpu.so (include/sigtk_gpu.h): error plumbing, per-kernel timing,
// workspace carving, the pa / event / synth entry points and their host-pointer layer.
// (stat / jnn / prefix entry points live in api_stat.hip.)
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "event_args.h"
#include "stat_args.h"
#include "host_util.h"
#include "sgk_common.h"
#include "synth.h"

namespace sgk {

// ---------------------------------------------------------------- errors
static thread_local char g_hip_err[512] = "";

void set_hip_error(hipError_t e, const char *what, const char *file, int line) {
    snprintf(g_hip_err, sizeof g_hip_err, "%s: %s (%s:%d)", hipGetErrorName(e), what, file, line);
}

// ---------------------------------------------------------------- profiling
struct ProfRec {
    const char *name;
    hipEvent_t t0, t1;
    int device;  // the events belong to this device: it is made current again to read / destroy them
};
static std::mutex g_prof_mu;
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof;

ProfScope::ProfScope(const char *n, hipStream_t s) : name(n), stream(s), slot(-1) {
    if (!g_prof_on) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    ProfRec r;
    r.name = n;
    r.device = 0;
    (void)hipGetDevice(&r.device);
    if (hipEventCreate(&r.t0) != hipSuccess) return;
    if (hipEventCreate(&r.t1) != hipSuccess) {
        (void)hipEventDestroy(r.t0);
        return;
    }
    (void)hipEventRecord(r.t0, s);
    g_prof.push_back(r);
    slot = (int)g_prof.size() - 1;
}
ProfScope::~ProfScope() {
    if (slot < 0) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    (void)hipEventRecord(g_prof[slot].t1, stream);
}

// ---------------------------------------------------------------- event options -> configuration
// (no process-wide state: every entry point derives its configuration from the caller's options)
EvSegConfig event_config(const sgk_event_options_t *o) {
    EvSegConfig c;
    c.dev = 0;
    c.seg_len = 131072;    // one wavefront's share of a long read: ~0.6 ms of detector + builder
    c.long_min = 262144;   // reads at least this long are cut into segments
    c.lead_override = 0;
    c.multi = 0;
    c.multi_max = 0;
    c.tail_split = 0;
    c.auto_geometry = !o || (o->segment_len == 0 && o->long_min == 0);
    if (!o) return c;
#ifdef SGK_DEV
    c.dev = o->reserved[0];
#endif
    if (o->segment_len >= 1024 && o->segment_len <= (1u << 30)) c.seg_len = o->segment_len / 1024 * 1024;
    if (o->long_min >= 1) c.long_min = o->long_min;
    if (c.long_min <= c.seg_len) c.long_min = c.seg_len + 1;  // a long read has at least two segments
    if (o->warmup >= 16 && o->warmup <= 512) c.lead_override = o->warmup / 16 * 16;
    if (o->lanes_per_short_read < 0) c.multi = -1;
    else if (o->lanes_per_short_read > 0) {
        int p = 1;
        while (p * 2 <= o->lanes_per_short_read && p < 32) p *= 2;
        c.multi = p;
    }
    if (o->short_max >= 1024) {
        uint32_t p2 = 1024;
        while (p2 < (1u << 30) && (uint64_t)p2 * 2 <= o->short_max) p2 *= 2;
        c.multi_max = p2;
    }
    c.tail_split = o->tail_split < 0 ? -1 : o->tail_split;   // (> 0: that many reads are cut, whatever the batch)
    return c;
}

// Resident wavefront slots of the current device for the event kernels (CUs x 4 SIMDs x waves per SIMD: 3 with the
// DNA preset, 2 with RNA parameters)
static uint32_t event_wave_slots(int rna) {
    static std::mutex mu;
    static int cus[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    std::lock_guard<std::mutex> lk(mu);
    if (cus[dev] == 0) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        cus[dev] = v;
    }
    return (uint32_t)cus[dev] * 4u * (rna ? 2u : 3u);
}

// From what length on a read is shared by several wavefronts, per batch (round 5).  A read should be cut when its one
// wavefront would outlast the rest of the batch -- which depends on the batch: at 10 000 x 100 000 samples (a third of
// a millisecond per 100 000 samples and wave, 3.6 ms per batch) reads up to 262 144 samples finish inside it and cutting
// them only costs (log-normal lengths around 100 000, 10^9 samples: 3.59 ms with 131 072-sample segments from 262 144 on,
// 3.72 with 65 536 from 131 072 on); at 3 000 such reads the same geometry leaves the GPU waiting for a few 200 000-sample
// reads (1.85 ms against 1.49; 3 000 x 100 000 + 16 x 250 000: 1.65 against 1.15).  The batch's samples per wave slot are
// what a wave's share of it is: long_min = 0.9 of that, between 131 072 and 262 144; segments of half of it.
// tests/test_gpu_event_long.py::test_long_read_threshold_is_no_cliff times both ends of the range on both kinds of batch.
EvSegConfig event_config_for(const EvSegConfig &c0, uint64_t n_samples, int rna) {
    EvSegConfig c = c0;
    if (!c.auto_geometry) return c;
    uint64_t lm = n_samples / event_wave_slots(rna) * 9u / 10u;
    lm = (lm + 2047u) / 2048u * 2048u;
    lm = lm < 131072u ? 131072u : (lm > 262144u ? 262144u : lm);
    c.long_min = (uint32_t)lm;
    c.seg_len = (uint32_t)(lm / 2u);
    return c;
}

// The tail split (event_kernels.hip: seg_len_of).  A batch of fewer than 8 rounds of waves whose last round is a SMALL
// fraction of one: the reads of that round (the last dispatch positions) are cut into segments, as many per read as fill
// a round (up to 8), never shorter than 16 384 samples.  Not in a batch with packed short reads (they balance by
// themselves), not for reads under 32 768 samples on average.
// A cut read costs ~1.5 x a whole one (every lane of every segment warms up, the chain waits for the segment in front,
// the builder starts from the last boundary in front of it), so cutting pays only where FEW reads would otherwise keep
// the whole GPU waiting.  The rule is fitted to whole-call times of the shipped kernels with the split forced on / off
// over 50 batch sizes per preset (tools/tail_sweep.py -> profiles/r05_tail_split_sweep.txt, 100 000-sample reads), and
// tests/test_gpu_event_long.py::test_tail_split_rule_is_no_cliff times both either side of every cut of it:
//   a batch of less than one round of waves: DNA preset up to 0.55 of a round (1 536 reads 0.88 -> 0.65 ms; a tie from
//     0.58 on), RNA preset up to 7 / 8 of one (1 024 reads 1.12 -> 0.78 ms, 1 877 reads 1.31 -> 1.25);
//   a longer batch, DNA preset: the last round at most a quarter of a round behind ONE full round (3 840 reads 1.54 ->
//     1.48 ms), a sixth behind more (9 728 reads 3.56 -> 3.42; 9 984 reads 3.54 -> 3.60);
//   a longer batch, RNA preset: never -- at two waves per SIMD the first waves of a round are done at half time and a
//     small surplus runs in their slots (2 218 reads 1.55 ms whole, 1.72 split); the best case gains 6 %, most lose.
// (Round 4's rule -- any last round under 7 / 8 full -- was tuned on a kernel whose whole reads were 6 % slower; the
// first rule of round 5, a sixth / a third, came from a sweep of an instrumented build and left 25 % on the table for
// batches of 1 100 - 1 600 reads: the guard test's first run found that.)
void event_tail_plan(const EvSegConfig &sc, uint32_t n_reads, uint64_t n_samples, uint32_t max_read_len, int rna,
                     bool packed, uint32_t &split_from, uint32_t &split_seg) {
    split_from = n_reads;
    split_seg = 0;
    if (n_reads == 0 || packed || sc.tail_split < 0) return;
    const uint32_t slots = event_wave_slots(rna);
    const uint64_t mean = n_samples / n_reads;
    if (mean < 32768 || (sc.tail_split == 0 && n_reads >= 8ull * slots)) return;
    uint32_t rem = n_reads % slots;
    if (sc.tail_split > 0) rem = (uint32_t)sc.tail_split < n_reads ? (uint32_t)sc.tail_split : n_reads;   // the caller's number
    else if (n_reads < slots) {
        if (rna ? (uint64_t)n_reads * 8u > (uint64_t)slots * 7u : (uint64_t)n_reads * 20u > (uint64_t)slots * 11u) return;
    } else if (rem == 0 || rna || rem * (n_reads < 2u * slots ? 4u : 6u) > slots) return;
    // (the number of segments per read hardly matters: 2 .. 16 per read, one or two rounds of them: 3.77 - 3.89 ms)
    uint32_t G = (slots - slots / 16 + rem - 1) / rem;   // units of the split reads ~ one round
    if (G < 2) G = 2;
    if (G > 8) G = 8;
    uint64_t seg = (mean + G - 1) / G;
    seg = (seg + 1023) / 1024 * 1024;
    if (seg < 16384) seg = 16384;
    if (seg >= sc.long_min) return;   // (reads that long are cut anyway)
    split_from = n_reads - rem;
    split_seg = (uint32_t)seg;
}

void event_seg_capacity(const EvSegConfig &c, uint32_t n_reads, uint64_t n_samples, uint32_t max_read_len, int rna,
                        bool packed, uint32_t &max_segs, uint32_t &max_long) {
    max_segs = 0;
    max_long = 0;
    uint64_t ns = 0, ml = 0;
    if (max_read_len >= c.long_min) {
        uint64_t nl = n_samples / c.long_min;
        if (nl > n_reads) nl = n_reads;
        if (nl < 1) nl = 1;
        ns = n_samples / c.seg_len + nl;  // sum of ceil(n_r / seg_len) over at most nl reads
        // every long read has at least two segments; k_seg_plan admits long reads while their segments fit
        ml = ns / 2;
        if (ml < nl) ml = nl;
    }
    uint32_t sf = n_reads, ss = 0;
    event_tail_plan(c, n_reads, n_samples, max_read_len, rna, packed, sf, ss);
    if (ss) {
        // the split reads: at most n_reads - sf of them, each under long_min samples
        const uint64_t nsplit = n_reads - sf;
        ns += nsplit * (((uint64_t)c.long_min + ss - 1) / ss);
        ml += nsplit;
    }
    if (ml > n_reads) ml = n_reads;
    max_segs = ns > 0x7fffffffull ? 0x7fffffffu : (uint32_t)ns;
    max_long = (uint32_t)ml;
}

// whether a batch with these totals gets packed short reads (k_event_multi)
static bool event_batch_packed(const EvSegConfig &c, uint32_t n_reads, uint64_t n_samples, uint32_t max_read_len, int rna) {
    uint32_t lanes = 0, mmax = 0;
    const bool sorted = n_reads >= ORDER_MIN_READS && (uint64_t)max_read_len * n_reads > n_samples + n_samples / 4;
    event_multi_plan(c, n_reads, n_samples, max_read_len, rna, sorted, lanes, mmax);
    return lanes != 0;
}

EvWorkspace event_workspace_layout(const EvSegConfig &c, uint32_t n_reads, uint64_t n_samples, uint32_t max_read_len,
                                   size_t available) {
    EvWorkspace w;
    const uint64_t nr = n_reads ? n_reads : 1;
    size_t o = 0;
    w.off_hdr = o;     o += sizeof(EvHeader);
    w.off_flags = o;   o += round_up(nr, 64);
    w.off_list = o;    o += round_up(nr * 4, 64);
    w.off_order = o;   o += round_up(nr * 4 + 128 * 4, 64);
    w.off_bitmap = o;  o += round_up((n_samples / 64 + nr + 2) * 8, 64);
    // (the layout does not know the preset: the larger of the two presets' capacities)
    {
        uint32_t s0 = 0, l0 = 0, s1 = 0, l1 = 0;
        const EvSegConfig c0 = event_config_for(c, n_samples, 0), c1 = event_config_for(c, n_samples, 1);
        event_seg_capacity(c0, n_reads, n_samples, max_read_len, 0, event_batch_packed(c0, n_reads, n_samples, max_read_len, 0), s0, l0);
        event_seg_capacity(c1, n_reads, n_samples, max_read_len, 1, event_batch_packed(c1, n_reads, n_samples, max_read_len, 1), s1, l1);
        w.max_segs = s0 > s1 ? s0 : s1;
        w.max_long = l0 > l1 ? l0 : l1;
    }
    w.off_segs = o;       o += round_up((size_t)w.max_segs * sizeof(SegDesc), 64);
    w.off_seg_state = o;  o += round_up((size_t)w.max_segs * sizeof(SegState), 64);
    w.off_longs = o;      o += round_up((size_t)w.max_long * sizeof(LongRead), 64);
    w.off_scratch = o;
    w.scratch_stride = round_up(2ull * ((uint64_t)max_read_len + 1), 2);
    const size_t per_block = (size_t)w.scratch_stride * sizeof(double);
    uint64_t nb = nr < 256 ? nr : 256;  // persistent fallback blocks (each owns a scratch region)
    // default sizing: keep the scratch under 2 GiB even for multi-million-sample reads (fewer, still >= 1, blocks)
    const uint64_t budget = (2ull << 30) / per_block;
    if (nb > budget) nb = budget ? budget : 1;
    if (available) {
        const size_t room = available > o ? available - o : 0;
        uint64_t fit = room / per_block;
        if (fit > 256) fit = 256;
        if (fit > nr) fit = nr;
        nb = fit;
    }
    w.n_fb_blocks = (uint32_t)nb;
    w.total = o + (size_t)nb * per_block;
    return w;
}

int launch_pa(const sgk_batch_t *b, float *out, hipStream_t st);
int launch_synth(int16_t *samples, const uint64_t *offsets, const uint32_t *lengths, double *dig, double *off,
                 double *rng, uint32_t n_reads, uint32_t max_read_len, uint64_t first_read, uint64_t seed, int kind,
                 hipStream_t st);

int check_batch(const sgk_batch_t *b) {
    if (!b) return SGK_ERR_ARG;
    if (b->n_reads == 0) return SGK_OK;
    if (!b->samples || !b->offsets || !b->lengths || !b->digitisation || !b->offset || !b->range) return SGK_ERR_ARG;
    if (reinterpret_cast<uintptr_t>(b->samples) & 15u) return SGK_ERR_ALIGN;
    if (b->n_samples & 7u) return SGK_ERR_ARG;
    return SGK_OK;
}

// Short reads.  On all 64 lanes a 5 000-sample read spends more steps on warm-ups (lead per lane) than on its samples;
// k_event_multi gives a read fewer lanes and a wave several reads.  Worth it when the batch has enough reads to fill
// the GPU that way (>= 4 rounds of waves) -- a small batch wants every lane it can get.  `sorted`: the batch gets a
// dispatch order (its short reads are the order's tail); without one the batch must be short as a whole.
void event_multi_plan(const EvSegConfig &sc, uint32_t n_reads, uint64_t n_samples, uint32_t max_read_len, int rna,
                      bool sorted, uint32_t &multi_lanes, uint32_t &multi_max) {
    multi_lanes = 0;
    // (RNA parameters: warm-ups of 128 / 256 samples make the 64-lane layout 1.3 x its samples' worth up to ~64 k
    // samples: 33 333 x 30 000 samples 6.15 -> 5.13 ms; DNA parameters: 32 / 64, 20 000-sample reads already cost what
    // 100 000-sample reads cost.  Above that the packed kernel's per-lane read parameters cost more than they save:
    // two 100 000-sample reads per wave are 6-9 % slower than one.)
    multi_max = sc.multi_max ? sc.multi_max : (rna ? 65536u : 16384u);
    if (n_reads == 0) return;
    const uint64_t mean = n_samples / n_reads;
    if (sc.multi >= 0 && mean < multi_max && (sorted || max_read_len < multi_max)) {
        const uint32_t lead = sc.lead_override > 0 ? (uint32_t)sc.lead_override : (rna ? 128u : 32u);
        uint32_t lanes = 1;
        while (lanes < 64 && (uint64_t)lanes * 2 * 8 * lead <= mean) lanes *= 2;       // chunks of >= 8 warm-ups
        while (lanes < 64 && (uint64_t)n_reads * lanes < 64ull * 4 * 3072) lanes *= 2;  // >= 4 rounds of waves
        if (sc.multi > 0) lanes = (uint32_t)sc.multi;
        if (lanes < 64) multi_lanes = lanes;
    }
}

static int run_event(const void *samples, bool float_input, const uint64_t *offsets, const uint32_t *lengths,
                     const double *dig, const double *off, const double *rng, uint32_t n_reads,
                     uint32_t max_read_len, uint64_t n_samples, int rna, const uint64_t *ev_slots,
                     sgk_event_rec_t *events, uint32_t *n_events, void *ws, size_t ws_bytes, void *stream,
                     const sgk_event_options_t *opt) {
    if (n_reads == 0) return SGK_OK;
    if (!samples || !offsets || !lengths || !ev_slots || !events || !n_events || !ws) return SGK_ERR_ARG;
    if (reinterpret_cast<uintptr_t>(samples) & 15u) return SGK_ERR_ALIGN;
    if (reinterpret_cast<uintptr_t>(events) & 15u) return SGK_ERR_ALIGN;
    if (reinterpret_cast<uintptr_t>(ws) & 63u) return SGK_ERR_ALIGN;
    const EvSegConfig sc0 = event_config(opt), sc = event_config_for(sc0, n_samples, rna);
    const EvWorkspace w = event_workspace_layout(sc0, n_reads, n_samples, max_read_len, ws_bytes);
    if (w.n_fb_blocks == 0 || w.total > ws_bytes) return SGK_ERR_WORKSPACE;
    char *base = static_cast<char *>(ws);
    EvArgs a;
    a.samples = samples;
    a.offsets = offsets;
    a.lengths = lengths;
    a.dig = dig;
    a.off = off;
    a.rng = rng;
    a.n_reads = n_reads;
    a.n_alloc = n_samples;
    a.ev_slots = ev_slots;
    a.events = events;
    a.n_events = n_events;
    a.hdr = reinterpret_cast<EvHeader *>(base + w.off_hdr);
    a.flags = reinterpret_cast<uint8_t *>(base + w.off_flags);
    a.flag_list = reinterpret_cast<uint32_t *>(base + w.off_list);
    // longest-first dispatch only pays when lengths differ: a batch of equal-length reads is launched in batch order
    a.order = ((uint64_t)max_read_len * n_reads > n_samples + n_samples / 4) ? reinterpret_cast<uint32_t *>(base + w.off_order) : nullptr;
    a.bitmap = reinterpret_cast<unsigned long long *>(base + w.off_bitmap);
    a.scratch = reinterpret_cast<double *>(base + w.off_scratch);
    a.scratch_stride = w.scratch_stride;
    a.max_segs = w.max_segs;
    a.max_long = w.max_long;
    a.seg_len = sc.seg_len;
    a.long_min = sc.long_min;
    a.lead_override = sc.lead_override;
    a.dev = sc.dev;
    a.segs = reinterpret_cast<SegDesc *>(base + w.off_segs);
    a.seg_state = reinterpret_cast<SegState *>(base + w.off_seg_state);
    a.longs = reinterpret_cast<LongRead *>(base + w.off_longs);
    const bool sorted = a.order != nullptr && n_reads >= ORDER_MIN_READS;
    event_multi_plan(sc, n_reads, n_samples, max_read_len, rna, sorted, a.multi_lanes, a.multi_max);
    event_tail_plan(sc, n_reads, n_samples, max_read_len, rna, a.multi_lanes != 0, a.split_from, a.split_seg);
    a.has_long = max_read_len >= sc.long_min ? 1u : 0u;
    return launch_event(a, rna, float_input, w.n_fb_blocks, static_cast<hipStream_t>(stream));
}

}  // namespace sgk

using namespace sgk;

// ====================================================================== C ABI

extern "C" {

const char *sgk_strerror(int code) {
    switch (code) {
        case SGK_OK: return "ok";
        case SGK_ERR_ARG: return "invalid argument";
        case SGK_ERR_HIP: return "HIP runtime error";
        case SGK_ERR_NODEVICE: return "no usable GPU";
        case SGK_ERR_WORKSPACE: return "workspace too small";
        case SGK_ERR_CAPACITY: return "output arena slot range too small";
        case SGK_ERR_ALIGN: return "buffer not sufficiently aligned";
        case SGK_ERR_NOMEM: return "host allocation failed";
        case SGK_ERR_FORMAT: return "malformed compressed signal";
        default: return "unknown sgk error";
    }
}

const char *sgk_version(void) { return SGK_VERSION_STRING; }
const char *sgk_last_hip_error(void) { return g_hip_err; }

int sgk_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int sgk_set_device(int ordinal) {
    if (sgk_device_count() <= 0) return SGK_ERR_NODEVICE;
    SGK_HIP_TRY(hipSetDevice(ordinal));
    return SGK_OK;
}

void sgk_profile_enable(int on) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_on = on != 0;
}

void sgk_profile_reset(void) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (auto &r : g_prof) {
        (void)hipSetDevice(r.device);
        (void)hipEventSynchronize(r.t1);
        (void)hipEventDestroy(r.t0);
        (void)hipEventDestroy(r.t1);
    }
    (void)hipSetDevice(cur);
    g_prof.clear();
}

int sgk_profile_read(const char **names, double *ms, uint32_t *calls, int cap) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    int k = 0;
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (auto &r : g_prof) {
        (void)hipSetDevice(r.device);  // records of several devices (sigtk-amd --gpus N) share the list
        if (hipEventSynchronize(r.t1) != hipSuccess) continue;
        float t = 0.f;
        if (hipEventElapsedTime(&t, r.t0, r.t1) != hipSuccess) continue;
        int j = 0;
        for (; j < k; ++j)
            if (strcmp(names[j], r.name) == 0) break;
        if (j == k) {
            if (k >= cap) continue;
            names[k] = r.name;
            ms[k] = 0.0;
            calls[k] = 0;
            ++k;
        }
        ms[j] += (double)t;
        calls[j] += 1;
    }
    (void)hipSetDevice(cur);
    return k;
}

// ---------------------------------------------------------------- pa
int sgk_pa(const sgk_batch_t *batch, float *pa_out, void *stream) {
    const int rc = check_batch(batch);
    if (rc != SGK_OK) return rc;
    if (batch->n_reads == 0) return SGK_OK;
    if (!pa_out) return SGK_ERR_ARG;
    return launch_pa(batch, pa_out, static_cast<hipStream_t>(stream));
}

// ---------------------------------------------------------------- event
size_t sgk_event_workspace_bytes_opt(uint32_t n_reads, uint64_t n_samples, uint32_t max_read_len,
                                     const sgk_event_options_t *opt) {
    return event_workspace_layout(event_config(opt), n_reads, n_samples, max_read_len, 0).total;
}
size_t sgk_event_workspace_bytes(uint32_t n_reads, uint64_t n_samples, uint32_t max_read_len) {
    return sgk_event_workspace_bytes_opt(n_reads, n_samples, max_read_len, nullptr);
}
#ifdef SGK_DEV
// development builds only: where in the workspace the fallback scratch -- SGK_DEV_TRACE's records -- begins
size_t sgk_debug_scratch_offset(uint32_t n_reads, uint64_t n_samples, uint32_t max_read_len, const sgk_event_options_t *opt,
                                size_t ws_bytes) {
    return event_workspace_layout(event_config(opt), n_reads, n_samples, max_read_len, ws_bytes).off_scratch;
}
#endif

int sgk_event_opt(const sgk_batch_t *b, int rna, const uint64_t *ev_slots, sgk_event_rec_t *events, uint32_t *n_events,
                  void *ws, size_t ws_bytes, void *stream, const sgk_event_options_t *opt) {
    const int rc = check_batch(b);
    if (rc != SGK_OK) return rc;
    return run_event(b->samples, false, b->offsets, b->lengths, b->digitisation, b->offset, b->range, b->n_reads,
                     b->max_read_len, b->n_samples, rna, ev_slots, events, n_events, ws, ws_bytes, stream, opt);
}
int sgk_event(const sgk_batch_t *b, int rna, const uint64_t *ev_slots, sgk_event_rec_t *events, uint32_t *n_events,
              void *ws, size_t ws_bytes, void *stream) {
    return sgk_event_opt(b, rna, ev_slots, events, n_events, ws, ws_bytes, stream, nullptr);
}

int sgk_event_pa_opt(const float *pa, const uint64_t *offsets, const uint32_t *lengths, uint32_t n_reads,
                     uint32_t max_read_len, uint64_t n_samples, int rna, const uint64_t *ev_slots,
                     sgk_event_rec_t *events, uint32_t *n_events, void *ws, size_t ws_bytes, void *stream,
                     const sgk_event_options_t *opt) {
    return run_event(pa, true, offsets, lengths, nullptr, nullptr, nullptr, n_reads, max_read_len, n_samples, rna,
                     ev_slots, events, n_events, ws, ws_bytes, stream, opt);
}
int sgk_event_pa(const float *pa, const uint64_t *offsets, const uint32_t *lengths, uint32_t n_reads,
                 uint32_t max_read_len, uint64_t n_samples, int rna, const uint64_t *ev_slots,
                 sgk_event_rec_t *events, uint32_t *n_events, void *ws, size_t ws_bytes, void *stream) {
    return sgk_event_pa_opt(pa, offsets, lengths, n_reads, max_read_len, n_samples, rna, ev_slots, events, n_events, ws,
                            ws_bytes, stream, nullptr);
}

// pa -> event -> stat over one resident batch (BASELINE config 5): the fused stat + pA pass (per-read statistics and the pA
// array: two reads of the samples, one write), then event on the raw samples (it scales on the fly).
// (Round 5 also had the event BUILDER write the pA -- it converts every sample on its walk anyway -- with stat on the lane
// path: bit-identical, and not faster: 62.4 ms against 63.1 at 125 000 x 100 000; k_event went from 39 to 52 ms, 15.8 GB
// of traffic per 1e9 samples make it HBM-bound where it is issue-bound without.  profiles/r05_event_experiments.md.)
int sgk_pipeline(const sgk_batch_t *b, int rna, const uint64_t *ev_slots, sgk_event_rec_t *events, uint32_t *n_events,
                 float *pa_out, sgk_stat_rec_t *stat_out, void *event_ws, size_t event_ws_bytes, void *stat_ws,
                 size_t stat_ws_bytes, void *stream, const sgk_event_options_t *event_opt, const sgk_stat_options_t *stat_opt) {
    int rc = check_batch(b);
    if (rc != SGK_OK) return rc;
    if (!pa_out || !stat_out) return SGK_ERR_ARG;
    rc = sgk_stat_pa_opt(b, stat_out, pa_out, stat_ws, stat_ws_bytes, stream, stat_opt);
    if (rc != SGK_OK) return rc;
    return run_event(b->samples, false, b->offsets, b->lengths, b->digitisation, b->offset, b->range, b->n_reads, b->max_read_len,
                     b->n_samples, rna, ev_slots, events, n_events, event_ws, event_ws_bytes, stream, event_opt);
}

int sgk_event_status(const void *ws, sgk_event_status_t *out, void *stream) {
    if (!ws || !out) return SGK_ERR_ARG;
    EvHeader h;
    hipStream_t st = static_cast<hipStream_t>(stream);
    SGK_HIP_TRY(hipMemcpyAsync(&h, ws, sizeof h, hipMemcpyDeviceToHost, st));
    SGK_HIP_TRY(hipStreamSynchronize(st));
    out->n_fallback_reads = h.n_flagged;
    out->n_rerun_passes = h.n_rerun;
    out->n_capacity_overflow = h.n_overflow;
    out->n_long_replays = h.n_hot_runs;
    out->n_events_total = h.n_events_total;
    out->n_split_reads = h.n_long;   // (reads k_seg_plan could not place count here and under n_fallback_reads)
    out->n_segments = h.n_segs;
    out->n_seam_reruns = h.n_seam_rerun;
    out->reserved = 0;
    out->n_replay_indices = h.n_replay_idx;
    return h.n_overflow ? SGK_ERR_CAPACITY : SGK_OK;
}

int sgk_event_plan_opt(uint32_t n_reads, uint64_t n_samples, uint32_t max_read_len, int rna, const sgk_event_options_t *opt,
                       sgk_event_plan_t *out) {
    if (!out) return SGK_ERR_ARG;
    const sgk::EvSegConfig sc = sgk::event_config_for(sgk::event_config(opt), n_samples, rna);
    uint32_t max_segs = 0, max_long = 0, lanes = 0, mmax = 0, sf = n_reads, ss = 0;
    const bool sorted = n_reads >= sgk::ORDER_MIN_READS && (uint64_t)max_read_len * n_reads > n_samples + n_samples / 4;
    sgk::event_multi_plan(sc, n_reads, n_samples, max_read_len, rna, sorted, lanes, mmax);
    sgk::event_seg_capacity(sc, n_reads, n_samples, max_read_len, rna, lanes != 0, max_segs, max_long);
    sgk::event_tail_plan(sc, n_reads, n_samples, max_read_len, rna, lanes != 0, sf, ss);
    memset(out, 0, sizeof *out);
    out->segment_len = sc.seg_len;
    out->long_min = sc.long_min;
    out->max_segments = max_segs;
    out->max_long_reads = max_long;
    out->short_max = mmax;
    out->lanes_per_short_read = lanes;
    out->warmup_override = (uint32_t)sc.lead_override;
    out->tail_split_from = sf;
    out->tail_segment_len = ss;
    return SGK_OK;
}
// the 0.1.0 form under its 0.1.0 name (0.2.0 / 0.2.1 had put the six-argument form under this symbol: a caller compiled
// against 0.1.0 then passed its `out` in the options' slot): the defaults, and only the 32 bytes that struct had
int sgk_event_plan(uint32_t n_reads, uint64_t n_samples, uint32_t max_read_len, int rna, void *out32) {
    if (!out32) return SGK_ERR_ARG;
    sgk_event_plan_t p;
    const int rc = sgk_event_plan_opt(n_reads, n_samples, max_read_len, rna, nullptr, &p);
    if (rc == SGK_OK) memcpy(out32, &p, 32);
    return rc;
}

// ---------------------------------------------------------------- synthetic reads
int sgk_synth_reads(int16_t *samples, const uint64_t *offsets, const uint32_t *lengths, double *dig, double *off,
                    double *rng, uint32_t n_reads, uint32_t max_read_len, uint64_t first_read, uint64_t seed,
                    int kind, void *stream) {
    if (n_reads == 0) return SGK_OK;
    if (!samples || !offsets || !lengths || !dig || !off || !rng) return SGK_ERR_ARG;
    return launch_synth(samples, offsets, lengths, dig, off, rng, n_reads, max_read_len, first_read, seed, kind,
                        static_cast<hipStream_t>(stream));
}

void sgk_synth_reads_host(int16_t *samples, const uint64_t *offsets, const uint32_t *lengths, double *dig,
                          double *off, double *rng, uint32_t n_reads, uint64_t first_read, uint64_t seed, int kind) {
    for (uint32_t r = 0; r < n_reads; ++r) {
        const int64_t n = (int64_t)lengths[r];
        const sgk_synth_read_t R = sgk_synth_read_init(seed, first_read + r, n, kind);
        dig[r] = SGK_SYNTH_DIGITISATION;
        off[r] = (double)R.offset;
        rng[r] = SGK_SYNTH_RANGE;
        for (int64_t i = 0; i < n; ++i) samples[offsets[r] + i] = sgk_synth_sample(R, i);
    }
}

// ====================================================================== Host API

int sgk_pa_host(const sgk_host_batch_t *hb, float *pa_out) {
    DeviceBatch db;
    int rc = db.upload(hb);
    if (rc != SGK_OK) return rc;
    if (hb->n_reads == 0) return SGK_OK;
    if (!pa_out) return SGK_ERR_ARG;
    DevBuf d_out;
    if ((rc = d_out.alloc((size_t)db.n_samples * sizeof(float))) != SGK_OK) return rc;
    if ((rc = sgk_pa(&db.view, d_out.as<float>(), nullptr)) != SGK_OK) return rc;
    SGK_HIP_TRY(hipDeviceSynchronize());
    for (uint32_t r = 0; r < hb->n_reads; ++r) {
        if (!db.lengths[r]) continue;
        SGK_HIP_TRY(hipMemcpy(pa_out + hb->offsets[r], d_out.as<float>() + db.offsets[r],
                              (size_t)db.lengths[r] * sizeof(float), hipMemcpyDeviceToHost));
    }
    return SGK_OK;
}

// shared tail of sgk_event_host / sgk_getevents: run on an uploaded batch view (raw or pA input)
static int event_collect(const void *d_samples, bool float_input, const DeviceBatch &db, const sgk_batch_t *view,
                         int rna, sgk_events_host_t *out, const sgk_event_options_t *opt = nullptr) {
    const uint32_t nr = (uint32_t)db.lengths.size();
    memset(out, 0, sizeof *out);
    out->n_reads = nr;
    out->ev_offsets = (uint64_t *)calloc((size_t)nr + 1, sizeof(uint64_t));
    if (!out->ev_offsets) return SGK_ERR_NOMEM;
    if (nr == 0) return SGK_OK;
    const std::vector<uint64_t> slots = make_slots(db.lengths, [](uint32_t n) { return sgk_event_slots_for(n); });
    const uint64_t nslots = slots[nr];
    DevBuf d_slots, d_ev, d_nev, d_ws;
    int rc;
    if ((rc = d_slots.alloc((nr + 1) * sizeof(uint64_t))) != SGK_OK) return rc;
    if ((rc = d_ev.alloc(nslots * sizeof(sgk_event_rec_t))) != SGK_OK) return rc;
    if ((rc = d_nev.alloc((size_t)nr * 4)) != SGK_OK) return rc;
    const size_t wsb = sgk_event_workspace_bytes_opt(nr, db.n_samples, db.max_len, opt);
    if ((rc = d_ws.alloc(wsb)) != SGK_OK) return rc;
    SGK_HIP_TRY(hipMemcpy(d_slots.p, slots.data(), (nr + 1) * sizeof(uint64_t), hipMemcpyHostToDevice));
    if (float_input)
        rc = sgk_event_pa_opt(static_cast<const float *>(d_samples), view->offsets, view->lengths, nr, db.max_len,
                              db.n_samples, rna, d_slots.as<uint64_t>(), d_ev.as<sgk_event_rec_t>(), d_nev.as<uint32_t>(),
                              d_ws.p, wsb, nullptr, opt);
    else
        rc = sgk_event_opt(view, rna, d_slots.as<uint64_t>(), d_ev.as<sgk_event_rec_t>(), d_nev.as<uint32_t>(), d_ws.p, wsb,
                           nullptr, opt);
    if (rc != SGK_OK) return rc;
    rc = sgk_event_status(d_ws.p, &out->status, nullptr);
    if (rc != SGK_OK) return rc;
    std::vector<uint32_t> nev(nr);
    SGK_HIP_TRY(hipMemcpy(nev.data(), d_nev.p, (size_t)nr * 4, hipMemcpyDeviceToHost));
    uint64_t tot = 0;
    for (uint32_t r = 0; r < nr; ++r) {
        out->ev_offsets[r] = tot;
        tot += nev[r];
    }
    out->ev_offsets[nr] = tot;
    out->start = (uint32_t *)malloc((tot ? tot : 1) * 4);
    out->length = (uint32_t *)malloc((tot ? tot : 1) * 4);
    out->mean = (float *)malloc((tot ? tot : 1) * 4);
    out->stdv = (float *)malloc((tot ? tot : 1) * 4);
    if (!out->start || !out->length || !out->mean || !out->stdv) return SGK_ERR_NOMEM;
    // one bulk copy of the records (capacity layout), compacted and split into the four arrays on the host
    std::vector<sgk_event_rec_t> tmp((size_t)nslots ? (size_t)nslots : 1);
    SGK_HIP_TRY(hipMemcpy(tmp.data(), d_ev.p, (size_t)nslots * sizeof(sgk_event_rec_t), hipMemcpyDeviceToHost));
    for (uint32_t r = 0; r < nr; ++r) {
        // an overflowing read keeps what fitted
        const uint64_t cap = slots[r + 1] - slots[r];
        const uint64_t k = nev[r] < cap ? nev[r] : cap;
        const sgk_event_rec_t *src = tmp.data() + slots[r];
        const uint64_t o = out->ev_offsets[r];
        for (uint64_t i = 0; i < k; ++i) {
            out->start[o + i] = src[i].start;
            out->length[o + i] = src[i].length;
            out->mean[o + i] = src[i].mean;
            out->stdv[o + i] = src[i].stdv;
        }
    }
    return SGK_OK;
}

int sgk_event_host_opt(const sgk_host_batch_t *hb, int rna, sgk_events_host_t *out, const sgk_event_options_t *opt) {
    if (!out) return SGK_ERR_ARG;
    memset(out, 0, sizeof *out);
    DeviceBatch db;
    int rc = db.upload(hb);
    if (rc != SGK_OK) return rc;
    rc = event_collect(db.view.samples, false, db, &db.view, rna, out, opt);
    if (rc != SGK_OK) sgk_events_host_free(out);
    return rc;
}
int sgk_event_host(const sgk_host_batch_t *hb, int rna, sgk_events_host_t *out) {
    return sgk_event_host_opt(hb, rna, out, nullptr);
}

void sgk_events_host_free(sgk_events_host_t *ev) {
    if (!ev) return;
    free(ev->ev_offsets);
    free(ev->start);
    free(ev->length);
    free(ev->mean);
    free(ev->stdv);
    memset(ev, 0, sizeof *ev);
}

// ---------------------------------------------------------------- per-read shims
float *sgk_signal_in_picoamps(const int16_t *raw, uint64_t n, double digitisation, double offset, double range) {
    float *out = (float *)malloc(sizeof(float) * (n ? n : 1));
    if (!out) return nullptr;
    const uint64_t offs[2] = {0, n};
    sgk_host_batch_t hb = {raw, offs, &digitisation, &offset, &range, 1};
    if (sgk_pa_host(&hb, out) != SGK_OK) {
        free(out);
        return nullptr;
    }
    return out;
}

sgk_event_table sgk_getevents(size_t nsample, float *rawptr, int8_t rna) {
    sgk_event_table et;
    memset(&et, 0, sizeof et);
    if (!rawptr || nsample == 0 || nsample > 0x7fffffffull || sgk_device_count() <= 0) return et;
    DeviceBatch db;  // only offsets/lengths bookkeeping is used here
    db.offsets.assign(1, 256);  // head/tail room for the event fast path
    db.lengths.assign(1, (uint32_t)nsample);
    db.max_len = (uint32_t)nsample;
    db.n_samples = round_up(nsample, 64) + 320;
    DevBuf d_pa, d_o, d_l;
    if (d_pa.alloc((size_t)db.n_samples * sizeof(float)) != SGK_OK) return et;
    if (d_o.alloc(8) != SGK_OK || d_l.alloc(4) != SGK_OK) return et;
    if (hipMemset(d_pa.p, 0, (size_t)db.n_samples * sizeof(float)) != hipSuccess) return et;
    if (hipMemcpy(d_pa.as<float>() + 256, rawptr, nsample * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
        return et;
    if (hipMemcpy(d_o.p, db.offsets.data(), 8, hipMemcpyHostToDevice) != hipSuccess) return et;
    if (hipMemcpy(d_l.p, db.lengths.data(), 4, hipMemcpyHostToDevice) != hipSuccess) return et;
    sgk_batch_t view;
    memset(&view, 0, sizeof view);
    view.offsets = d_o.as<uint64_t>();
    view.lengths = d_l.as<uint32_t>();
    sgk_events_host_t ev;
    if (event_collect(d_pa.p, true, db, &view, rna, &ev) != SGK_OK) {
        sgk_events_host_free(&ev);
        return et;
    }
    const size_t k = (size_t)ev.ev_offsets[1];
    et.event = (sgk_event_t *)calloc(k ? k : 1, sizeof(sgk_event_t));
    if (et.event) {
        et.n = k;
        et.start = 0;
        et.end = k;
        for (size_t i = 0; i < k; ++i) {
            et.event[i].start = ev.start[i];
            et.event[i].length = (float)ev.length[i];
            et.event[i].mean = ev.mean[i];
            et.event[i].stdv = ev.stdv[i];
        }
    }
    sgk_events_host_free(&ev);
    return et;
}

}  // extern "C"
