// job.hip -- pipelined host jobs (sgk_job_*): the host-side runtime the CLI drives.
//
// A job owns everything one batch needs on its way through a GPU: pinned host staging for the
// signal (raw int16 or svb-zd blobs exactly as they sit in the BLOW5 records), the device
// buffers, a private HIP stream, and pinned host buffers for the results.  All buffers only ever
// grow, so a job that is recycled batch after batch stops allocating after the first few batches.
//
//   sgk_job_begin   lays the batch out (64-sample aligned reads, head/tail room for the event fast
//                   path) and hands the caller the pinned staging pointers; reader threads
//                   inflate/copy records straight into them
//   sgk_job_submit  enqueues H2D -> (svb-zd decode) -> subtool kernels -> D2H on the job's stream
//                   and returns immediately
//   sgk_job_wait    synchronises the stream and validates decode/event status
//
// Several jobs in flight (on one or several GPUs) overlap reading/inflating, PCIe transfers,
// kernels and output formatting; the reference does all of this strictly one record at a time
// (src/cmain.c:118-120).
#include <stdlib.h>
#include <string.h>

#include <new>

#include "sgk_common.h"
#include "stat_args.h"

namespace sgk {

struct GrowDev {
    void *p = nullptr;
    size_t cap = 0;
    ~GrowDev() {
        if (p) (void)hipFree(p);
    }
    int ensure(size_t bytes) {
        if (bytes <= cap) return SGK_OK;
        if (p) SGK_HIP_TRY(hipFree(p));
        p = nullptr;
        cap = 0;
        const size_t want = round_up(bytes + bytes / 8, 4096);  // some slack: batches vary a little
        SGK_HIP_TRY(hipMalloc(&p, want));
        cap = want;
        return SGK_OK;
    }
    template <typename T>
    T *as() const {
        return static_cast<T *>(p);
    }
};

struct GrowPin {
    void *p = nullptr;
    size_t cap = 0;
    ~GrowPin() {
        if (p) (void)hipHostFree(p);
    }
    int ensure(size_t bytes) {
        if (bytes <= cap) return SGK_OK;
        if (p) SGK_HIP_TRY(hipHostFree(p));
        p = nullptr;
        cap = 0;
        const size_t want = round_up(bytes + bytes / 8, 4096);
        SGK_HIP_TRY(hipHostMalloc(&p, want, hipHostMallocDefault));
        cap = want;
        return SGK_OK;
    }
    template <typename T>
    T *as() const {
        return static_cast<T *>(p);
    }
};

}  // namespace sgk

using namespace sgk;

struct sgk_job {
    int device = 0;
    hipStream_t st = nullptr;
    // batch geometry
    uint32_t n_reads = 0, max_len = 0;
    uint64_t n_samples = 0, blob_bytes = 0;
    int fmt = SGK_SIGNAL_INT16;
    // input staging (pinned) and device mirrors
    GrowPin h_samples, h_blobs, h_offsets, h_lengths, h_boffs, h_blens, h_dig, h_off, h_rng, h_slots;
    GrowDev d_samples, d_blobs, d_offsets, d_lengths, d_boffs, d_blens, d_dig, d_off, d_rng, d_slots, d_ws, d_dstat;
    // outputs: four generic arrays (start/length/mean/stdv | seg x/y | pa | records) + per-read counts
    GrowDev d_out[4], d_cnt;
    GrowDev d_dense[4], d_doffs;      // event / jnn: items gathered to dense per-read ranges before the download
    GrowPin h_out[4], h_cnt, h_dstat, h_doffs, h_err, h_long;
    // SGK_SIGNAL_ZREC: the records as they sit in the file (h_blobs / d_blobs), inflated on the device into d_inflated;
    // per record: where it goes there, how much room it has, where its signal blob starts (all relative to d_inflated)
    GrowPin h_ioffs, h_icaps, h_istat, h_soffs, h_slens;
    GrowDev d_inflated, d_ioffs, d_icaps, d_ilens, d_istat, d_soffs, d_slens;
    uint64_t inflated_bytes = 0;
    bool long_fetched = false;        // stat / jnn / prefix: the long-read header of the call is on its way to h_long
    uint32_t long_declined = 0;       // ... long reads the long path declined (n_timeouts of sgk_long_status_t), after wait
    int n_dense = 0;                  // arrays to fetch in sgk_job_wait once the dense total is known
    size_t ws_bytes = 0;
    int tool = -1, flags = 0;
    bool begun = false, submitted = false, ent_over = false;
    sgk_event_status_t ev_status;
    sgk_event_options_t ev_opt;   // sgk_job_set_options (all zero: the defaults)
    sgk_stat_options_t st_opt;
};

// stat / jnn / prefix: the long-read path's header (64 bytes behind the dispatch order in the workspace) rides home with
// the results, so that sgk_job_wait can tell how many long reads were declined and redone (n_timeouts of
// sgk_long_status_t; the results are right either way)
static int fetch_long_hdr(sgk_job *j, hipStream_t st) {
    j->long_fetched = false;
    const size_t off = order_workspace_bytes(j->n_reads);
    if (!j->d_ws.p || j->d_ws.cap < off + long_workspace_bytes(0, 0)) return SGK_OK;
    int rc;
    if ((rc = j->h_long.ensure(sizeof(LongHdr))) != SGK_OK) return rc;
    SGK_HIP_TRY(hipMemcpyAsync(j->h_long.p, static_cast<const char *>(j->d_ws.p) + off, sizeof(LongHdr), hipMemcpyDeviceToHost, st));
    j->long_fetched = true;
    return SGK_OK;
}

extern "C" {

int sgk_job_create(int device, sgk_job_t **out) {
    if (!out) return SGK_ERR_ARG;
    *out = nullptr;
    const int nd = sgk_device_count();
    if (nd <= 0) return SGK_ERR_NODEVICE;
    if (device < 0 || device >= nd) return SGK_ERR_ARG;
    SGK_HIP_TRY(hipSetDevice(device));
    sgk_job *j = new (std::nothrow) sgk_job();
    if (!j) return SGK_ERR_NOMEM;
    j->device = device;
    memset(&j->ev_status, 0, sizeof j->ev_status);
    memset(&j->ev_opt, 0, sizeof j->ev_opt);
    memset(&j->st_opt, 0, sizeof j->st_opt);
    const hipError_t e = hipStreamCreateWithFlags(&j->st, hipStreamNonBlocking);
    if (e != hipSuccess) {
        set_hip_error(e, "hipStreamCreateWithFlags", __FILE__, __LINE__);
        delete j;
        return SGK_ERR_HIP;
    }
    *out = j;
    return SGK_OK;
}

int sgk_job_set_options(sgk_job_t *j, const sgk_event_options_t *event_opt, const sgk_stat_options_t *stat_opt) {
    if (!j) return SGK_ERR_ARG;
    if (event_opt) j->ev_opt = *event_opt;
    else memset(&j->ev_opt, 0, sizeof j->ev_opt);
    if (stat_opt) j->st_opt = *stat_opt;
    else memset(&j->st_opt, 0, sizeof j->st_opt);
    return SGK_OK;
}

void sgk_job_destroy(sgk_job_t *j) {
    if (!j) return;
    (void)hipSetDevice(j->device);
    if (j->st) {
        (void)hipStreamSynchronize(j->st);
        (void)hipStreamDestroy(j->st);
    }
    delete j;
}

int sgk_job_device(const sgk_job_t *j) { return j ? j->device : -1; }

static int job_begin(sgk_job_t *j, uint32_t n_reads, const uint32_t *lengths, int signal_format, const uint32_t *blob_bytes,
                     const uint32_t *sig_offset, const uint32_t *sig_bytes, const uint32_t *rec_room, sgk_job_input_t *in);

int sgk_job_begin(sgk_job_t *j, uint32_t n_reads, const uint32_t *lengths, int signal_format,
                  const uint32_t *blob_bytes, sgk_job_input_t *in) {
    if (signal_format != SGK_SIGNAL_INT16 && signal_format != SGK_SIGNAL_SVBZD) return SGK_ERR_ARG;
    return job_begin(j, n_reads, lengths, signal_format, blob_bytes, nullptr, nullptr, nullptr, in);
}

int sgk_job_begin_zrec(sgk_job_t *j, uint32_t n_reads, const uint32_t *lengths, const uint32_t *rec_bytes,
                       const uint32_t *sig_offset, const uint32_t *sig_bytes, const uint32_t *rec_room,
                       sgk_job_input_t *in) {
    if (n_reads && (!rec_bytes || !sig_offset || !sig_bytes || !rec_room)) return SGK_ERR_ARG;
    return job_begin(j, n_reads, lengths, SGK_SIGNAL_ZREC, rec_bytes, sig_offset, sig_bytes, rec_room, in);
}

static int job_begin(sgk_job_t *j, uint32_t n_reads, const uint32_t *lengths, int signal_format, const uint32_t *blob_bytes,
                     const uint32_t *sig_offset, const uint32_t *sig_bytes, const uint32_t *rec_room, sgk_job_input_t *in) {
    if (!j || !in || (n_reads && !lengths)) return SGK_ERR_ARG;
    if (signal_format != SGK_SIGNAL_INT16 && n_reads && !blob_bytes) return SGK_ERR_ARG;
    SGK_HIP_TRY(hipSetDevice(j->device));
    if (j->submitted) return SGK_ERR_ARG;  // previous batch still in flight: sgk_job_wait first
    const size_t nr = n_reads, nr1 = nr ? nr : 1;
    int rc;
    if ((rc = j->h_offsets.ensure(nr1 * 8)) != SGK_OK) return rc;
    if ((rc = j->h_lengths.ensure(nr1 * 4)) != SGK_OK) return rc;
    if ((rc = j->h_dig.ensure(nr1 * 8)) != SGK_OK) return rc;
    if ((rc = j->h_off.ensure(nr1 * 8)) != SGK_OK) return rc;
    if ((rc = j->h_rng.ensure(nr1 * 8)) != SGK_OK) return rc;
    uint64_t *offs = j->h_offsets.as<uint64_t>();
    uint32_t *lens = j->h_lengths.as<uint32_t>();
    // every read starts on a 64-sample (128-byte) boundary; 256 samples of head room and 64 of tail
    // room so that the event fast path never has to decline a read (sigtk_gpu.h, "layout")
    uint64_t o = 256;
    uint32_t mx = 0;
    for (size_t r = 0; r < nr; ++r) {
        if (lengths[r] > 0x7fffffffu) return SGK_ERR_ARG;  // nsample is int32 in the reference (misc.c:20)
        offs[r] = o;
        lens[r] = lengths[r];
        if (lengths[r] > mx) mx = lengths[r];
        o += round_up(lengths[r], 64);
    }
    j->n_reads = n_reads;
    j->max_len = mx;
    j->n_samples = o + 64;
    j->fmt = signal_format;
    j->blob_bytes = 0;
    memset(in, 0, sizeof *in);
    if (signal_format == SGK_SIGNAL_SVBZD || signal_format == SGK_SIGNAL_ZREC) {
        if ((rc = j->h_boffs.ensure(nr1 * 8)) != SGK_OK) return rc;
        if ((rc = j->h_blens.ensure(nr1 * 4)) != SGK_OK) return rc;
        uint64_t *bo = j->h_boffs.as<uint64_t>();
        uint32_t *bl = j->h_blens.as<uint32_t>();
        uint64_t b = 0;
        for (size_t r = 0; r < nr; ++r) {
            bo[r] = b;
            bl[r] = blob_bytes[r];
            b += round_up(blob_bytes[r], 8);
        }
        j->blob_bytes = b + 16;  // the decoder reads whole aligned dwords
        if ((rc = j->h_blobs.ensure(j->blob_bytes)) != SGK_OK) return rc;
        in->blobs = j->h_blobs.as<uint8_t>();
        in->blob_offsets = bo;
        if (signal_format == SGK_SIGNAL_ZREC) {
            // where every record inflates to (16-byte aligned), its room, and -- for the svb-zd decoder --
            // where its signal blob then lies and how long it is
            if ((rc = j->h_ioffs.ensure(nr1 * 8)) != SGK_OK) return rc;
            if ((rc = j->h_icaps.ensure(nr1 * 4)) != SGK_OK) return rc;
            if ((rc = j->h_soffs.ensure(nr1 * 8)) != SGK_OK) return rc;     // (blob offsets inside d_inflated)
            if ((rc = j->h_slens.ensure(nr1 * 4)) != SGK_OK) return rc;     // (blob lengths)
            uint64_t *io = j->h_ioffs.as<uint64_t>(), *so = j->h_soffs.as<uint64_t>();
            uint32_t *ic = j->h_icaps.as<uint32_t>(), *sl = j->h_slens.as<uint32_t>();
            uint64_t o2 = 0;
            for (size_t r = 0; r < nr; ++r) {
                if ((uint64_t)sig_offset[r] + sig_bytes[r] > rec_room[r]) return SGK_ERR_ARG;
                io[r] = o2;
                ic[r] = rec_room[r];
                so[r] = o2 + sig_offset[r];
                sl[r] = sig_bytes[r];
                o2 += round_up(rec_room[r], 16);
            }
            j->inflated_bytes = o2 + 16;
        }
    } else {
        if ((rc = j->h_samples.ensure(j->n_samples * sizeof(int16_t))) != SGK_OK) return rc;
        in->samples = j->h_samples.as<int16_t>();
    }
    in->offsets = offs;
    in->digitisation = j->h_dig.as<double>();
    in->offset = j->h_off.as<double>();
    in->range = j->h_rng.as<double>();
    in->n_samples = j->n_samples;
    j->begun = true;
    return SGK_OK;
}

static int h2d(GrowDev &d, const GrowPin &h, size_t bytes, hipStream_t st) {
    int rc = d.ensure(bytes ? bytes : 64);
    if (rc != SGK_OK) return rc;
    if (bytes) SGK_HIP_TRY(hipMemcpyAsync(d.p, h.p, bytes, hipMemcpyHostToDevice, st));
    return SGK_OK;
}
static int d2h(GrowPin &h, const GrowDev &d, size_t bytes, hipStream_t st) {
    int rc = h.ensure(bytes ? bytes : 64);
    if (rc != SGK_OK) return rc;
    if (bytes) SGK_HIP_TRY(hipMemcpyAsync(h.p, d.p, bytes, hipMemcpyDeviceToHost, st));
    return SGK_OK;
}

// enqueue the uploads of a staged batch (and the svb-zd decode) and describe the device batch
static int job_upload(sgk_job_t *j, sgk_batch_t *view) {
    const size_t nr = j->n_reads;
    hipStream_t st = j->st;
    int rc;
    if ((rc = h2d(j->d_offsets, j->h_offsets, nr * 8, st)) != SGK_OK) return rc;
    if ((rc = h2d(j->d_lengths, j->h_lengths, nr * 4, st)) != SGK_OK) return rc;
    if ((rc = h2d(j->d_dig, j->h_dig, nr * 8, st)) != SGK_OK) return rc;
    if ((rc = h2d(j->d_off, j->h_off, nr * 8, st)) != SGK_OK) return rc;
    if ((rc = h2d(j->d_rng, j->h_rng, nr * 8, st)) != SGK_OK) return rc;
    if (j->fmt == SGK_SIGNAL_ZREC) {
        // the records as they sit in the file -> inflated on the device -> their svb-zd blobs decoded from there
        if ((rc = j->d_samples.ensure(j->n_samples * sizeof(int16_t))) != SGK_OK) return rc;
        if ((rc = h2d(j->d_blobs, j->h_blobs, j->blob_bytes, st)) != SGK_OK) return rc;
        if ((rc = h2d(j->d_boffs, j->h_boffs, nr * 8, st)) != SGK_OK) return rc;
        if ((rc = h2d(j->d_blens, j->h_blens, nr * 4, st)) != SGK_OK) return rc;
        if ((rc = h2d(j->d_ioffs, j->h_ioffs, nr * 8, st)) != SGK_OK) return rc;
        if ((rc = h2d(j->d_icaps, j->h_icaps, nr * 4, st)) != SGK_OK) return rc;
        if ((rc = h2d(j->d_soffs, j->h_soffs, nr * 8, st)) != SGK_OK) return rc;
        if ((rc = h2d(j->d_slens, j->h_slens, nr * 4, st)) != SGK_OK) return rc;
        if ((rc = j->d_inflated.ensure(j->inflated_bytes)) != SGK_OK) return rc;
        if ((rc = j->d_ilens.ensure(nr * 4)) != SGK_OK) return rc;
        if ((rc = j->d_istat.ensure(nr * 4)) != SGK_OK) return rc;
        if ((rc = j->d_dstat.ensure(nr * 4)) != SGK_OK) return rc;
        rc = sgk_inflate(j->d_blobs.as<uint8_t>(), j->d_boffs.as<uint64_t>(), j->d_blens.as<uint32_t>(), j->n_reads,
                         j->d_inflated.as<uint8_t>(), j->d_ioffs.as<uint64_t>(), j->d_icaps.as<uint32_t>(),
                         j->d_ilens.as<uint32_t>(), j->d_istat.as<uint32_t>(), st);
        if (rc != SGK_OK) return rc;
        // (a record that did not inflate leaves whatever it leaves: its blob then fails the decoder's own checks or
        // decodes to garbage nobody reads -- sgk_job_wait refuses the batch on the inflate status)
        rc = sgk_svbzd_decode(j->d_inflated.as<uint8_t>(), j->d_soffs.as<uint64_t>(), j->d_slens.as<uint32_t>(), j->n_reads,
                              j->d_samples.as<int16_t>(), j->d_offsets.as<uint64_t>(), j->d_lengths.as<uint32_t>(),
                              j->d_dstat.as<uint32_t>(), st);
        if (rc != SGK_OK) return rc;
        if ((rc = d2h(j->h_dstat, j->d_dstat, nr * 4, st)) != SGK_OK) return rc;
        if ((rc = d2h(j->h_istat, j->d_istat, nr * 4, st)) != SGK_OK) return rc;
    } else if (j->fmt == SGK_SIGNAL_SVBZD) {
        if ((rc = j->d_samples.ensure(j->n_samples * sizeof(int16_t))) != SGK_OK) return rc;
        if ((rc = h2d(j->d_blobs, j->h_blobs, j->blob_bytes, st)) != SGK_OK) return rc;
        if ((rc = h2d(j->d_boffs, j->h_boffs, nr * 8, st)) != SGK_OK) return rc;
        if ((rc = h2d(j->d_blens, j->h_blens, nr * 4, st)) != SGK_OK) return rc;
        if ((rc = j->d_dstat.ensure(nr * 4)) != SGK_OK) return rc;
        rc = sgk_svbzd_decode(j->d_blobs.as<uint8_t>(), j->d_boffs.as<uint64_t>(), j->d_blens.as<uint32_t>(),
                              j->n_reads, j->d_samples.as<int16_t>(), j->d_offsets.as<uint64_t>(),
                              j->d_lengths.as<uint32_t>(), j->d_dstat.as<uint32_t>(), st);
        if (rc != SGK_OK) return rc;
        if ((rc = d2h(j->h_dstat, j->d_dstat, nr * 4, st)) != SGK_OK) return rc;
    } else {
        if ((rc = h2d(j->d_samples, j->h_samples, j->n_samples * sizeof(int16_t), st)) != SGK_OK) return rc;
    }
    view->samples = j->d_samples.as<int16_t>();
    view->offsets = j->d_offsets.as<uint64_t>();
    view->lengths = j->d_lengths.as<uint32_t>();
    view->digitisation = j->d_dig.as<double>();
    view->offset = j->d_off.as<double>();
    view->range = j->d_rng.as<double>();
    view->n_reads = j->n_reads;
    view->max_read_len = j->max_len;
    view->n_samples = j->n_samples;
    return SGK_OK;
}

// offs[r] = running sum of counts[0..r) with every count rounded up to `align` (a power of two); offs[n] = total.
// One 1024-thread workgroup: a contiguous slice of reads per thread, block scan of the slice sums in LDS.
// With `slots` (capacity arena, n+1 entries) a count is first clamped to its read's capacity.
__device__ inline uint64_t layout_item(const uint32_t *counts, const uint64_t *slots, uint32_t r, uint64_t m) {
    uint64_t c = counts[r];
    if (slots) {
        const uint64_t cap = slots[r + 1] - slots[r];
        c = c < cap ? c : cap;
    }
    return (c + m) & ~m;
}
__global__ __launch_bounds__(1024) void k_layout(const uint32_t *counts, const uint64_t *slots, uint32_t n, uint32_t align,
                                                 uint64_t *offs) {
    __shared__ uint64_t part[1024];
    const uint32_t t = threadIdx.x;
    const uint32_t per = (n + 1023u) / 1024u;
    const uint32_t lo = t * per < n ? t * per : n, hi = lo + per < n ? lo + per : n;
    const uint64_t m = (uint64_t)align - 1;
    uint64_t sum = 0;
    for (uint32_t r = lo; r < hi; ++r) sum += layout_item(counts, slots, r, m);
    part[t] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        const uint64_t v = t >= d ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    uint64_t o = part[t] - sum;
    for (uint32_t r = lo; r < hi; ++r) {
        offs[r] = o;
        o += layout_item(counts, slots, r, m);
    }
    if (t == 1023) offs[n] = part[1023];
}

// items of read r: src[k][slots[r] .. +counts[r]) -> dst[k][doffs[r] ..), k < narr (4-byte items)
struct GatherArgs {
    const uint32_t *src[4];
    uint32_t *dst[4];
};
__global__ __launch_bounds__(256) void k_gather(GatherArgs g, int narr, const uint64_t *slots, const uint32_t *counts,
                                                const uint64_t *doffs) {
    const uint32_t r = blockIdx.x;
    const uint64_t s = slots[r], d = doffs[r], cap = slots[r + 1] - s;
    const uint32_t c = counts[r] < cap ? counts[r] : (uint32_t)cap;  // an overflowing read keeps what fitted
    for (int k = 0; k < narr; ++k)
        for (uint32_t i = threadIdx.x; i < c; i += 256) g.dst[k][d + i] = g.src[k][s + i];
}

// events of read r: records src[slots[r] .. +counts[r]) -> the four dense arrays dst[k][doffs[r] ..), k < narr
// (start, length, mean, stdv as 4-byte items; narr = 2 for SGK_JOB_EVENTS_COMPACT; narr = 1: the lengths alone)
__global__ __launch_bounds__(256) void k_gather_events(const sgk_event_rec_t *src, GatherArgs g, int narr,
                                                       const uint64_t *slots, const uint32_t *counts,
                                                       const uint64_t *doffs) {
    const uint32_t r = blockIdx.x;
    const uint64_t s = slots[r], d = doffs[r], cap = slots[r + 1] - s;
    const uint32_t c = counts[r] < cap ? counts[r] : (uint32_t)cap;  // an overflowing read keeps what fitted
    for (uint32_t i = threadIdx.x; i < c; i += 256) {
        const uint4 v = *reinterpret_cast<const uint4 *>(src + s + i);
        if (narr == 1) {
            g.dst[0][d + i] = v.y;
            continue;
        }
        g.dst[0][d + i] = v.x;
        g.dst[1][d + i] = v.y;
        if (narr > 2) {
            g.dst[2][d + i] = v.z;
            g.dst[3][d + i] = v.w;
        }
    }
}

int sgk_job_submit(sgk_job_t *j, int tool, int rna, int pore, int flags) {
    if (!j || !j->begun || j->submitted) return SGK_ERR_ARG;
    if (tool < SGK_TOOL_PA || tool > SGK_TOOL_ENT) return SGK_ERR_ARG;
    SGK_HIP_TRY(hipSetDevice(j->device));
    j->tool = tool;
    j->flags = flags;
    const size_t nr = j->n_reads;
    hipStream_t st = j->st;
    int rc;
    if (nr == 0) {
        j->submitted = true;
        return SGK_OK;
    }
    sgk_batch_t view;
    if ((rc = job_upload(j, &view)) != SGK_OK) return rc;
    const uint32_t *lens = j->h_lengths.as<uint32_t>();
    // ---- kernels + results
    switch (tool) {
        case SGK_TOOL_PA: {
            const size_t bytes = j->n_samples * sizeof(float);
            if ((rc = j->d_out[0].ensure(bytes)) != SGK_OK) return rc;
            if ((rc = sgk_pa(&view, j->d_out[0].as<float>(), st)) != SGK_OK) return rc;
            if ((rc = d2h(j->h_out[0], j->d_out[0], bytes, st)) != SGK_OK) return rc;
            break;
        }
        case SGK_TOOL_EVENT:
        case SGK_TOOL_JNN: {
            const bool ev = tool == SGK_TOOL_EVENT;
            if ((rc = j->h_slots.ensure((nr + 1) * 8)) != SGK_OK) return rc;
            uint64_t *slots = j->h_slots.as<uint64_t>();
            uint64_t s = 0;
            for (size_t r = 0; r < nr; ++r) {
                slots[r] = s;
                s += ev ? sgk_event_slots_for(lens[r]) : sgk_jnn_slots_for(lens[r]);
            }
            slots[nr] = s;
            if ((rc = h2d(j->d_slots, j->h_slots, (nr + 1) * 8, st)) != SGK_OK) return rc;
            const int narr = ev ? 4 : 2;
            if (ev) {
                if ((rc = j->d_out[0].ensure(s * sizeof(sgk_event_rec_t))) != SGK_OK) return rc;
            } else {
                for (int k = 0; k < narr; ++k)
                    if ((rc = j->d_out[k].ensure(s * 4)) != SGK_OK) return rc;
            }
            if ((rc = j->d_cnt.ensure(nr * 4)) != SGK_OK) return rc;
            j->ws_bytes = ev ? sgk_event_workspace_bytes_opt(j->n_reads, j->n_samples, j->max_len, &j->ev_opt)
                             : sgk_jnn_workspace_bytes(j->n_reads, j->n_samples, j->max_len);
            if ((rc = j->d_ws.ensure(j->ws_bytes)) != SGK_OK) return rc;
            if (ev)
                rc = sgk_event_opt(&view, rna, j->d_slots.as<uint64_t>(), j->d_out[0].as<sgk_event_rec_t>(),
                                   j->d_cnt.as<uint32_t>(), j->d_ws.p, j->d_ws.cap, st, &j->ev_opt);
            else
                rc = sgk_jnn_opt(&view, rna, j->d_slots.as<uint64_t>(), j->d_out[0].as<int32_t>(),
                                 j->d_out[1].as<int32_t>(), j->d_cnt.as<uint32_t>(), j->d_ws.p, j->d_ws.cap, st, &j->st_opt);
            if (rc != SGK_OK) return rc;
            if (!ev && (rc = fetch_long_hdr(j, st)) != SGK_OK) return rc;
            if ((rc = d2h(j->h_cnt, j->d_cnt, nr * 4, st)) != SGK_OK) return rc;
            // the arena is capacity-sized (sgk_event_slots_for(n) slots per read): gather what was produced into dense ranges on the
            // device and download only that (sgk_job_wait fetches the arrays once the total is known)
            const int ncopy = (ev && (flags & SGK_JOB_EVENTS_LENGTHS)) ? 1 : ((ev && (flags & SGK_JOB_EVENTS_COMPACT)) ? 2 : narr);
            if ((rc = j->d_doffs.ensure((nr + 1) * 8)) != SGK_OK) return rc;
            hipLaunchKernelGGL(k_layout, dim3(1), dim3(1024), 0, st, j->d_cnt.as<uint32_t>(), j->d_slots.as<uint64_t>(),
                               j->n_reads, 1u, j->d_doffs.as<uint64_t>());
            SGK_HIP_TRY(hipGetLastError());
            GatherArgs g;
            for (int k = 0; k < 4; ++k) { g.src[k] = nullptr; g.dst[k] = nullptr; }
            for (int k = 0; k < ncopy; ++k) {
                if ((rc = j->d_dense[k].ensure(s * 4)) != SGK_OK) return rc;
                g.src[k] = ev ? nullptr : j->d_out[k].as<uint32_t>();
                g.dst[k] = j->d_dense[k].as<uint32_t>();
            }
            if (ev)
                hipLaunchKernelGGL(k_gather_events, dim3(j->n_reads), dim3(256), 0, st, j->d_out[0].as<sgk_event_rec_t>(),
                                   g, ncopy, j->d_slots.as<uint64_t>(), j->d_cnt.as<uint32_t>(), j->d_doffs.as<uint64_t>());
            else
                hipLaunchKernelGGL(k_gather, dim3(j->n_reads), dim3(256), 0, st, g, ncopy, j->d_slots.as<uint64_t>(),
                                   j->d_cnt.as<uint32_t>(), j->d_doffs.as<uint64_t>());
            SGK_HIP_TRY(hipGetLastError());
            if ((rc = d2h(j->h_doffs, j->d_doffs, (nr + 1) * 8, st)) != SGK_OK) return rc;
            if (!ev) {  // sgk_jnn counts the reads whose segments overflowed their slots in the workspace's first word
                if ((rc = j->h_err.ensure(64)) != SGK_OK) return rc;
                SGK_HIP_TRY(hipMemcpyAsync(j->h_err.p, j->d_ws.p, 4, hipMemcpyDeviceToHost, st));
            }
            j->n_dense = ncopy;
            break;
        }
        case SGK_TOOL_STAT: {
            if ((rc = j->d_out[0].ensure(nr * sizeof(sgk_stat_rec_t))) != SGK_OK) return rc;
            j->ws_bytes = sgk_stat_workspace_bytes(j->n_reads, j->n_samples, j->max_len);
            if ((rc = j->d_ws.ensure(j->ws_bytes)) != SGK_OK) return rc;
            rc = sgk_stat_opt(&view, j->d_out[0].as<sgk_stat_rec_t>(), j->d_ws.p, j->d_ws.cap, st, &j->st_opt);
            if (rc != SGK_OK) return rc;
            if ((rc = fetch_long_hdr(j, st)) != SGK_OK) return rc;
            if ((rc = d2h(j->h_out[0], j->d_out[0], nr * sizeof(sgk_stat_rec_t), st)) != SGK_OK) return rc;
            break;
        }
        case SGK_TOOL_ENT: {
            const size_t ob = j->n_samples * sizeof(uint16_t);
            if ((rc = j->d_out[0].ensure(nr * sizeof(sgk_ent_hist_t))) != SGK_OK) return rc;
            if ((rc = j->d_out[1].ensure(ob)) != SGK_OK) return rc;
            if ((rc = j->d_out[2].ensure(ob)) != SGK_OK) return rc;
            rc = sgk_ent(&view, j->d_out[0].as<sgk_ent_hist_t>(), j->d_out[1].as<uint16_t>(), j->d_out[2].as<uint16_t>(), st);
            if (rc != SGK_OK) return rc;
            if ((rc = d2h(j->h_out[0], j->d_out[0], nr * sizeof(sgk_ent_hist_t), st)) != SGK_OK) return rc;
            break;  // the overflow lists are fetched by sgk_job_wait only when some read has entries
        }
        case SGK_TOOL_PREFIX: {
            if ((rc = j->d_out[0].ensure(nr * sizeof(sgk_prefix_rec_t))) != SGK_OK) return rc;
            j->ws_bytes = sgk_prefix_workspace_bytes(j->n_reads, j->n_samples, j->max_len);
            if ((rc = j->d_ws.ensure(j->ws_bytes)) != SGK_OK) return rc;
            rc = sgk_prefix_opt(&view, rna, pore, j->d_out[0].as<sgk_prefix_rec_t>(), j->d_ws.p, j->d_ws.cap, st, &j->st_opt);
            if (rc != SGK_OK) return rc;
            if ((rc = fetch_long_hdr(j, st)) != SGK_OK) return rc;
            if ((rc = d2h(j->h_out[0], j->d_out[0], nr * sizeof(sgk_prefix_rec_t), st)) != SGK_OK) return rc;
            break;
        }
    }
    j->submitted = true;
    return SGK_OK;
}

int sgk_job_submit_qts(sgk_job_t *j, int bits, int method, int out_fmt) {
    if (!j || !j->begun || j->submitted) return SGK_ERR_ARG;
    if (out_fmt != SGK_SIGNAL_INT16 && out_fmt != SGK_SIGNAL_SVBZD) return SGK_ERR_ARG;
    if (bits < 1 || bits > 15 || method < SGK_QTS_FLOOR || method > SGK_QTS_FILL_ONES) return SGK_ERR_ARG;
    SGK_HIP_TRY(hipSetDevice(j->device));
    j->tool = SGK_TOOL_QTS;
    j->flags = out_fmt;
    const size_t nr = j->n_reads;
    hipStream_t st = j->st;
    int rc;
    if (nr == 0) {
        j->submitted = true;
        return SGK_OK;
    }
    sgk_batch_t view;
    if ((rc = job_upload(j, &view)) != SGK_OK) return rc;
    int16_t *smp = j->d_samples.as<int16_t>();
    if ((rc = sgk_qts(smp, view.offsets, view.lengths, j->n_reads, j->max_len, bits, method, st)) != SGK_OK) return rc;
    if (out_fmt == SGK_SIGNAL_INT16) {
        if ((rc = d2h(j->h_out[0], j->d_samples, j->n_samples * sizeof(int16_t), st)) != SGK_OK) return rc;
    } else {
        // d_out[0] blobs (worst-case sized), d_out[1] blob offsets, d_out[2] total, d_cnt blob lengths
        const uint32_t *lens = j->h_lengths.as<uint32_t>();
        size_t bound = 0;
        for (size_t r = 0; r < nr; ++r) bound += round_up(4 + ((size_t)lens[r] + 3) / 4 + 3 * (size_t)lens[r], 8);
        if ((rc = j->d_out[0].ensure(bound + 16)) != SGK_OK) return rc;
        if ((rc = j->d_out[1].ensure((nr + 1) * 8)) != SGK_OK) return rc;
        if ((rc = j->d_cnt.ensure(nr * 4)) != SGK_OK) return rc;
        if ((rc = sgk_svbzd_size(smp, view.offsets, view.lengths, j->n_reads, j->d_cnt.as<uint32_t>(), st)) != SGK_OK) return rc;
        hipLaunchKernelGGL(k_layout, dim3(1), dim3(1024), 0, st, j->d_cnt.as<uint32_t>(),
                           static_cast<const uint64_t *>(nullptr), j->n_reads, 8u, j->d_out[1].as<uint64_t>());
        SGK_HIP_TRY(hipGetLastError());
        rc = sgk_svbzd_encode(smp, view.offsets, view.lengths, j->n_reads, j->d_out[0].as<uint8_t>(),
                              j->d_out[1].as<uint64_t>(), j->d_cnt.as<uint32_t>(), st);
        if (rc != SGK_OK) return rc;
        if ((rc = d2h(j->h_cnt, j->d_cnt, nr * 4, st)) != SGK_OK) return rc;
        if ((rc = d2h(j->h_out[1], j->d_out[1], (nr + 1) * 8, st)) != SGK_OK) return rc;
        // the blobs themselves are fetched by sgk_job_wait once their total size is known
    }
    j->submitted = true;
    return SGK_OK;
}

int sgk_job_wait(sgk_job_t *j) {
    if (!j || !j->submitted) return SGK_ERR_ARG;
    SGK_HIP_TRY(hipSetDevice(j->device));
    j->submitted = false;  // the staged batch stays valid: it may be submitted again (e.g. with another tool)
    memset(&j->ev_status, 0, sizeof j->ev_status);
    SGK_HIP_TRY(hipStreamSynchronize(j->st));
    j->long_declined = 0u;
    if (j->n_reads == 0) return SGK_OK;
    if (j->long_fetched) {
        // n_timeouts of sgk_long_status_t: long reads whose workgroups gave up at a barrier; they were redone on one
        // wavefront, the records are right -- a caller that wants to know (the CLI warns) asks sgk_job_long_declined
        j->long_declined = reinterpret_cast<const LongHdr *>(j->h_long.p)->n_declined;
        j->long_fetched = false;
    }
    if (j->fmt == SGK_SIGNAL_ZREC) {
        // a record that did not inflate (malformed stream, check value, more bytes than its head announced): as the
        // reference's slow5_get_next error.  decode_status carries 0x100 | the inflate status for such a read.
        uint32_t *ds = j->h_dstat.as<uint32_t>();
        const uint32_t *is = j->h_istat.as<uint32_t>();
        bool bad = false;
        for (uint32_t r = 0; r < j->n_reads; ++r) {
            if (is[r] != 0) ds[r] = 0x100u | is[r];
            bad = bad || ds[r] != 0;
        }
        if (bad) return SGK_ERR_FORMAT;
    }
    if (j->fmt == SGK_SIGNAL_SVBZD) {
        const uint32_t *ds = j->h_dstat.as<uint32_t>();
        for (uint32_t r = 0; r < j->n_reads; ++r)
            if (ds[r] != 0) return SGK_ERR_FORMAT;
    }
    if (j->tool == SGK_TOOL_EVENT || j->tool == SGK_TOOL_JNN) {
        const uint64_t total = j->h_doffs.as<uint64_t>()[j->n_reads];
        int rc;
        for (int k = 0; k < j->n_dense; ++k)
            if ((rc = d2h(j->h_out[k], j->d_dense[k], (size_t)total * 4, j->st)) != SGK_OK) return rc;
        SGK_HIP_TRY(hipStreamSynchronize(j->st));
    }
    if (j->tool == SGK_TOOL_JNN) {
        uint32_t nerr;
        memcpy(&nerr, j->h_err.p, 4);
        if (nerr) return SGK_ERR_CAPACITY;
    }
    if (j->tool == SGK_TOOL_EVENT) return sgk_event_status(j->d_ws.p, &j->ev_status, j->st);
    if (j->tool == SGK_TOOL_QTS && j->flags == SGK_SIGNAL_SVBZD) {
        const uint64_t total = j->h_out[1].as<uint64_t>()[j->n_reads];
        int rc;
        if ((rc = d2h(j->h_out[0], j->d_out[0], (size_t)total, j->st)) != SGK_OK) return rc;
        SGK_HIP_TRY(hipStreamSynchronize(j->st));
    }
    if (j->tool == SGK_TOOL_ENT) {
        const sgk_ent_hist_t *rec = j->h_out[0].as<sgk_ent_hist_t>();
        j->ent_over = false;
        for (uint32_t r = 0; r < j->n_reads && !j->ent_over; ++r) j->ent_over = rec[r].n_over_raw || rec[r].n_over_delta;
        if (j->ent_over) {
            const size_t ob = j->n_samples * sizeof(uint16_t);
            int rc;
            if ((rc = d2h(j->h_out[1], j->d_out[1], ob, j->st)) != SGK_OK) return rc;
            if ((rc = d2h(j->h_out[2], j->d_out[2], ob, j->st)) != SGK_OK) return rc;
            SGK_HIP_TRY(hipStreamSynchronize(j->st));
        }
    }
    return SGK_OK;
}

uint32_t sgk_job_long_declined(const sgk_job_t *j) { return j ? j->long_declined : 0u; }

int sgk_job_output(const sgk_job_t *j, sgk_job_output_t *out) {
    if (!j || !out) return SGK_ERR_ARG;
    memset(out, 0, sizeof *out);
    out->n_reads = j->n_reads;
    out->offsets = j->h_offsets.as<uint64_t>();
    out->lengths = j->h_lengths.as<uint32_t>();
    out->decode_status = (j->fmt == SGK_SIGNAL_SVBZD || j->fmt == SGK_SIGNAL_ZREC) ? j->h_dstat.as<uint32_t>() : nullptr;
    switch (j->tool) {
        case SGK_TOOL_PA:
            out->pa = j->h_out[0].as<float>();
            break;
        case SGK_TOOL_EVENT:
            out->slots = j->h_doffs.as<uint64_t>();
            out->counts = j->h_cnt.as<uint32_t>();
            if (j->flags & SGK_JOB_EVENTS_LENGTHS) {
                out->ev_length = j->h_out[0].as<uint32_t>();
                out->event_status = j->ev_status;
                break;
            }
            out->ev_start = j->h_out[0].as<uint32_t>();
            out->ev_length = j->h_out[1].as<uint32_t>();
            if (!(j->flags & SGK_JOB_EVENTS_COMPACT)) {
                out->ev_mean = j->h_out[2].as<float>();
                out->ev_stdv = j->h_out[3].as<float>();
            }
            out->event_status = j->ev_status;
            break;
        case SGK_TOOL_JNN:
            out->slots = j->h_doffs.as<uint64_t>();
            out->counts = j->h_cnt.as<uint32_t>();
            out->seg_x = j->h_out[0].as<int32_t>();
            out->seg_y = j->h_out[1].as<int32_t>();
            break;
        case SGK_TOOL_STAT:
            out->stat = j->h_out[0].as<sgk_stat_rec_t>();
            break;
        case SGK_TOOL_PREFIX:
            out->prefix = j->h_out[0].as<sgk_prefix_rec_t>();
            break;
        case SGK_TOOL_QTS:
            if (j->flags == SGK_SIGNAL_SVBZD) {
                out->qts_blobs = j->h_out[0].as<uint8_t>();
                out->qts_blob_offsets = j->h_out[1].as<uint64_t>();
                out->qts_blob_lengths = j->h_cnt.as<uint32_t>();
            } else {
                out->qts_samples = j->h_out[0].as<int16_t>();
            }
            break;
        case SGK_TOOL_ENT:
            out->ent = j->h_out[0].as<sgk_ent_hist_t>();
            out->ent_over_raw = j->ent_over ? j->h_out[1].as<uint16_t>() : nullptr;
            out->ent_over_delta = j->ent_over ? j->h_out[2].as<uint16_t>() : nullptr;
            break;
        default:
            return SGK_ERR_ARG;
    }
    return SGK_OK;
}

}  // extern "C"
