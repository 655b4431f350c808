/* sigtk_gpu.h -- C ABI of libsigtk_gpu.so: sigtk's per-read raw-signal hot path
 * (pa / event / stat / jnn / prefix) as batched HIP kernels for AMD MI355X (gfx950).
 *
 * The reference (hasindu2008/sigtk, C99, single-threaded) has no FFI layer: its seam is
 * the per-record function-pointer slot `void (*func)(slow5_rec_t*, opt_t)` in
 * src/cmain.c:95-120 and, below it, the compute entry points declared in
 * src/sigtk.h:124-134 and src/jnn.h:104-109.  This header is what a sigtk maintainer
 * would bind instead of those entry points (see INTEGRATION.md for the stub):
 *
 *   reference (file:line)                         replaced by
 *   --------------------------------------------  -------------------------------------
 *   signal_in_picoamps   src/misc.c:15            sgk_pa            / sgk_signal_in_picoamps
 *   getevents            src/events.c:553         sgk_event         / sgk_getevents
 *   meanf..mediani16     src/stat.h:17-73         sgk_stat
 *   jnn_raw/jnn_print    src/jnn.c:282,309        sgk_jnn
 *   find_adaptor/jnnv2   src/jnn.c:99,181         sgk_prefix
 *   find_polya/jnn_pa    src/jnn.c:295,352        sgk_prefix
 *   prefix_func compute  src/cfunc.c:169-216      sgk_prefix
 *   entropy (ent)        src/ent.c:25-50,107-164  sgk_ent + sgk_ent_finish
 *   qts quantisers       src/qts.c:27-43,126-142  sgk_qts
 *   svb-zd decode/encode slow5_press.c:1063-1146  sgk_svbzd_decode / sgk_svbzd_size + sgk_svbzd_encode
 *   the record loop      src/cmain.c:118-120      sgk_job_* (pipelined batches: staging, stream, results)
 *
 * Conventions
 *   - Plain C: pointers and sizes only.  No exceptions, no exit(): every function returns
 *     SGK_OK (0) or a negative sgk error code; sgk_strerror() names it.
 *   - "Device API" functions take DEVICE pointers and a hipStream_t passed as void*
 *     (NULL = default stream) and only enqueue work; nothing is synchronised.
 *   - "Jobs" (sgk_job_*) are the throughput path for host callers: pinned staging the caller fills, asynchronous
 *     upload / kernels / download on a private stream, results in pinned buffers; several can be in flight.
 *   - "Host API" functions (suffix _host and the per-read shims) take HOST pointers,
 *     stage through the GPU and synchronise before returning.  Results they allocate are
 *     released with the matching *_free function.
 *   - The library never falls back to a CPU implementation: without a usable GPU every
 *     compute entry point returns SGK_ERR_NODEVICE / SGK_ERR_HIP.
 *
 * Batch layout (structure of arrays, one batch = many reads):
 *   samples  int16 raw samples of all reads in one buffer; 16-byte aligned; n_samples is the
 *            readable length of the buffer in samples and must be a multiple of 8;
 *   offsets  n_reads sample indices, lengths n_reads sample counts: read r is
 *            samples[offsets[r] .. offsets[r]+lengths[r]).  Reads must be stored in
 *            increasing, non-overlapping order; gaps between reads are allowed (and ignored),
 *            which lets a packer start every read on an aligned boundary.  Any offsets are
 *            accepted; reads starting on a multiple of 8 samples (16 bytes) take the
 *            vectorised load paths (the host packer aligns to 64 samples = 128 bytes).  The
 *            event fast path additionally wants >= 64 (DNA) / 256 (RNA parameters) readable samples in front of
 *            a read and >= 16 behind it (neighbouring reads or padding, any content); reads without that room, or
 *            starting on an odd sample, are processed by the slower exact fallback kernel;
 *   digitisation/offset/range  the three per-read doubles of slow5_rec_t
 *            (slow5lib/include/slow5/slow5_defs.h:84-92), narrowed to float on the device
 *            exactly as src/misc.c:17-19 does.
 */
#ifndef SIGTK_GPU_H
#define SIGTK_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 0.2.0: per-call options (sgk_event_options_t, sgk_stat_options_t) replace the process-wide sgk_event_configure* and
 * the environment variables of 0.1.0; sgk_event_plan takes the options; sgk_event_plan_t / sgk_event_status_t as below.
 * Bindings should compare sgk_version() with the header they were written against.
 * 0.2.1: sgk_stat_options_t::long_min (was reserved), sgk_stat_long_status, sgk_stat_plan; sgk_stat / sgk_jnn / sgk_prefix_workspace_bytes
 *        ask for the long reads' records as well (an older, smaller workspace still works: long reads then run on one
 *        wavefront).
 * 0.2.2: the long-read path of stat / jnn / prefix declines a read whose workgroups time out at a barrier and the wave
 *        kernel redoes it (no result depends on the long path having worked; sgk_long_status_t::n_timeouts counts those
 *        reads); sgk_stat_options_t::debug_fault (was reserved[0]); sgk_job_long_declined; sgk_inflate, SGK_SIGNAL_ZREC / sgk_job_begin_zrec;
 *        the six-argument plan call is sgk_event_plan_opt; sgk_event_plan is the 0.1.0 five-argument form again (deprecated).
 * 0.2.3: SGK_JOB_EVENTS_LENGTHS (additive submit flag). */
#define SGK_VERSION_STRING "0.2.3"

/* ---- error codes --------------------------------------------------------------- */
#define SGK_OK 0
#define SGK_ERR_ARG (-1)        /* NULL / inconsistent argument                         */
#define SGK_ERR_HIP (-2)        /* a HIP runtime call failed (sgk_last_hip_error())      */
#define SGK_ERR_NODEVICE (-3)   /* no usable GPU                                        */
#define SGK_ERR_WORKSPACE (-4)  /* workspace too small (see *_workspace_bytes)          */
#define SGK_ERR_CAPACITY (-5)   /* an output arena slot range was too small             */
#define SGK_ERR_ALIGN (-6)      /* samples buffer not 16-byte aligned                   */
#define SGK_ERR_NOMEM (-7)      /* host allocation failed                               */
#define SGK_ERR_FORMAT (-8)     /* malformed compressed signal (svb-zd blob)            */

const char *sgk_strerror(int code);
const char *sgk_version(void);
/* text of the last failing HIP call on this thread ("" if none) */
const char *sgk_last_hip_error(void);
/* number of visible GPUs (0 if none / HIP unusable); never fails */
int sgk_device_count(void);
/* bind the calling thread to a GPU ordinal */
int sgk_set_device(int ordinal);

/* pore ids, as OPT_PORE_* in src/sigtk.h:50-52 */
#define SGK_PORE_R9 0
#define SGK_PORE_R10 1
#define SGK_PORE_RNA004 2

/* ---- batch view (device pointers) ---------------------------------------------- */
typedef struct sgk_batch {
    const int16_t *samples;
    const uint64_t *offsets;   /* n_reads: first sample of each read */
    const uint32_t *lengths;   /* n_reads: samples in each read (len_raw_signal) */
    const double *digitisation;
    const double *offset;
    const double *range;
    uint32_t n_reads;
    uint32_t max_read_len; /* max over reads of lengths[r]; host-known */
    uint64_t n_samples;    /* readable length of `samples` (multiple of 8; all reads lie inside) */
} sgk_batch_t;

/* ---- pa: int16 -> picoamps (src/misc.c:15-32) ----------------------------------- */
/* pa_out[offsets[r]+i] for every sample i of every read r (same layout as `samples`;
 * gap samples are not written). */
int sgk_pa(const sgk_batch_t *batch, float *pa_out, void *stream);

/* ---- event: Scrappie-derived event detection (src/events.c:553) ------------------ */
/* One event = event_t of src/sigtk.h:55-62 with integer start / length (16 bytes). */
typedef struct sgk_event_rec {
    uint32_t start;   /* first raw sample of the event                       */
    uint32_t length;  /* samples (event_t.length is this number as a float)  */
    float mean;       /* pA                                                  */
    float stdv;       /* pA                                                  */
} sgk_event_rec_t;
/* Events of read r are written to events[ev_slots[r] .. ev_slots[r]+n_events[r]-1] (16-byte aligned array of
 * records).  ev_slots (device, n_reads+1, increasing) is an INPUT describing the arena: read r may use at most
 * ev_slots[r+1]-ev_slots[r] slots.  sgk_event_slots_for() gives a capacity that can never overflow (peaks are at
 * least 3 samples apart).  If a read would overflow, its surplus events are dropped, n_events[r] still holds the
 * true count and the status word reports it (sgk_event_status -> SGK_ERR_CAPACITY).
 * Where the reference aborts or is undefined the library defines: a read with no peak
 * (incl. reads shorter than 2*window) yields one event [0,n); empty reads yield none. */
static inline uint64_t sgk_event_slots_for(uint64_t n_samples_of_read) { return n_samples_of_read / 3 + 2; }

/* (the tail split is planned from the CURRENT device's compute units -- 256 without one: size a workspace with the
 * device the launch will use current, sgk_set_device; a workspace sized for another device makes the launch return
 * SGK_ERR_WORKSPACE, never write out of bounds) */
size_t sgk_event_workspace_bytes(uint32_t n_reads, uint64_t n_samples, uint32_t max_read_len);

int sgk_event(const sgk_batch_t *batch, int rna, const uint64_t *ev_slots, sgk_event_rec_t *events,
              uint32_t *n_events, void *workspace, size_t workspace_bytes, void *stream);

/* Same path fed with pA floats instead of raw int16 (drop-in for getevents(), whose input
 * is the pA array): pa is packed like `samples` (float per sample, 16-byte aligned). */
int sgk_event_pa(const float *pa, const uint64_t *offsets, const uint32_t *lengths, uint32_t n_reads,
                 uint32_t max_read_len, uint64_t n_samples, int rna, const uint64_t *ev_slots,
                 sgk_event_rec_t *events, uint32_t *n_events, void *workspace, size_t workspace_bytes, void *stream);

typedef struct sgk_event_status {
    uint32_t n_fallback_reads;   /* reads re-done by the sequential-prefix exact path (no room around the read,
                                  * the exactness guard on the sample magnitudes failed, inf/nan samples, or
                                  * more than 512 event boundaries within 2048 samples): same results, slower */
    uint32_t n_rerun_passes;     /* speculative chunk boundaries that needed a re-run     */
    uint32_t n_capacity_overflow;/* reads whose events did not fit their slot range       */
    uint32_t n_long_replays;     /* lanes (chunks) in which a run of the lazily evaluated long detector could not be
                                  * proven silent and was re-played exactly (diagnostic)  */
    uint64_t n_events_total;
    uint32_t n_split_reads;      /* reads taken by several wavefronts: long reads (from long_min samples on, in
                                  * segments of segment_len) and the reads of the tail split */
    uint32_t n_segments;         /* their segments                                         */
    uint32_t n_seam_reruns;      /* segments whose speculative start was wrong and that were run again */
    uint32_t reserved;
    uint64_t n_replay_indices;   /* indices the exact replay of the long detector walked (diagnostic: one lane each) */
} sgk_event_status_t;
/* Per-call options of the event path.  All zero (or a null pointer) = the defaults.  The library keeps no mutable
 * process-wide configuration and reads no environment variable: two callers in one process may use different options
 * at the same time; size a workspace with the options the call will use.  Results do not depend on any of them.
 *   segment_len / long_min   a read of at least long_min samples is cut into segments of segment_len samples (a multiple
 *                            of 1024), one wavefront each      (both 0: chosen per batch -- long_min = 0.9 x the
 *                            batch's samples per wavefront slot, between 131 072 and 262 144, segments of half of it;
 *                            sgk_event_plan_opt tells)
 *   warmup                   speculative warm-up in samples, a multiple of 16, <= 512  (0: the presets' own; tests use
 *                            16 to make speculation fail at every few chunk boundaries)
 *   lanes_per_short_read     short reads (< short_max samples) in large batches: a wavefront takes 64 / lanes reads,
 *                            lanes lanes each (a power of two, 1 .. 32)           (0: chosen per batch, -1: off)
 *   short_max                (0: 16 384, 65 536 with RNA parameters; a power of two >= 1024)
 *   tail_split               a batch of fewer than 8 rounds of wavefronts: the reads of its last, partial round are cut
 *                            into segments so that the GPU drains on small units  (0: chosen per batch -- where it
 *                            was measured to pay: a small batch, a small last round, DESIGN.md 3.1 --, -1: off,
 *                            n > 0: the last n reads are cut whatever the batch -- tuning / tests) */
typedef struct sgk_event_options {
    uint32_t segment_len, long_min;
    int32_t warmup;
    int32_t lanes_per_short_read;
    uint32_t short_max;
    int32_t tail_split;
    uint32_t reserved[2];
} sgk_event_options_t; /* 32 bytes */
size_t sgk_event_workspace_bytes_opt(uint32_t n_reads, uint64_t n_samples, uint32_t max_read_len,
                                     const sgk_event_options_t *opt);
int sgk_event_opt(const sgk_batch_t *batch, int rna, const uint64_t *ev_slots, sgk_event_rec_t *events,
                  uint32_t *n_events, void *workspace, size_t workspace_bytes, void *stream,
                  const sgk_event_options_t *opt);
int sgk_event_pa_opt(const float *pa, const uint64_t *offsets, const uint32_t *lengths, uint32_t n_reads,
                     uint32_t max_read_len, uint64_t n_samples, int rna, const uint64_t *ev_slots,
                     sgk_event_rec_t *events, uint32_t *n_events, void *workspace, size_t workspace_bytes, void *stream,
                     const sgk_event_options_t *opt);

/* How sgk_event_opt would take a batch with these totals under `opt` (host only, no GPU work; the tail split assumes
 * the current device's CU count, 256 without a device): the segment geometry and list capacities of the long reads, the
 * threshold and the lanes per read of the short ones (0: every read has a wavefront of its own), the tail split.
 * Run it with the device the launch will use current (sgk_set_device): so must sgk_event_workspace_bytes(_opt). */
typedef struct sgk_event_plan {
    uint32_t segment_len, long_min;         /* long reads: >= long_min samples, cut into segments of segment_len */
    uint32_t max_segments, max_long_reads;  /* capacities reserved in the workspace (0: no read is cut)           */
    uint32_t short_max;                     /* reads under this many samples are short ...                      */
    uint32_t lanes_per_short_read;          /* ... and get this many lanes each (0: packing is off for the batch) */
    uint32_t warmup_override;               /* 0: the presets' warm-ups                                          */
    uint32_t tail_split_from;               /* tail split: reads at dispatch positions >= this (n_reads: none) ... */
    uint32_t tail_segment_len;              /* ... are cut into segments of this many samples (0: none)           */
    uint32_t reserved[3];
} sgk_event_plan_t;
int sgk_event_plan_opt(uint32_t n_reads, uint64_t n_samples, uint32_t max_read_len, int rna,
                       const sgk_event_options_t *opt, sgk_event_plan_t *out);
/* Deprecated: the 0.1.0 call under its 0.1.0 name and signature -- the defaults, and only the first 32 bytes of the
 * struct (what sgk_event_plan_t was then: up to warmup_override + one reserved word).  0.2.0 / 0.2.1 had the
 * six-argument form under this symbol; it is sgk_event_plan_opt now, so that no caller can bind the wrong one. */
int sgk_event_plan(uint32_t n_reads, uint64_t n_samples, uint32_t max_read_len, int rna, void *out32);

/* Synchronises `stream`, copies the status block of the last sgk_event on this workspace.
 * Returns SGK_ERR_CAPACITY if any read overflowed its slots. */
int sgk_event_status(const void *workspace, sgk_event_status_t *out, void *stream);

/* ---- stat: per-read mean/std/median of raw and pA (src/stat.h, src/cfunc.c:126-159) */
typedef struct sgk_stat_rec {
    float raw_mean, pa_mean, raw_std, pa_std;
    int32_t raw_median;
    float pa_median;
    uint32_t n;
    uint32_t reserved;
} sgk_stat_rec_t; /* 32 bytes per read */

size_t sgk_stat_workspace_bytes(uint32_t n_reads, uint64_t n_samples, uint32_t max_read_len);
int sgk_stat(const sgk_batch_t *batch, sgk_stat_rec_t *out, void *workspace, size_t workspace_bytes,
             void *stream);
/* fused stat + pa (BASELINE config 4): one launch sequence producing both outputs */
int sgk_stat_pa(const sgk_batch_t *batch, sgk_stat_rec_t *out, float *pa_out, void *workspace,
                size_t workspace_bytes, void *stream);

/* ---- jnn: state-machine segmenter on raw signal (src/jnn.c:190-350) --------------- */
/* Segments of read r go to slots seg_slots[r].. (same arena convention as events);
 * seg_x/seg_y are jnn_pair_t (src/jnn.h:13-16) as SoA with 32-bit members. */
/* sgk_jnn may use ALL slots of a read as scratch (the wave-per-read kernel stages kept segments in the upper half
 * before merging them into the lower half): only the first n_segs[r] entries are results afterwards. */
static inline uint64_t sgk_jnn_slots_for(uint64_t n_samples_of_read) { return n_samples_of_read / 32 + 2; }
size_t sgk_jnn_workspace_bytes(uint32_t n_reads, uint64_t n_samples, uint32_t max_read_len);
int sgk_jnn(const sgk_batch_t *batch, int rna, const uint64_t *seg_slots, int32_t *seg_x, int32_t *seg_y,
            uint32_t *n_segs, void *workspace, size_t workspace_bytes, void *stream);

/* ---- prefix: adaptor / polyA finder (src/cfunc.c:169-216, src/jnn.c:99-188,352-374) */
typedef struct sgk_prefix_rec {
    int32_t adapt_x, adapt_y;  /* find_adaptor(): -1/-1 read too short, 0/0 none found  */
    int32_t polya_x, polya_y;  /* find_polya(), RELATIVE to adapt_y; -1/-1 if none      */
    float adapt_mean, adapt_std, adapt_median;
    float polya_mean, polya_std, polya_median;
    uint32_t n;
    uint32_t reserved;
} sgk_prefix_rec_t; /* 48 bytes per read */

size_t sgk_prefix_workspace_bytes(uint32_t n_reads, uint64_t n_samples, uint32_t max_read_len);
int sgk_prefix(const sgk_batch_t *batch, int rna, int pore, sgk_prefix_rec_t *out, void *workspace,
               size_t workspace_bytes, void *stream);

/* Per-call options of stat / jnn / prefix (null = defaults).  The library has two implementations of these subtools:
 * one read per wavefront (round 2; any batch) and one read per lane (round 1; wins on large batches of short reads of
 * similar length: stat >= 49 152 reads of <= 32 768 samples or >= 16 384 reads of <= 16 384 (without the pA output also
 * >= 81 920 reads of <= 131 072), jnn >= 65 536 reads of <= 12 288 samples,
 * the longest <= 1.5 x the mean; prefix: the wave finders, and the lane kernels for the statistics of the regions they find
 * when the batch has >= 49 152 reads; sgk_stat_plan tells).  kernels: 0 chosen per batch,
 * 1 one read per lane, 2 one read per wavefront.  Results do not depend on it (the tests compare the two bit for bit). */
typedef struct sgk_stat_options {
    int32_t kernels;
    int32_t long_min;   /* reads of at least this many samples get 16 workgroups (64 wavefronts) of their own before
                         * the wave-per-read kernel runs (sequential float sums composed from tile summaries, stat's
                         * histogram and pA output, jnn's automaton): 0 = chosen per batch and tool, max(n_samples / 2048 (jnn: / 3072),
                         * a floor between 131 072 and 262 144 by the size of the batch; sgk_stat_plan tells)
                         * -- the reads one wavefront would still be busy with when the rest of the batch is done (and
                         * none if that still lists more than 128 reads: a batch of similar long reads balances itself);
                         * -1 = never; else the threshold (>= 8 192).  Results do not depend on it.  Needs the workspace
                         * sgk_*_workspace_bytes asks for (with less, such reads run on one wavefront). */
    uint32_t debug_fault; /* 0.  Tests only: fault injection into the long-read path's barriers (a withheld workgroup /
                           * a tiny spin bound, see lc_barrier in csrc/stat_kernels.hip) to exercise the decline-and-redo path */
    uint32_t reserved;
} sgk_stat_options_t;
int sgk_stat_opt(const sgk_batch_t *batch, sgk_stat_rec_t *out, void *workspace, size_t workspace_bytes, void *stream,
                 const sgk_stat_options_t *opt);
int sgk_stat_pa_opt(const sgk_batch_t *batch, sgk_stat_rec_t *out, float *pa_out, void *workspace,
                    size_t workspace_bytes, void *stream, const sgk_stat_options_t *opt);
int sgk_jnn_opt(const sgk_batch_t *batch, int rna, const uint64_t *seg_slots, int32_t *seg_x, int32_t *seg_y,
                uint32_t *n_segs, void *workspace, size_t workspace_bytes, void *stream, const sgk_stat_options_t *opt);
int sgk_prefix_opt(const sgk_batch_t *batch, int rna, int pore, sgk_prefix_rec_t *out, void *workspace,
                   size_t workspace_bytes, void *stream, const sgk_stat_options_t *opt);
/* ---- the pa -> event -> stat pipeline over one batch (BASELINE config 5; src/cfunc.c:72-83, 85-102, 126-159) -------- */
/* One call for what `pa`, `event` and `stat` print of the same reads: pa_out as sgk_pa, events as sgk_event_opt, stat_out as
 * sgk_stat_opt -- the fused stat + pA pass, then event on the raw samples; workspaces as for sgk_event_opt and
 * sgk_stat_pa_opt.  0.2.2. */
int sgk_pipeline(const sgk_batch_t *batch, int rna, const uint64_t *ev_slots, sgk_event_rec_t *events, uint32_t *n_events,
                 float *pa_out, sgk_stat_rec_t *stat_out, void *event_workspace, size_t event_workspace_bytes,
                 void *stat_workspace, size_t stat_workspace_bytes, void *stream, const sgk_event_options_t *event_opt,
                 const sgk_stat_options_t *stat_opt);

/* What the long-read path of the last stat / jnn / prefix call on `workspace` did (copied from the device: call it when
 * the stream has drained).  A long read's sequential float sums (src/stat.h:17-54, src/jnn.c:106-124, 195-199) are
 * composed from per-tile summaries; n_true_tiles of the n_tiles tile sums had to be evaluated from the true accumulator
 * instead (binade crossings, mispredicted binades).  All zero when the call had no long read or no room for them. */
/* What a stat (tool 0) / jnn (1) / prefix (2) / stat_pa (3) call would do with a batch of these totals (host arithmetic only, no GPU):
 * which implementation kernels = 0 resolves to, the long-read threshold it uses (0: no read of the batch can be long)
 * and how many long reads it accepts (under the per-batch threshold more than that many and none is treated as long;
 * under an explicit one the surplus runs on one wavefront each), the workspace sgk_*_workspace_bytes asks for. */
typedef struct sgk_stat_plan {
    uint32_t kernels;         /* 1: one read per lane, 2: one read per wavefront */
    uint32_t long_min;
    uint32_t long_max_reads;
    uint32_t reserved;
    uint64_t workspace_bytes;
} sgk_stat_plan_t;
int sgk_stat_plan(int tool, uint32_t n_reads, uint64_t n_samples, uint32_t max_read_len, const sgk_stat_options_t *opt,
                  sgk_stat_plan_t *out);
/* The table behind kernels = 0, one row per tool (0 stat, 1 jnn, 3 stat + pA, 4 the region statistics of prefix): a batch
 * of similar read lengths takes the lane-per-read kernels iff n_reads >= min_reads and its longest read has at most
 * min(cap, slope_x1024 * n_reads / 1024 + intercept) samples.  Returns the number of rows; fills at most cap of them.
 * (0.2.2; the guard test times both implementations either side of that line.) */
typedef struct sgk_stat_lane_rule {
    uint32_t tool, min_reads, slope_x1024;
    int32_t intercept;
    uint32_t cap, reserved;
} sgk_stat_lane_rule_t;
int sgk_stat_lane_rules(sgk_stat_lane_rule_t *out, int cap);
typedef struct sgk_long_status {
    uint32_t n_long_reads, n_tiles, n_true_tiles;
    uint32_t n_timeouts;  /* long reads DECLINED by the long path: a workgroup of the read gave up at a barrier (after
                           * seconds: never on a GPU of its own) or was told that another one had.  Nothing of such a read
                           * is written by the long path; the wave-per-read kernel redoes it on one wavefront behind the
                           * join, so the results are right either way (0.2.1 left them wrong and only counted here);
                           * jobs report the count through sgk_job_long_declined() */
} sgk_long_status_t;
int sgk_stat_long_status(const void *workspace, size_t workspace_bytes, uint32_t n_reads, sgk_long_status_t *out);

/* ---- ent: the histograms behind `sigtk ent` (src/ent.c) ----------------- */
/* The counting runs on the GPU, the sum of -p*log2(p) over the (few thousand) non-empty bins on the host
 * with the reference's own operations and order (sgk_ent_finish), so the printed entropies are identical.
 * Values below the windows are counted in the record; the rare others (raw < 0 or >= 8192, zigzag delta
 * >= 4096) are listed, unordered, in over_raw / over_delta at [offsets[r], offsets[r] + n_over_*). */
#define SGK_ENT_RAW_WINDOW 8192
#define SGK_ENT_DELTA_WINDOW 4096
typedef struct sgk_ent_hist {
    uint32_t n;            /* samples of the read */
    uint32_t n_over_raw;   /* entries of this read in over_raw */
    uint32_t n_over_delta; /* entries of this read in over_delta */
    uint32_t reserved;
    uint32_t raw[SGK_ENT_RAW_WINDOW];     /* count of samples with (uint16)raw == v, all n samples */
    uint32_t delta[SGK_ENT_DELTA_WINDOW]; /* count of (uint16)zigzag(raw[i]-raw[i-1]) == v, i = 0..n-2, raw[-1] = 0 */
    uint32_t hi[256], lo[256];            /* byte planes of those n-1 values */
} sgk_ent_hist_t;
/* over_raw / over_delta: device arrays of batch->n_samples uint16 each (same layout as the samples) */
int sgk_ent(const sgk_batch_t *batch, sgk_ent_hist_t *out, uint16_t *over_raw, uint16_t *over_delta, void *stream);
/* HOST function: out[0..2] = raw, delta, byte entropy of one read from host copies of its record and of its
 * two overflow lists (which it sorts in place; may be NULL when the counts are 0) */
void sgk_ent_finish(const sgk_ent_hist_t *hist, uint16_t *over_raw, uint16_t *over_delta, double *out);

/* ---- svb-zd signal decode on the device ------------------------------- */
/* Expands BLOW5 svb-zd signal blobs (slow5lib/src/slow5_press.c:1116-1146: u32 count, streamvbyte
 * keys + data of the zigzag deltas) into int16 samples, one wavefront per read.  The record layer
 * (zlib) stays on the host; files written with signal compression svb-zd hand the blob over as it
 * sits in the record.  blobs: device buffer, readable up to its size rounded up to a multiple of 4
 * bytes (the kernel uses aligned 4-byte loads); blob_offsets/blob_lengths: byte offset and byte length
 * of each read's blob (the length INCLUDES the 4-byte count); samples/offsets/lengths as in
 * sgk_batch_t (lengths[r] must equal the blob's count); status[r]: 0 ok, 1 count mismatch,
 * 2 truncated / inconsistent blob (the read's samples are then undefined). */
int sgk_svbzd_decode(const uint8_t *blobs, const uint64_t *blob_offsets, const uint32_t *blob_lengths,
                     uint32_t n_reads, int16_t *samples, const uint64_t *offsets, const uint32_t *lengths,
                     uint32_t *status, void *stream);

/* ---- zlib record inflate on the device (round 5; SURVEY 8f-1 / 8f-2 taken to the record layer) ------------------ */
/* BLOW5 files compress every record as one zlib stream (slow5lib/src/slow5.c:2583-2598); the reference inflates them one
 * at a time on its one thread, and a pool of host threads is what bounded the drop-in CLI's rate.  sgk_inflate inflates n
 * zlib streams (RFC 1950: header, DEFLATE blocks of every type, Adler-32), one wavefront per stream.  in: device buffer
 * readable up to its size rounded up to a multiple of 4 bytes; in_offsets / in_lengths: byte offset and length of every
 * stream; out: 16-byte aligned device buffer, stream r's bytes go to out + out_offsets[r] (offsets multiples of 16), which
 * must have room for out_caps[r] bytes >= what the stream inflates to (matches that reach far back read the stream's own
 * earlier bytes from there); out_lengths[r]: what the stream inflated to; status[r]: 0 ok, 1 bad zlib header / preset
 * dictionary, 2 bad block type / stored length, 3 bad code lengths, 4 invalid code, 5 distance in front of the stream,
 * 6 truncated input, 7 Adler-32 mismatch, 8 the stream inflates to more than out_caps[r] (1 - 8: the bytes are undefined). */
int sgk_inflate(const uint8_t *in, const uint64_t *in_offsets, const uint32_t *in_lengths, uint32_t n, uint8_t *out,
                const uint64_t *out_offsets, const uint32_t *out_caps, uint32_t *out_lengths, uint32_t *status,
                void *stream);

/* ---- qts: quantise the raw signal (src/qts.c:27-43, :126-142) and re-encode it ------ */
#define SGK_QTS_FLOOR 0     /* (raw >> b) << b                                       */
#define SGK_QTS_ROUND 1     /* round_to_power_of_2(raw, b): the reference's default   */
#define SGK_QTS_FILL_ONES 2 /* raw | ((1 << b) - 1)                                   */
/* in place on the samples of every read; bits in [1, 15] */
int sgk_qts(int16_t *samples, const uint64_t *offsets, const uint32_t *lengths, uint32_t n_reads,
            uint32_t max_read_len, int bits, int method, void *stream);
/* svb-zd ENCODE (slow5lib/src/slow5_press.c:1063-1089): canonical streamvbyte, so the blobs equal slow5lib's byte
 * for byte.  Two steps: sgk_svbzd_size fills blob_lengths[r] (4 + key bytes + data bytes); the caller lays the blobs
 * out (blob_offsets[r], 4-byte aligned) and sgk_svbzd_encode writes them. */
int sgk_svbzd_size(const int16_t *samples, const uint64_t *offsets, const uint32_t *lengths, uint32_t n_reads,
                   uint32_t *blob_lengths, void *stream);
int sgk_svbzd_encode(const int16_t *samples, const uint64_t *offsets, const uint32_t *lengths, uint32_t n_reads,
                     uint8_t *blobs, const uint64_t *blob_offsets, const uint32_t *blob_lengths, void *stream);

/* ---- synthetic reads (BASELINE configs 2-5) ---------------------------- */
/* Deterministic counter-based generator, identical on host and device (integer only).
 * kind 0: DNA-like (mean dwell 9 samples); kind 1: RNA-like (mean dwell 36, adaptor +
 * polyA prefix structure).  Read r of the batch is global read index first_read + r. */
int sgk_synth_reads(int16_t *samples, const uint64_t *offsets, const uint32_t *lengths, double *digitisation,
                    double *offset, double *range, uint32_t n_reads, uint32_t max_read_len,
                    uint64_t first_read, uint64_t seed, int kind, void *stream); /* device pointers */
void sgk_synth_reads_host(int16_t *samples, const uint64_t *offsets, const uint32_t *lengths,
                          double *digitisation, double *offset, double *range, uint32_t n_reads,
                          uint64_t first_read, uint64_t seed, int kind);

/* ---- per-launch timing (HIP events on the caller's stream) ------------------------ */
/* When enabled, every device-API call records hipEvents around each kernel it launches;
 * sgk_profile_read() synchronises and returns the accumulated milliseconds per kernel
 * name since the last reset.  Used by bench.py for the roofline figure. */
void sgk_profile_enable(int on);
void sgk_profile_reset(void);
/* names/ms/calls: caller arrays of length cap; returns the number of distinct kernels */
int sgk_profile_read(const char **names, double *ms, uint32_t *calls, int cap);

/* ================================ Host API ======================================== */

typedef struct sgk_host_batch {
    const int16_t *samples;   /* host, packed; no alignment/padding requirement */
    const uint64_t *offsets;  /* host, n_reads+1 */
    const double *digitisation, *offset, *range; /* host, n_reads */
    uint32_t n_reads;
} sgk_host_batch_t;

typedef struct sgk_events_host {
    uint32_t n_reads;
    uint64_t *ev_offsets; /* n_reads+1, compact CSR: events of read r are [ev_offsets[r], ev_offsets[r+1]) */
    uint32_t *start, *length;
    float *mean, *stdv;
    sgk_event_status_t status;
} sgk_events_host_t;
int sgk_event_host(const sgk_host_batch_t *batch, int rna, sgk_events_host_t *out);
int sgk_event_host_opt(const sgk_host_batch_t *batch, int rna, sgk_events_host_t *out, const sgk_event_options_t *opt);
void sgk_events_host_free(sgk_events_host_t *ev);

int sgk_pa_host(const sgk_host_batch_t *batch, float *pa_out /* host, n_samples */);
int sgk_stat_host(const sgk_host_batch_t *batch, sgk_stat_rec_t *out /* host, n_reads */);
int sgk_stat_host_opt(const sgk_host_batch_t *batch, sgk_stat_rec_t *out, const sgk_stat_options_t *opt);

typedef struct sgk_segs_host {
    uint32_t n_reads;
    uint64_t *seg_offsets; /* n_reads+1, compact CSR */
    int32_t *x, *y;
} sgk_segs_host_t;
int sgk_jnn_host(const sgk_host_batch_t *batch, int rna, sgk_segs_host_t *out);
int sgk_jnn_host_opt(const sgk_host_batch_t *batch, int rna, sgk_segs_host_t *out, const sgk_stat_options_t *opt);
void sgk_segs_host_free(sgk_segs_host_t *s);

int sgk_prefix_host(const sgk_host_batch_t *batch, int rna, int pore, sgk_prefix_rec_t *out /* host, n_reads */);
int sgk_prefix_host_opt(const sgk_host_batch_t *batch, int rna, int pore, sgk_prefix_rec_t *out,
                        const sgk_stat_options_t *opt);

/* ---- pipelined host jobs (the batched, overlapped replacement of the ------
 *      reference's one-record-at-a-time loop, src/cmain.c:118-120) ---------------------
 * A job owns pinned host staging, device buffers, a private stream and pinned result buffers
 * for one batch; all of them only grow, so recycled jobs stop allocating.  Typical use, with
 * several jobs in flight so that reading/inflating, PCIe, kernels and formatting overlap:
 *     sgk_job_begin(job, n, lengths, SGK_SIGNAL_SVBZD, blob_bytes, &in);
 *     ... reader threads copy blob r to in.blobs + in.blob_offsets[r], fill in.digitisation[r] ...
 *     sgk_job_submit(job, SGK_TOOL_EVENT, rna, pore, 0);       // returns at once
 *     ... (another thread) sgk_job_wait(job); sgk_job_output(job, &out); format rows ...
 * begin/submit/wait may be called from different threads, one at a time per job.  After a call on a job has
 * returned an error other than SGK_ERR_FORMAT / SGK_ERR_CAPACITY (which describe the data, the job stays usable),
 * destroy the job: work may still be queued on its stream. */
typedef struct sgk_job sgk_job_t;

#define SGK_TOOL_PA 0
#define SGK_TOOL_EVENT 1
#define SGK_TOOL_STAT 2
#define SGK_TOOL_JNN 3
#define SGK_TOOL_PREFIX 4
#define SGK_TOOL_ENT 5
#define SGK_TOOL_QTS 6 /* submitted with sgk_job_submit_qts */

#define SGK_SIGNAL_INT16 0 /* caller stages decoded int16 samples                        */
#define SGK_SIGNAL_SVBZD 1 /* caller stages svb-zd blobs; decoded on the GPU (8f-1)      */
#define SGK_SIGNAL_ZREC 2  /* caller stages whole zlib-compressed BLOW5 records (svb-zd signal): inflated and decoded on the
                            * GPU (sgk_job_begin_zrec) */

#define SGK_JOB_EVENTS_COMPACT 1 /* submit flag: only event start/length are copied back (event -c) */
#define SGK_JOB_EVENTS_LENGTHS 2 /* submit flag (0.2.3): only the event lengths are copied back -- ev_start, ev_mean and
                                  * ev_stdv are NULL.  The events of a read are contiguous from sample 0
                                  * (src/events.c:491-501), so start_i is the sum of the lengths in front of event i: half
                                  * of SGK_JOB_EVENTS_COMPACT's bytes over PCIe, which is what `event -c` waits for */

typedef struct sgk_job_input {
    int16_t *samples;             /* pinned host (SGK_SIGNAL_INT16): read r at samples + offsets[r] */
    uint8_t *blobs;               /* pinned host (SGK_SIGNAL_SVBZD): blob r at blobs + blob_offsets[r] */
    const uint64_t *offsets;      /* n_reads: sample offset of every read (also indexes the pa output) */
    const uint64_t *blob_offsets; /* n_reads */
    double *digitisation, *offset, *range; /* pinned host, n_reads: filled by the caller */
    uint64_t n_samples;           /* length of the sample arena */
} sgk_job_input_t;

typedef struct sgk_job_output {
    uint32_t n_reads;
    const uint64_t *offsets;        /* as in sgk_job_input_t */
    const uint32_t *lengths;
    const uint32_t *decode_status;  /* svb-zd input: per-read status of sgk_svbzd_decode, else NULL */
    const float *pa;                /* pa: pa[offsets[r] + i] */
    const uint64_t *slots;          /* event / jnn: arena slot of read r's first item (n_reads+1) */
    const uint32_t *counts;         /* event / jnn: items of read r */
    const uint32_t *ev_start, *ev_length;
    const float *ev_mean, *ev_stdv; /* NULL with SGK_JOB_EVENTS_COMPACT / _LENGTHS (_LENGTHS: ev_start as well) */
    const int32_t *seg_x, *seg_y;
    const sgk_stat_rec_t *stat;
    const sgk_prefix_rec_t *prefix;
    sgk_event_status_t event_status;
    /* qts: the quantised signal, as svb-zd blobs (blob r at qts_blobs + qts_blob_offsets[r], qts_blob_lengths[r]
     * bytes) or, for files without signal compression, as int16 samples at qts_samples + offsets[r] */
    const uint8_t *qts_blobs;
    const uint64_t *qts_blob_offsets;
    const uint32_t *qts_blob_lengths;
    const int16_t *qts_samples;
    const sgk_ent_hist_t *ent;            /* ent: one record per read */
    uint16_t *ent_over_raw, *ent_over_delta; /* ent: host copies of the overflow lists (NULL if all empty) */
} sgk_job_output_t;

int sgk_job_create(int device, sgk_job_t **out);
/* the options the job's submits use from now on (copied; null = defaults) */
int sgk_job_set_options(sgk_job_t *job, const sgk_event_options_t *event_opt, const sgk_stat_options_t *stat_opt);
void sgk_job_destroy(sgk_job_t *job);
int sgk_job_device(const sgk_job_t *job);
/* lengths[r]: samples of read r; blob_bytes[r] (SGK_SIGNAL_SVBZD only): byte length of its blob */
int sgk_job_begin(sgk_job_t *job, uint32_t n_reads, const uint32_t *lengths, int signal_format,
                  const uint32_t *blob_bytes, sgk_job_input_t *in);
/* The records as they sit in a BLOW5 file with zlib records and svb-zd signal (0.2.2): the caller stages record r's
 * rec_bytes[r] on-disk bytes at in->blobs + in->blob_offsets[r] and fills the scaling; the job inflates them on the GPU
 * (sgk_inflate) and decodes the signal blobs from there.  What the caller must know of a record it knows from its head
 * (a few hundred inflated bytes): lengths[r] samples, the signal blob at sig_offset[r] of the inflated record and
 * sig_bytes[r] long, the inflated record rec_room[r] bytes at most (head + signal + auxiliary fields; a record that
 * inflates to more fails with status 0x108).  sgk_job_wait returns SGK_ERR_FORMAT if a record did not inflate or decode
 * (decode_status[r]: 0x100 | sgk_inflate's status, or sgk_svbzd_decode's). */
int sgk_job_begin_zrec(sgk_job_t *job, uint32_t n_reads, const uint32_t *lengths, const uint32_t *rec_bytes,
                       const uint32_t *sig_offset, const uint32_t *sig_bytes, const uint32_t *rec_room,
                       sgk_job_input_t *in);
/* may be called again after sgk_job_wait to run another tool over the same staged batch */
int sgk_job_submit(sgk_job_t *job, int tool, int rna, int pore, int flags);
/* qts over the staged batch: quantise (bits in [1,15], method SGK_QTS_*), then hand the signal back as svb-zd blobs
 * (out_signal_format SGK_SIGNAL_SVBZD) or int16 samples (SGK_SIGNAL_INT16) */
int sgk_job_submit_qts(sgk_job_t *job, int bits, int method, int out_signal_format);
/* SGK_ERR_FORMAT if a blob did not decode, SGK_ERR_CAPACITY on event-slot overflow */
int sgk_job_wait(sgk_job_t *job);
/* valid after sgk_job_wait until the job's next sgk_job_begin */
int sgk_job_output(const sgk_job_t *job, sgk_job_output_t *out);
/* stat / jnn / prefix jobs, after sgk_job_wait: long reads the 16-workgroup path declined (a barrier wait timed out) and
 * the wave-per-read kernel redid -- sgk_long_status_t::n_timeouts of the job's last submit.  The records are right
 * either way; non-zero means the GPU did not dispatch a grid's workgroups the way the long path assumes (0.2.2). */
uint32_t sgk_job_long_declined(const sgk_job_t *job);

/* ---- per-read shims with the reference's own signatures (batch of one) ------------- */
/* event_t / event_table exactly as src/sigtk.h:55-70 */
typedef struct {
    uint64_t start;
    float length;
    float mean;
    float stdv;
} sgk_event_t;
typedef struct {
    size_t n;
    size_t start;
    size_t end;
    sgk_event_t *event; /* malloc'd; caller frees, as with getevents() (src/cfunc.c:82) */
} sgk_event_table;
/* float *signal_in_picoamps(slow5_rec_t*) (src/misc.c:15): malloc'd, caller frees */
float *sgk_signal_in_picoamps(const int16_t *raw, uint64_t len_raw_signal, double digitisation,
                              double offset, double range);
/* event_table getevents(size_t nsample, float *rawptr, int8_t rna) (src/events.c:553) */
sgk_event_table sgk_getevents(size_t nsample, float *rawptr, int8_t rna);

/* jnn_pair_t, jnn_param_t, jnnv2_param_t exactly as src/jnn.h:13-16, 18-27, 74-81 */
typedef struct {
    int64_t x;
    int64_t y;
} sgk_jnn_pair_t;
typedef struct {
    float std_scale;
    int corrector;
    int seg_dist;
    int window;
    float stall_len;
    int error;
    float top;
    float bot;
} sgk_jnn_param_t;
typedef struct {
    float std_scale;
    int seg_dist;
    int window; /* must be 2000 (both presets of the reference): the rolling mean divides by it exactly */
    float stall_len;
    int hi_thresh;
    int lo_thresh;
} sgk_jnnv2_param_t;
/* jnn_pair_t *jnn_raw(const int16_t*, int64_t, jnn_param_t, int *n) (src/jnn.c:282): malloc'd, caller frees; NULL with
 * *n = 0 for an empty read, as the reference */
sgk_jnn_pair_t *sgk_jnn_raw(const int16_t *raw, int64_t nsample, sgk_jnn_param_t param, int *n);
/* jnn_pair_t *jnn_pa(const float*, int64_t, jnn_param_t, int *n) (src/jnn.c:295) */
sgk_jnn_pair_t *sgk_jnn_pa(const float *raw, int64_t nsample, sgk_jnn_param_t param, int *n);
/* jnn_pair_t jnnv2(const int16_t*, int64_t, jnnv2_param_t) (src/jnn.c:99): {-1,-1} when nsample <= window */
sgk_jnn_pair_t sgk_jnnv2(const int16_t *sig, int64_t nsample, sgk_jnnv2_param_t param);
/* jnn_pair_t find_adaptor(slow5_rec_t*, int8_t pore) (src/jnn.c:181): the record's raw signal and its length */
sgk_jnn_pair_t sgk_find_adaptor(const int16_t *raw, int64_t nsample, int8_t pore);
/* jnn_pair_t find_polya(const float*, int64_t, float top, float bot, int8_t pore) (src/jnn.c:352) */
sgk_jnn_pair_t sgk_find_polya(const float *raw, int64_t nsample, float top, float bot, int8_t pore);
/* the six inlines of src/stat.h:17-73 (sequential float sums in sample order; medians = rank n/2) */
float sgk_meanf(const float *x, int n);
float sgk_meani16(const int16_t *x, int n);
float sgk_stdvf(const float *x, int n);
float sgk_stdvi16(const int16_t *x, int n);
float sgk_medianf(const float *x, int n);
int16_t sgk_mediani16(const int16_t *x, int n);
/* SGK_OK, or why the last shim call of this thread returned NULL / {-1,-1} / NaN */
int sgk_shim_status(void);

#ifdef __cplusplus
}
#endif
#endif /* SIGTK_GPU_H */
