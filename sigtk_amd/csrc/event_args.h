// event_args.h -- argument block and workspace header of the event kernels.
#pragma once
#include "sgk_common.h"

namespace sgk {

struct EvHeader {
    uint32_t n_flagged;   // reads that failed the exactness guard
    uint32_t n_rerun;     // speculative chunks that were re-run
    uint32_t n_overflow;  // reads whose events did not fit their slot range
    uint32_t fb_next;     // work counter of the persistent fallback kernel
    unsigned long long n_events_total;
    uint32_t n_hot_runs;  // lanes that replayed long-detector runs exactly (lazy long detector)
    uint32_t n_long;      // reads taken by several wavefronts (segments), see SegDesc
    uint32_t n_segs;      // their segments
    uint32_t n_seam_rerun;  // segments whose speculative start was wrong and that were run again from the true state
    unsigned long long n_replay_idx;  // indices the exact replay of the long detector walked (one lane each)
    uint32_t pad[4];
};
static_assert(sizeof(EvHeader) == 64, "header is one 64-byte block");

// ---- reads longer than one wavefront should take (round 3) ----------------------------------------------------
// A read of at least `long_min` samples is cut into segments of `seg_len` samples (a multiple of 1024), one wavefront
// each: the same speculative scheme that lets the 64 lanes of a wave start in the middle of a read lets a wave do
// so.  Segment g > 0 starts its first lane `lead` samples early from the fresh state; the state it reaches at the
// segment's first index must equal the state segment g-1 ended with (k_event_seam checks, and runs a segment whose
// speculation failed again from the true state).  The builder then runs per segment as well, from the last boundary
// in front of the segment, at the event rank the segments in front of it determine.
struct LzSnapState {  // detector state at an index, absolute (read-relative) positions: what chunks and segments hand over
    int sp;         // short peak_pos, -1 when not in a peak
    float sv;       // short peak_value
    int lm;         // long masked_to (= short peak_pos at the last reset + W1), LZ_NONE when it no longer masks
    int r0;         // first index of the long detector's current run: behind its last reset and behind the mask that
                    // reset set (pass start of a speculative pass)
    uint32_t bits;  // 1: in a peak, 2: valid, 4: strong, 8: hot (long run since r0 needs the exact replay)
};
struct LzRun {
    int a, b;  // exact replay of the long detector over [a, b) from the fresh state
};
constexpr int SEG_CROSS_MAX = 23;
constexpr int SEG_PRE_MAX = 15;
constexpr uint32_t SEG_NONE = 0xffffffffu;
struct SegDesc {
    uint32_t read;   // read index in the batch (SEG_NONE: an entry nobody owns, see k_seg_plan)
    uint32_t g;      // segment index within the read
    uint32_t lread;  // index into the long-read list
    uint32_t pad;
};
struct SegState {           // 320 bytes per segment
    LzSnapState init0;      // state the speculative first lane reached at the segment's first index (g > 0)
    LzSnapState end;        // state at the segment's end (final once `stage` is set)
    int status;             // detect_span's return code (non-zero: the read goes to the exact fallback)
    uint32_t n_cross;       // hot long-detector runs that begin in front of the segment
    LzRun cross[SEG_CROSS_MAX];
    uint32_t n_pre;         // peaks in front of the segment that were pending at its first index and emitted inside it
    int pre[SEG_PRE_MAX];   // (their positions): boundaries this segment owns although they lie in front of it
    // the chain (round 4, chain_segment): what the segment behind needs, published with agent-scope atomic stores, `stage`
    // last.  The wave of segment g + 1 polls `stage`, compares `end` with its own init0, and takes its event rank and
    // the boundary its first event starts at from here -- every dependency points at a LOWER workgroup index.
    uint32_t stage;         // 0: not yet, 1: the fields below and `end` are final
    uint32_t cum_cnt;       // boundaries owned by the segments up to and including this one
    int last_pos;           // the last of them (-1: none so far)
    uint32_t cflags;        // 1: this or an earlier segment declined the read (it goes to the exact fallback)
    // builder outputs
    uint32_t ext_lo, ext_hi;  // extremes of the samples walked: raw int16 (as int) or float bit patterns
    uint32_t bflags;          // 1: a tile with more boundaries than the builder records, 2: event slots overflowed
    uint32_t pad;
};
static_assert(sizeof(SegState) == 328, "SegState layout");
struct LongRead {
    uint32_t read, seg0, nseg;
    uint32_t seg_len;   // long reads: EvArgs::seg_len; reads of the tail split: EvArgs::split_seg
    // accumulated by the segments' waves (device-scope atomics); the wave that finishes last gives the verdict
    uint32_t ext_lo, ext_hi;  // extremes over all segments (int16 input: as signed ints)
    uint32_t flags;           // 1: declined / dense tile / failed seam chain, 2: slot overflow
    uint32_t built;           // segments done
};

struct EvArgs {
    const void *samples;            // int16 or float, packed
    const uint64_t *offsets;
    const uint32_t *lengths;
    const double *dig, *off, *rng;  // null for float (pA) input
    uint32_t n_reads;
    uint64_t n_alloc;               // readable samples: the batch's n_samples
    const uint64_t *ev_slots;
    sgk_event_rec_t *events;        // 16-byte records; slots of read r: [ev_slots[r], ev_slots[r+1])
    uint32_t *n_events;
    // workspace
    EvHeader *hdr;
    uint8_t *flags;                 // n_reads
    uint32_t *flag_list;            // n_reads
    unsigned long long *bitmap;     // word base of read r: offsets[r]/64 + r
    double *scratch;                // fallback: per block scratch_stride doubles
    uint64_t scratch_stride;        // 2 * (max_read_len + 1), rounded up to even
    uint32_t *order;                // n_reads + 128: workgroup i of k_event takes read order[i] (longest first)
    // long reads (max_segs == 0: none in this batch, nothing below is used)
    uint32_t max_segs, max_long;    // capacities, from the batch's totals (event_seg_capacity)
    uint32_t seg_len, long_min;
    int lead_override;              // > 0: speculative warm-up in samples (tests: a short one makes speculation fail)
    SegDesc *segs;
    SegState *seg_state;
    LongRead *longs;
    // short reads (multi_lanes == 0: every read has a wavefront of its own)
    uint32_t multi_lanes;           // lanes per short read (a power of two below 64): k_event_multi packs 64 / lanes reads
    uint32_t multi_max;             // reads shorter than this (a power of two) are short
    // tail split (k_seg_plan): reads at dispatch positions >= split_from are cut into split_seg-sample segments
    uint32_t split_from, split_seg;  // (split_from >= n_reads: none)
    uint32_t has_long;               // the batch may hold reads of long_min samples or more
    uint32_t dev;                    // development builds only (-DSGK_DEV, tools/build_variant.sh): SGK_DEV_* below; else 0
};
// What a development build (never the shipped library) can switch off or record per call, from
// sgk_event_options_t::reserved[0]: the ablation behind profiles/r05_event_instruction_table.md and the per-wave
// timestamps behind profiles/r05_event_first_round.md.
constexpr uint32_t SGK_DEV_NO_BUILD = 1;     // k_event: detector + bitmap only
constexpr uint32_t SGK_DEV_NO_DETECT = 2;    // ... builder only (the bitmap of the previous call on this workspace)
constexpr uint32_t SGK_DEV_NO_ROUNDS = 4;    // builder: the sample walk only, no event rounds
constexpr uint32_t SGK_DEV_RAW_EVENTS = 8;   // builder: event rounds without create_event's arithmetic (sums stored as they are)
constexpr uint32_t SGK_DEV_TRACE = 16;       // k_event: s_memrealtime at wave start / detector end / builder end + HW_ID, XCC_ID
                                             // per workgroup, 32 bytes each, into the fallback scratch (unused when no read is flagged)


struct EvSegConfig {
    uint32_t dev;        // SGK_DEV builds: sgk_event_options_t::reserved[0]
    uint32_t seg_len, long_min;
    int lead_override;
    int multi;  // lanes per short read: 0 = chosen per batch, -1 = off (64 lanes per read), 1 .. 32 = forced
    uint32_t multi_max;  // 0 = default; reads shorter than this (a power of two) count as short
    int tail_split;      // 0 = chosen per batch, -1 = off
    bool auto_geometry;  // neither segment_len nor long_min was given: event_config_for picks them per batch
};
EvSegConfig event_config(const sgk_event_options_t *opt);  // null: the defaults
// the configuration with the long reads' geometry resolved for a batch of these totals and this preset
EvSegConfig event_config_for(const EvSegConfig &c, uint64_t n_samples, int rna);
void event_multi_plan(const EvSegConfig &c, uint32_t n_reads, uint64_t n_samples, uint32_t max_read_len, int rna,
                      bool sorted, uint32_t &multi_lanes, uint32_t &multi_max);
void event_tail_plan(const EvSegConfig &c, uint32_t n_reads, uint64_t n_samples, uint32_t max_read_len, int rna,
                     bool packed, uint32_t &split_from, uint32_t &split_seg);
void event_seg_capacity(const EvSegConfig &c, uint32_t n_reads, uint64_t n_samples, uint32_t max_read_len, int rna,
                        bool packed, uint32_t &max_segs, uint32_t &max_long);

// workspace carving shared by sgk_event_workspace_bytes and sgk_event
struct EvWorkspace {
    size_t off_hdr, off_flags, off_list, off_order, off_bitmap, off_segs, off_seg_state, off_longs, off_scratch, total;
    uint32_t max_segs, max_long;
    uint64_t scratch_stride;
    uint32_t n_fb_blocks;
};
EvWorkspace event_workspace_layout(const EvSegConfig &c, uint32_t n_reads, uint64_t n_samples, uint32_t max_read_len,
                                   size_t available /* 0 = default sizing */);

int launch_event(const EvArgs &a, int rna, bool float_input, uint32_t n_fb_blocks, hipStream_t st);

}  // namespace sgk
