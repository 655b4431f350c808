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
    uint32_t pad[9];
};
static_assert(sizeof(EvHeader) == 64, "header is one 64-byte block");

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
};

// workspace carving shared by sgk_event_workspace_bytes and sgk_event
struct EvWorkspace {
    size_t off_hdr, off_flags, off_list, off_order, off_bitmap, off_scratch, total;
    uint64_t scratch_stride;
    uint32_t n_fb_blocks;
};
EvWorkspace event_workspace_layout(uint32_t n_reads, uint64_t n_samples, uint32_t max_read_len,
                                   size_t available /* 0 = default sizing */);

int launch_event(const EvArgs &a, int rna, bool float_input, uint32_t n_fb_blocks, hipStream_t st);

}  // namespace sgk
