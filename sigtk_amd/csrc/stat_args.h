// stat_args.h -- argument block of the stat / jnn / prefix kernels.
#pragma once
#include "sgk_common.h"

namespace sgk {

enum { REG_WHOLE = 0, REG_ADAPT = 1, REG_POLYA = 2, REG_TAIL = 3 };

struct StatArgs {
    sgk_batch_t b;               // device pointers
    sgk_stat_rec_t *stat;        // stat output (or null)
    sgk_prefix_rec_t *prefix;    // prefix output / region source (or null)
    const uint64_t *seg_slots;   // jnn arena
    int32_t *seg_x, *seg_y;
    uint32_t *n_segs;
    uint32_t *err_count;         // workspace: number of reads whose segments overflowed their slots
    float *pa_out;               // stat+pa fused: pA of every sample, written by the median pass (or null)
    const uint32_t *order;       // wave-per-read kernels: wave i takes read order[i] (longest reads first), or null
    uint32_t jnn_redo;           // k_jnn: 1 = only the reads k_jnn_wave gave up on (n_segs[r] == JNN_REDO_MARK)
    int kernels;                 // sgk_stat_options_t::kernels: 0 chosen per batch, 1 read per lane, 2 read per wavefront
    // long reads (k_long_chains): null / 0 when the batch has none or the workspace no room for them
    struct LongHdr *long_hdr;
    uint32_t *long_list;         // LC_CAP read indices
    struct LongSums *longs;      // LC_CAP records, entry i belongs to read long_list[i]
    struct LongWork *long_work;  // LC_CAP blocks the workgroups of a long read meet in
    unsigned long long *long_pool;  // tile records: [0, long_pool_tiles) of the first sum, then of the second
    uint32_t *long_hist;         // stat: LC_CAP window histograms of LC_HIST_BINS bins
    uint32_t long_pool_tiles;
    uint32_t long_min;           // reads of at least this many samples are long
    uint32_t long_redo;          // wave kernels: 1 = this launch only takes the long reads k_long_chains DECLINED (a barrier
                                 // of theirs timed out): wave i looks at long_list[i], everything else returns at once
    uint32_t long_fault;         // sgk_stat_options_t::debug_fault (tests): see lc_barrier
};
constexpr uint32_t JNN_REDO_MARK = 0xffffffffu;

// ---- long reads (round 4): the sequential float sums of a read of long_min samples or more are evaluated by
// LC_PARTS workgroups of four wavefronts BEFORE the wave-per-read kernel runs (k_long_chains); that kernel then finds the
// sums here and walks the read only for what is cheap per sample (histogram, pA output, automaton, run finder).
constexpr int LC_PARTS = 16;                  // workgroups per long read
constexpr int LC_WG_WAVES = 4;
constexpr int LC_WAVES = LC_PARTS * LC_WG_WAVES;
static_assert(LC_WAVES == 64, "lc_stage reads one wave's total per lane");
struct LongHdr {
    uint32_t n_long;     // long reads found (k_long_list; entries beyond LC_CAP have no record)
    uint32_t n_tiles;    // tile sums summarised
    uint32_t n_true;     // ... of which the composition had to evaluate from the true accumulator
    uint32_t pool_used;  // tile records handed out
    uint32_t n_timeout;  // barrier waits given up after seconds (never, on a GPU that dispatches a grid's workgroups in
                         // order): the read is DECLINED -- nothing of it is written -- and redone on one wavefront
    uint32_t n_declined; // reads declined that way (counted by the redo launch that takes them)
    uint32_t pad[10];
};
struct LongSums {
    uint32_t read;
    uint32_t valid;      // set by k_long_chains when it is done with the read (1: nothing to do for it, 2: the subtool's
                         // whole output is written); WHO does a read is decided by rec_off, see find_long
    float s1[2];         // first-stage sums (stat: raw, pA; jnn: clamped raw; prefix: rolling means), signed
    float s2[2];         // second-stage sums (squared deviations from the first stage's means)
    uint32_t rec_off;    // the read's tile records in the pool (LC_NO_REC: none, the read runs on one wave)
    uint32_t pad;
};
struct LongWork {        // what the workgroups of one long read exchange (agent-scope atomics only)
    uint32_t arrive;     // barrier counter
    uint32_t n_true;
    float m[2];          // the sums' accumulators after level 2 (oriented)
    uint32_t failed;     // a workgroup of the read gave up at a barrier: every workgroup leaves the read, its output is
                         // whatever the redo launch of the wave kernel writes (lc_barrier)
    uint32_t pad;
    unsigned long long seg_tot[LC_WAVES][2];  // pass A: sum of the terms of a wave's tiles (a double's bits)
};
static_assert(sizeof(LongSums) == 32 && sizeof(LongHdr) == 64 && sizeof(LongWork) == 24 + 16 * LC_WAVES, "long-read workspace layout");
constexpr uint32_t LC_CAP = 512;              // long reads per batch that get a record (the rest run as before)
constexpr uint32_t LC_NO_REC = 0xffffffffu;     // LongSums::rec_off of a long read without tile records
constexpr uint32_t LC_HIST_BINS = 2048;       // stat's window histogram (WH_BINS)
constexpr uint32_t LC_POOL_TILES = 1u << 20;  // tile records per sum (2^30 samples of long reads; 16 MB)
constexpr uint32_t LC_LONG_MIN = 262144;      // default long_min
constexpr uint32_t LC_AUTO_MAX_READS = 128;   // with the per-batch threshold: more long reads than this and none is treated as long
constexpr uint32_t LC_LONG_MIN_FLOOR = 8192;  // smallest long_min an option can ask for
size_t long_workspace_bytes(uint64_t n_samples, uint32_t max_read_len);
// The threshold a call uses: the option if positive, else per batch and tool
//     max(n_samples / div, clamp(n_samples / floor_div, floor_lo, 262 144))       (floor_div 0: floor_lo as it is)
// -- a read is long when its one wavefront would outlast the rest of the batch (the first term, large batches), and never
// under what the long path's barriers cost (the second: 262 144 samples in a batch of 3 x 10^8 samples, less in a smaller
// one, where fewer reads hide a long one.  Round 5, measured with 4 - 8 reads of 150 000 - 800 000 samples among 1 000 /
// 3 000 / 10 000 / 20 000 of 100 000: with the floor at 262 144 everywhere a batch of 1 000 reads with four of 200 000
// took 0.49 / 0.65 / 0.48 ms (stat / jnn / prefix) against 0.33 / 0.41 / 0.42 with those four on the long path, while
// 3 000 reads with eight of 150 000 are better off without it for stat and prefix (0.50 against 0.62, 0.57 against 0.59).
// tests/test_gpu_stat_long.py::test_long_read_threshold_is_no_cliff times both choices either side of the threshold.)
struct LongRule {
    uint32_t div, floor_lo, floor_div;
};
uint32_t long_threshold(uint64_t n_samples, int32_t opt_long_min, LongRule rule);
constexpr LongRule LC_AUTO_DIV_STAT = {2048, 131072, 1024}, LC_AUTO_DIV_JNN = {3072, 131072, 0},
                   LC_AUTO_DIV_PREFIX = {2048, 196608, 512};
// Which of the two implementations a batch of SIMILAR read lengths (the longest at most 1.5 x the mean) takes when the
// caller leaves the choice to the library.  The lane-per-read kernels have 64 reads per wavefront and none of the wave
// kernels' per-read costs, so they win where there are MANY reads for their length; measured over a grid of batch shapes
// (profiles/r05_lane_vs_wave_sweep.jsonl: 120 shapes, both implementations of every tool) the line on which the two
// cost the same is close to straight in (reads, samples per read).  One row per tool:
//     one read per lane  iff  n_reads >= min_reads  and  max_read_len <= min(cap, slope_x1024 * n_reads / 1024 + intercept)
// (min_reads: where that line reaches ~500 samples -- below it a batch is a few microseconds either way)
// One table: sgk_stat_plan, the launchers and the guard test (tests/test_gpu_stat.py::test_each_cut_of_the_choice_is_no_cliff,
// which times both implementations either side of the line at several batch sizes) read the same numbers.
// Tool: 0 stat, 1 jnn, 2 prefix' finders (no row: the wave finders win at every shape), 3 stat + pA, 4 the statistics of
// the regions prefix finds.  (Round 4 had five hand-placed steps instead; the guard test's first run found a batch
// 64 reads under one of them on the 30 % slower implementation.)
struct LaneRule {
    int tool;
    uint32_t min_reads;
    uint32_t slope_x1024;   // samples per read, per read of the batch, x 1024
    int32_t intercept;
    uint32_t cap;
};
constexpr LaneRule LANE_RULES[] = {
    {0, 1024u, 0u, 2048, 2048u},             // stat: tiny reads (the wave kernels' per-read costs: native heads, tile set-up) from 1 024 reads on
    {0, 5248u, 1178u, -5500, 131072u},       // stat: 16 384 reads of up to 13 300 samples, 32 768 of 32 200, 65 536 of 69 800
    {0, 81920u, 0u, 131072, 131072u},        // ... and from 81 920 reads on, where the medians come out of k_moments' second pass
                                             // (a real step in that implementation: 82 000 x 100 000 8.0 against 9.2 ms), up to 131 072
    {3, 2048u, 768u, -1000, 49152u},         // stat + pA (the wave kernel's fused pass carries the pA stores better)
    {1, 4096u, 133u, 0, 15000u},             // jnn: 65 536 reads of up to 8 500 samples
    {4, 49152u, 0u, 0x7fffffff, 0x7fffffffu},  // prefix: the statistics of the (short) regions the finders return, whatever the reads' length
};
constexpr int N_LANE_RULES = (int)(sizeof(LANE_RULES) / sizeof(LANE_RULES[0]));
constexpr uint32_t STAT_MOMENTS_MEDIAN_MIN_READS = LANE_RULES[2].min_reads;   // lane path of plain stat: from here on the medians come out of k_moments' second pass
static_assert(LANE_RULES[2].tool == 0 && LANE_RULES[2].slope_x1024 == 0u && LANE_RULES[2].min_reads == 81920u, "the step row of plain stat");
inline uint32_t lane_rule_max_len(const LaneRule &q, uint32_t n_reads) {
    const long long v = (long long)q.slope_x1024 * n_reads / 1024 + q.intercept;
    return v <= 0 ? 0u : (v > (long long)q.cap ? q.cap : (uint32_t)v);
}
bool stat_lane_per_read(int tool /* 0 stat, 1 jnn, 2 prefix, 3 stat + pA */, int kernels, uint32_t n_reads, uint64_t n_samples, uint32_t max_read_len);
// fills a.long_* from the workspace behind the dispatch order (when the batch has a long read and there is room),
// clears the header and lists the long reads; rule: the tool's threshold when the option is 0 (long_threshold)
int prepare_long(StatArgs &a, void *ws, size_t ws_bytes, int32_t opt_long_min, LongRule rule, hipStream_t st);

// workspace layout of stat / jnn / prefix: [0, 64) counters (jnn: overflow count), then the dispatch order of the
// wave-per-read kernels (n_reads x 4 bytes) and the 2 x 128 words of its counting sort
constexpr uint32_t ORDER_MIN_READS = 1024;  // below that a batch is one round of waves anyway
size_t order_workspace_bytes(uint32_t n_reads);
// fills a.order from the workspace when it is large enough (else leaves it null) and launches the sort
int prepare_order(StatArgs &a, void *ws, size_t ws_bytes, hipStream_t st);
// the sort itself: order[0 .. n) = read indices, longest reads first; scratch: 128 words
int launch_order(const uint32_t *lengths, uint32_t n, uint32_t *order, uint32_t *scratch128, hipStream_t st);

// jnn_param_t (src/jnn.h:18-27) and the run-finder part of jnnv2_param_t (src/jnn.h:74-81; its window is fixed at
// 2000, the value of both presets: the rolling mean divides by it with the exact constant division)
struct JnnP {
    float std_scale;
    int corrector, seg_dist, window;
    float stall_len;
    int error;
    float top, bot;
};
struct AdaptP {
    float std_scale;
    int seg_dist, lo_thresh, hi_thresh;
};
JnnP jnn_preset(int rna);          // jnn_print's presets (src/jnn.c:313-319)
JnnP jnn_polya_preset();           // JNNV1_R9_POLYA == JNNV1_RNA004_POLYA (src/jnn.h:52-72)
AdaptP adaptor_preset(int pore);   // find_adaptor's presets (src/jnn.c:181-188)

int check_batch(const sgk_batch_t *b);
int launch_pa(const sgk_batch_t *b, float *out, hipStream_t st);
int launch_stat(const StatArgs &a, hipStream_t st);  // a.pa_out != null: fused stat + pa
int launch_jnn(const StatArgs &a, const JnnP &p, hipStream_t st);
int launch_prefix(const StatArgs &a, int rna, int pore, hipStream_t st);
int launch_adaptor(const StatArgs &a, const AdaptP &p, hipStream_t st);  // jnnv2 only: prefix[r].adapt_x / adapt_y
// reference-signature shims on float input (one array per call): sequential float moments + order statistic, and
// jnn_core over rm_outlierf(x); device pointers
int launch_stat_f32(const float *x, int n, float *out3, hipStream_t st);
int launch_jnn_f32(const float *x, int64_t n, const JnnP &p, int32_t *seg_x, int32_t *seg_y, uint32_t cap,
                   uint32_t *n_segs, hipStream_t st);

}  // namespace sgk
